// svg-ir_amd/csrc/binning.hip -- tile binning: depth sort, instance offsets, key emit, tile sort, tile ranges.
//
// Replaces cub::DeviceScan::InclusiveSum (rasterizer_impl.cu:307), duplicateWithKeys (:70-111),
// cub::DeviceRadixSort::SortPairs on 64-bit (tile|depth) keys (:333-338) and identifyTileRanges (:116-138).
//
// MI355X-first re-design (same result, ~1/4 of the sort traffic):
//   the reference sorts R (Gaussian,tile) instances by a 64-bit key in 6 radix passes.  Here the P Gaussians
//   are sorted ONCE by their 32-bit depth key (stable, ties by ascending id), instances are emitted in that
//   order, and the R instances are then stably sorted by the tile id alone (ceil(log2 T) bits => 2 passes of
//   6-7 bits at 800x800 / 1600x1600).  A stable sort by tile of a depth-ordered list is exactly the
//   (tile, depth, id) order the reference's stable 64-bit sort produces (quirk Q12).
//
// The radix pass is a hand-written stable LSD pass: per-block digit histogram -> one-block exclusive scan of the
// [digit][block] table -> stable scatter using wave-level digit matching (ballots) for the in-wave rank.
#include "common.hpp"

namespace svgir {

namespace {

// ---- radix pass --------------------------------------------------------------------------------------------
// ITEMS keys per thread (1024 or 4096 keys per block).  FUSED: the digit table is [block][256] and the scatter kernel
// derives its own output cursors from it (no separate scan launch); otherwise the table is [digit][block] and a
// one-block scan kernel turns it into exclusive offsets (very large inputs).
template <int ITEMS, bool FUSED>
__global__ void __launch_bounds__(BLOCK) radix_hist_kernel(const uint32_t* __restrict__ keys, int n, int bit_lo,
                                                           uint32_t mask, int nblocks, uint32_t* __restrict__ table) {
    __shared__ uint32_t hist[256];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const int base = blockIdx.x * (BLOCK * ITEMS);
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const int e = base + i * BLOCK + threadIdx.x;
        if (e < n) atomicAdd(&hist[(keys[e] >> bit_lo) & mask], 1u);
    }
    __syncthreads();
    if (FUSED) table[(size_t)blockIdx.x * 256 + threadIdx.x] = hist[threadIdx.x];
    else if (threadIdx.x <= mask) table[(size_t)threadIdx.x * nblocks + blockIdx.x] = hist[threadIdx.x];
}

// One-block exclusive scan of `count` uint32 values, in place.
__global__ void __launch_bounds__(1024) table_scan_kernel(uint32_t* __restrict__ table, int count) {
    __shared__ uint32_t wsum[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int per = (count + 1023) / 1024;
    const int lo = min(count, t * per), hi = min(count, lo + per);
    uint32_t s = 0;
    for (int i = lo; i < hi; i++) s += table[i];
    // block exclusive scan of s
    uint32_t incl = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    if (t == 0) {
        uint32_t acc = 0;
        for (int w = 0; w < 16; w++) { const uint32_t v = wsum[w]; wsum[w] = acc; acc += v; }
    }
    __syncthreads();
    uint32_t run = wsum[wave] + incl - s;
    for (int i = lo; i < hi; i++) { const uint32_t v = table[i]; table[i] = run; run += v; }
}

template <int ITEMS, bool FUSED>
__global__ void __launch_bounds__(BLOCK) radix_scatter_kernel(const uint32_t* __restrict__ kin,
                                                              const uint32_t* __restrict__ vin,
                                                              uint32_t* __restrict__ kout, uint32_t* __restrict__ vout,
                                                              int n, int bit_lo, int nbits, int nblocks,
                                                              const uint32_t* __restrict__ table) {
    __shared__ uint32_t running[256];     // global output cursor per digit for this block
    __shared__ uint32_t wave_cnt[4][256];  // per-wave digit counts of the current chunk
    __shared__ uint32_t wsum[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t mask = (1u << nbits) - 1;
    if (FUSED) {
        // cursor[d] = sum over digits d' < d of (count of d' in all blocks) + count of d in the blocks before this one
        uint32_t before = 0, total = 0;
        for (int b = 0; b < nblocks; b++) {
            const uint32_t c = table[(size_t)b * 256 + t];
            total += c;
            if (b < (int)blockIdx.x) before += c;
        }
        uint32_t incl = total;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; w++) woff += wsum[w];
        running[t] = woff + incl - total + before;
    } else {
        running[t] = (t <= (int)mask) ? table[(size_t)t * nblocks + blockIdx.x] : 0;
    }
    const int base = blockIdx.x * (BLOCK * ITEMS);
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int c = 0; c < ITEMS; c++) {
#pragma unroll
        for (int w = 0; w < 4; w++) wave_cnt[w][t] = 0;
        __syncthreads();
        const int e = base + c * BLOCK + t;
        const bool valid = e < n;
        uint32_t k = 0, v = 0, d = 0;
        if (valid) { k = kin[e]; v = vin[e]; d = (k >> bit_lo) & mask; }
        // lanes of this wave holding the same digit
        unsigned long long same = __ballot(valid);
        for (int b = 0; b < nbits; b++) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? m : ~m;
        }
        const uint32_t rank_in_wave = __popcll(same & lt_mask);
        if (valid && rank_in_wave == 0) wave_cnt[wave][d] = __popcll(same);
        __syncthreads();
        uint32_t pos = 0;
        if (valid) {
            pos = running[d] + rank_in_wave;
            for (int w = 0; w < wave; w++) pos += wave_cnt[w][d];
        }
        __syncthreads();
        if (t <= (int)mask) running[t] += wave_cnt[0][t] + wave_cnt[1][t] + wave_cnt[2][t] + wave_cnt[3][t];
        if (valid) { kout[pos] = k; vout[pos] = v; }
        __syncthreads();
    }
}

// ---- instance offsets: exclusive scan of tiles[order[i]] ---------------------------------------------------
__device__ __forceinline__ uint32_t block_exclusive_scan_2048(uint32_t (&v)[8], uint32_t* wsum, uint32_t& total) {
    // each thread owns 8 consecutive values; returns the exclusive prefix of the thread's first value
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += v[i];
    uint32_t incl = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { const uint32_t x = wsum[w]; if (w < wave) woff += x; tot += x; }
    total = tot;
    return woff + incl - s;
}

__global__ void __launch_bounds__(BLOCK) offsets_reduce_kernel(const uint32_t* __restrict__ tiles,
                                                               const uint32_t* __restrict__ order, int n,
                                                               uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t wsum[4];
    uint32_t v[8];
    const int base = blockIdx.x * SCAN_BLOCK_ELEMS + threadIdx.x * 8;
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = (base + i < n) ? tiles[order[base + i]] : 0;
    uint32_t total;
    block_exclusive_scan_2048(v, wsum, total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

__global__ void __launch_bounds__(BLOCK) offsets_write_kernel(const uint32_t* __restrict__ tiles,
                                                              const uint32_t* __restrict__ order, int n,
                                                              const uint32_t* __restrict__ block_sums,
                                                              uint32_t* __restrict__ offsets, int nblocks,
                                                              uint32_t* __restrict__ total_out) {
    __shared__ uint32_t wsum[4];
    uint32_t v[8];
    const int base = blockIdx.x * SCAN_BLOCK_ELEMS + threadIdx.x * 8;
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = (base + i < n) ? tiles[order[base + i]] : 0;
    uint32_t total;
    uint32_t run = block_sums[blockIdx.x] + block_exclusive_scan_2048(v, wsum, total);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (base + i < n) offsets[base + i] = run;
        run += v[i];
    }
    if (blockIdx.x == nblocks - 1 && threadIdx.x == 0) total_out[0] = block_sums[blockIdx.x] + total;
}

// ---- emit --------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BLOCK) emit_kernel(int P, const uint32_t* __restrict__ order,
                                                     const uint32_t* __restrict__ tiles,
                                                     const uint32_t* __restrict__ offsets, const float* __restrict__ rec,
                                                     const int32_t* __restrict__ radii, int gx, int gy,
                                                     uint32_t* __restrict__ tile_keys, uint32_t* __restrict__ vals) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= P) return;
    const uint32_t g = order[i];
    if (tiles[g] == 0) return;
    const float px = rec[(size_t)g * REC + R_X], py = rec[(size_t)g * REC + R_Y];
    const float r = (float)radii[g];
    // same rectangle as the preprocess stage (auxiliary.h:53-63); plain divisions, nothing to contract
    const int x0 = min(gx, max(0, (int)((px - r) / TILE))), y0 = min(gy, max(0, (int)((py - r) / TILE)));
    const int x1 = min(gx, max(0, (int)((px + r + TILE - 1) / TILE))), y1 = min(gy, max(0, (int)((py + r + TILE - 1) / TILE)));
    uint32_t off = offsets[i];
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++) {
            tile_keys[off] = (uint32_t)(y * gx + x);
            vals[off] = g;
            off++;
        }
}

__global__ void __launch_bounds__(BLOCK) ranges_kernel(int R, const uint32_t* __restrict__ tile_keys,
                                                       uint32_t* __restrict__ ranges) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= R) return;
    const uint32_t cur = tile_keys[i];
    if (i == 0) ranges[2 * cur] = 0;
    else {
        const uint32_t prev = tile_keys[i - 1];
        if (cur != prev) { ranges[2 * prev + 1] = (uint32_t)i; ranges[2 * cur] = (uint32_t)i; }
    }
    if (i == R - 1) ranges[2 * cur + 1] = (uint32_t)R;
}

// One-block counting sort of the T tiles by descending list length (1024 length buckets).
__global__ void __launch_bounds__(1024) tile_order_kernel(const uint32_t* __restrict__ ranges, int T,
                                                          uint32_t* __restrict__ order) {
    __shared__ uint32_t hist[1024];
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t maxlen_s;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    hist[t] = 0;
    if (t == 0) maxlen_s = 0;
    __syncthreads();
    uint32_t mx = 0;
    for (int i = t; i < T; i += 1024) mx = max(mx, ranges[2 * i + 1] - ranges[2 * i]);
    atomicMax(&maxlen_s, mx);
    __syncthreads();
    const uint32_t maxlen = maxlen_s + 1;
    for (int i = t; i < T; i += 1024) {
        const uint32_t len = ranges[2 * i + 1] - ranges[2 * i];
        const uint32_t b = 1023u - min(1023u, (uint32_t)(((unsigned long long)len * 1024ull) / maxlen));
        atomicAdd(&hist[b], 1u);
    }
    __syncthreads();
    // exclusive scan of hist over the 1024 threads
    const uint32_t v = hist[t];
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; w++) woff += wsum[w];
    __syncthreads();
    hist[t] = woff + incl - v;  // becomes the bucket cursor
    __syncthreads();
    for (int i = t; i < T; i += 1024) {
        const uint32_t len = ranges[2 * i + 1] - ranges[2 * i];
        const uint32_t b = 1023u - min(1023u, (uint32_t)(((unsigned long long)len * 1024ull) / maxlen));
        order[atomicAdd(&hist[b], 1u)] = (uint32_t)i;
    }
}

}  // namespace

void launch_tile_order(const uint32_t* ranges, int T, uint32_t* order, hipStream_t s) {
    hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, s, ranges, T, order);
}

template <int ITEMS, bool FUSED>
static void radix_pass_impl(const uint32_t* kin, const uint32_t* vin, uint32_t* kout, uint32_t* vout, int n, int bit_lo,
                            int nbits, uint32_t* table, hipStream_t s) {
    const int per = BLOCK * ITEMS;
    const int nb = (n + per - 1) / per;
    const uint32_t mask = (1u << nbits) - 1;
    hipLaunchKernelGGL((radix_hist_kernel<ITEMS, FUSED>), dim3(nb), dim3(BLOCK), 0, s, kin, n, bit_lo, mask, nb, table);
    if (!FUSED) hipLaunchKernelGGL(table_scan_kernel, dim3(1), dim3(1024), 0, s, table, (int)((mask + 1) * nb));
    hipLaunchKernelGGL((radix_scatter_kernel<ITEMS, FUSED>), dim3(nb), dim3(BLOCK), 0, s, kin, vin, kout, vout, n,
                       bit_lo, nbits, nb, table);
}

void launch_radix_pass(const uint32_t* kin, const uint32_t* vin, uint32_t* kout, uint32_t* vout, int n, int bit_lo,
                       int nbits, uint32_t* table, hipStream_t s) {
    if (n <= 0) return;
    // small inputs: 1024 keys per block for parallelism; medium: 4096; both with the scan fused into the scatter.
    if (n <= 256 * 1024) radix_pass_impl<4, true>(kin, vin, kout, vout, n, bit_lo, nbits, table, s);
    else if (n <= 512 * 4096) radix_pass_impl<16, true>(kin, vin, kout, vout, n, bit_lo, nbits, table, s);
    else radix_pass_impl<16, false>(kin, vin, kout, vout, n, bit_lo, nbits, table, s);
}

void launch_offsets_scan(const uint32_t* tiles, const uint32_t* order, uint32_t* offsets, uint32_t* scan_tmp, int n,
                         uint32_t* total_out, hipStream_t s) {
    const int nb = scan_blocks(n);
    hipLaunchKernelGGL(offsets_reduce_kernel, dim3(nb), dim3(BLOCK), 0, s, tiles, order, n, scan_tmp);
    hipLaunchKernelGGL(table_scan_kernel, dim3(1), dim3(1024), 0, s, scan_tmp, nb);
    hipLaunchKernelGGL(offsets_write_kernel, dim3(nb), dim3(BLOCK), 0, s, tiles, order, n, scan_tmp, offsets, nb,
                       total_out);
}

void launch_emit(int P, const uint32_t* order, const uint32_t* tiles, const uint32_t* offsets, const float* rec,
                 const int32_t* radii, int gx, int gy, uint32_t* tile_keys, uint32_t* vals, hipStream_t s) {
    hipLaunchKernelGGL(emit_kernel, dim3((P + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, P, order, tiles, offsets, rec,
                       radii, gx, gy, tile_keys, vals);
}

void launch_ranges(int R, const uint32_t* tile_keys, uint32_t* ranges, int T, hipStream_t s) {
    hipMemsetAsync(ranges, 0, (size_t)T * 8, s);
    if (R > 0) hipLaunchKernelGGL(ranges_kernel, dim3((R + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, R, tile_keys, ranges);
}

}  // namespace svgir
