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
// The radix pass is a hand-written stable LSD pass of TWO kernels: per-block digit histogram (+ per-group-of-32-blocks
// digit totals) -> stable scatter.  The scatter block derives its own output cursors from the group totals and the
// <= 31 histogram rows of the preceding blocks of its group (a two-level prefix, ~50 coalesced 1 KB row reads per
// block), so no separate scan kernel sits between the two; the in-wave rank uses wave-level digit matching (ballots).
#include <algorithm>

#include "common.hpp"

namespace svgir {

namespace {

// ---- radix pass --------------------------------------------------------------------------------------------
// Digit table layout: table[block][256] counts; gtot[group][256] = counts summed over the GS blocks of a group.
constexpr int GS = 32;

template <int ITEMS>
__global__ void __launch_bounds__(BLOCK) radix_hist_kernel(const uint32_t* __restrict__ keys, int n_cap,
                                                           const uint32_t* __restrict__ n_dev, int bit_lo,
                                                           uint32_t mask, uint32_t* __restrict__ table,
                                                           uint32_t* __restrict__ gtot) {
    __shared__ uint32_t hist[256];
    // element count: launch-time bound n_cap, or -- when the count is still being computed on the device at launch
    // time (speculative launch before the host has read it) -- the device-side value clamped to that bound
    const int n = n_dev ? (int)min((uint32_t)n_cap, n_dev[0]) : n_cap;
    hist[threadIdx.x] = 0;
    __syncthreads();
    const int base = blockIdx.x * (BLOCK * ITEMS);
    uint32_t k[ITEMS];   // all loads first (clamped index), then the LDS atomics: one memory latency, not ITEMS
#pragma unroll
    for (int i = 0; i < ITEMS; i++) k[i] = keys[max(0, min(base + i * BLOCK + (int)threadIdx.x, n - 1))];
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const int e = base + i * BLOCK + threadIdx.x;
        if (e < n) atomicAdd(&hist[(k[i] >> bit_lo) & mask], 1u);
    }
    __syncthreads();
    const uint32_t c = hist[threadIdx.x];
    table[(size_t)blockIdx.x * 256 + threadIdx.x] = c;
    if (c) atomicAdd(&gtot[(size_t)(blockIdx.x / GS) * 256 + threadIdx.x], c);
}

template <int ITEMS>
__global__ void __launch_bounds__(BLOCK) radix_scatter_kernel(const uint32_t* __restrict__ kin,
                                                              const uint32_t* __restrict__ vin,
                                                              uint32_t* __restrict__ kout, uint32_t* __restrict__ vout,
                                                              int n_cap, const uint32_t* __restrict__ n_dev, int bit_lo, int nbits,
                                                              const uint32_t* __restrict__ table,
                                                              const uint32_t* __restrict__ gtot, int ngroups) {
    __shared__ uint32_t wcnt[4][256];   // per-wave digit counters, later the waves' output cursors
    __shared__ uint32_t wtot[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t mask = (1u << nbits) - 1;
    const int n = n_dev ? (int)min((uint32_t)n_cap, n_dev[0]) : n_cap;
    // wave w owns the w-th quarter of the block's keys (consecutive keys: round-major, lane-minor = key order)
    const int base = blockIdx.x * (BLOCK * ITEMS) + wave * (64 * ITEMS);
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    // all loads of the block up front
    uint32_t ks[ITEMS], vs[ITEMS], rk[ITEMS];
#pragma unroll
    for (int c = 0; c < ITEMS; c++) {
        const int e = base + c * 64 + lane;
        ks[c] = e < n ? kin[e] : 0u;
        vs[c] = e < n ? vin[e] : 0u;
    }
#pragma unroll
    for (int w = 0; w < 4; w++) wcnt[w][t] = 0;
    // Output cursor of digit t for this block:
    //   sum_{d < t} total[d]  +  sum_{groups before mine} gtot[g][t]  +  sum_{blocks before me in my group} table[b][t]
    uint32_t cursor;
    {
        const int g = blockIdx.x / GS;
        uint32_t tot = 0, pre = 0;
#pragma unroll 8
        for (int gg = 0; gg < ngroups; gg++) {
            const uint32_t v = gtot[(size_t)gg * 256 + t];
            tot += v;
            pre += gg < g ? v : 0u;
        }
#pragma unroll 8
        for (int b = g * GS; b < (int)blockIdx.x; b++) pre += table[(size_t)b * 256 + t];
        uint32_t incl = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();   // (also: the wave counters are zero)
        uint32_t woff = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) woff += w < wave ? wtot[w] : 0u;
        cursor = woff + incl - tot + pre;
    }
    // rank of every key among the keys of its WAVE with the same digit: wave-level digit matching (ballots) against a
    // wave-private digit counter in LDS -- no workgroup barrier inside the ranking
#pragma unroll
    for (int c = 0; c < ITEMS; c++) {
        const bool valid = base + c * 64 + lane < n;
        const uint32_t d = (ks[c] >> bit_lo) & mask;
        unsigned long long same = __ballot(valid);
        for (int b = 0; b < nbits; b++) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? m : ~m;
        }
        const uint32_t in_round = (uint32_t)__popcll(same & lt_mask);
        const uint32_t prior = wcnt[wave][d];   // (the DS operations of a wave execute in order: every lane reads before the leader writes)
        rk[c] = prior + in_round;
        if (valid && in_round == 0) wcnt[wave][d] = prior + (uint32_t)__popcll(same);
    }
    __syncthreads();
    if (ITEMS >= 16) {
        // Large inputs (bandwidth, not launch latency, sets the time): the block's keys are first put in digit order in LDS, then
        // written out -- consecutive lanes write consecutive addresses of a digit's run (a block of 4096 keys holds runs of ~32 per 7-bit
        // digit) instead of 64 lanes writing to 64 different runs.
        __shared__ uint32_t sK[BLOCK * (ITEMS >= 16 ? ITEMS : 1)], sV[BLOCK * (ITEMS >= 16 ? ITEMS : 1)];
        __shared__ uint32_t bstart[256], gcur[256];
        const uint32_t c0 = wcnt[0][t], c1 = wcnt[1][t], c2 = wcnt[2][t], c3 = wcnt[3][t];
        const uint32_t bc = c0 + c1 + c2 + c3;            // keys of digit t in this block
        uint32_t incl = bc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        __syncthreads();                                   // (wtot was read by every thread above)
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        uint32_t woff = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) woff += w < wave ? wtot[w] : 0u;
        const uint32_t bs = woff + incl - bc;              // first slot of digit t in the block's staging order
        bstart[t] = bs; gcur[t] = cursor;
        wcnt[0][t] = bs; wcnt[1][t] = bs + c0; wcnt[2][t] = bs + c0 + c1; wcnt[3][t] = bs + c0 + c1 + c2;   // the waves' staging cursors
        __syncthreads();
#pragma unroll
        for (int c = 0; c < ITEMS; c++) {
            if (base + c * 64 + lane < n) {
                const uint32_t lp = wcnt[wave][(ks[c] >> bit_lo) & mask] + rk[c];
                sK[lp] = ks[c]; sV[lp] = vs[c];
            }
        }
        __syncthreads();
        const int nblk = min(BLOCK * ITEMS, n - (int)blockIdx.x * (BLOCK * ITEMS));
#pragma unroll
        for (int c = 0; c < ITEMS; c++) {
            const int i = c * BLOCK + t;
            if (i < nblk) {
                const uint32_t k = sK[i], d = (k >> bit_lo) & mask;
                const uint32_t pos = gcur[d] + ((uint32_t)i - bstart[d]);
                kout[pos] = k;
                vout[pos] = sV[i];
            }
        }
        return;
    }
    {   // digit t: the waves' cursors = block cursor + counts of the waves in front
        const uint32_t c0 = wcnt[0][t], c1 = wcnt[1][t], c2 = wcnt[2][t];
        __syncthreads();
        wcnt[0][t] = cursor; wcnt[1][t] = cursor + c0; wcnt[2][t] = cursor + c0 + c1; wcnt[3][t] = cursor + c0 + c1 + c2;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < ITEMS; c++) {
        if (base + c * 64 + lane < n) {
            const uint32_t pos = wcnt[wave][(ks[c] >> bit_lo) & mask] + rk[c];
            kout[pos] = ks[c];
            vout[pos] = vs[c];
        }
    }
}

// ---- single-pass tile sort (T <= 4096 tiles) ----------------------------------------------------------------------
// The tile id has <= 12 bits at 800 x 800 and below: instead of two 6-bit radix passes (four launches) + a ranges kernel, ONE stable
// counting sort over the whole id -- (1) per-workgroup histograms of TS12_KEYS keys over the tile ids (LDS atomics), (2) per tile
// the exclusive prefix over the workgroups and the tile's total, (3) scatter: every workgroup scans the totals itself (tile bases = the
// tiles' RANGES: workgroup 0 writes them out, identifyTileRanges for free), adds its row of prefixes, ranks its keys stably (wave-level
// matching on the 12 bits against wave-private 16-bit LDS counters) and writes.  Rows hold nbins = T rounded up to 256 counters.
__global__ void __launch_bounds__(BLOCK) ts12_hist_kernel(const uint32_t* __restrict__ keys, int n_cap, const uint32_t* __restrict__ n_dev,
                                                          uint32_t* __restrict__ table, int nbins) {
    __shared__ uint32_t hist[TS12_BINS];
    const int n = n_dev ? (int)min((uint32_t)n_cap, n_dev[0]) : n_cap;
    const int t = threadIdx.x;
    for (int i = t; i < nbins; i += BLOCK) hist[i] = 0u;
    __syncthreads();
    const int base = blockIdx.x * TS12_KEYS;
    uint32_t k[TS12_KEYS / BLOCK];
#pragma unroll
    for (int i = 0; i < TS12_KEYS / BLOCK; i++) k[i] = keys[max(0, min(base + i * BLOCK + t, n - 1))];
#pragma unroll
    for (int i = 0; i < TS12_KEYS / BLOCK; i++)
        if (base + i * BLOCK + t < n) atomicAdd(&hist[k[i] & (TS12_BINS - 1)], 1u);
    __syncthreads();
    uint32_t* row = table + (size_t)blockIdx.x * nbins;
    for (int i = t; i < nbins; i += BLOCK) row[i] = hist[i];
}

// per tile id: exclusive prefix of the workgroups' counts (in place) and the total.  A workgroup = 16 tile ids x 16 row segments: a thread
// first sums its segment (loads only: all in flight together), the 16 segment sums of a tile meet in LDS, then the thread rewrites its
// segment as running prefixes -- two short memory round trips instead of one per row.
__global__ void __launch_bounds__(BLOCK) ts12_colscan_kernel(uint32_t* __restrict__ table, int nb, uint32_t* __restrict__ totals, int nbins) {
    __shared__ uint32_t psum[16][17];
    const int col = threadIdx.x & 15, seg = threadIdx.x >> 4;
    const int d = blockIdx.x * 16 + col;
    const int rps = (nb + 15) / 16, r0 = min(nb, seg * rps), r1 = min(nb, r0 + rps);
    uint32_t* colp = table + d;
    uint32_t sum = 0;
    int r = r0;
    for (; r + 8 <= r1; r += 8) {
        uint32_t c[8];
#pragma unroll
        for (int j = 0; j < 8; j++) c[j] = colp[(size_t)(r + j) * nbins];
#pragma unroll
        for (int j = 0; j < 8; j++) sum += c[j];
    }
    for (; r < r1; r++) sum += colp[(size_t)r * nbins];
    psum[seg][col] = sum;
    __syncthreads();
    uint32_t run = 0, all = 0;
#pragma unroll
    for (int sg = 0; sg < 16; sg++) { const uint32_t v = psum[sg][col]; run += sg < seg ? v : 0u; all += v; }
    if (seg == 0) totals[d] = all;
    r = r0;
    for (; r + 8 <= r1; r += 8) {
        uint32_t c[8];
#pragma unroll
        for (int j = 0; j < 8; j++) c[j] = colp[(size_t)(r + j) * nbins];
#pragma unroll
        for (int j = 0; j < 8; j++) { colp[(size_t)(r + j) * nbins] = run; run += c[j]; }
    }
    for (; r < r1; r++) { const uint32_t c = colp[(size_t)r * nbins]; colp[(size_t)r * nbins] = run; run += c; }
}

__global__ void __launch_bounds__(BLOCK) ts12_scatter_kernel(const uint32_t* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                             uint32_t* __restrict__ kout, uint32_t* __restrict__ vout, int n_cap,
                                                             const uint32_t* __restrict__ n_dev, const uint32_t* __restrict__ table,
                                                             const uint32_t* __restrict__ totals, uint32_t* __restrict__ ranges, int T, int nbins) {
    __shared__ uint32_t cur[TS12_BINS];          // global position of this workgroup's first key of every tile
    __shared__ uint16_t wcnt[4][TS12_BINS];      // per-wave tile counters (a workgroup holds 2048 keys)
    __shared__ uint32_t wtot[4];
    constexpr int PERMAX = TS12_BINS / BLOCK;    // up to 16 consecutive tiles per thread in the scan
    constexpr int ROUNDS = TS12_KEYS / BLOCK;    // 8 rounds of 64 keys per wave
    const int per = nbins / BLOCK;               // (nbins is a multiple of 256)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int n = n_dev ? (int)min((uint32_t)n_cap, n_dev[0]) : n_cap;
    const int base = blockIdx.x * TS12_KEYS + wave * (64 * ROUNDS);   // wave w owns the w-th quarter, round-major = key order
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    uint32_t ks[ROUNDS], vs[ROUNDS];
    uint16_t rk[ROUNDS];
#pragma unroll
    for (int c = 0; c < ROUNDS; c++) {
        const int e = base + c * 64 + lane;
        ks[c] = e < n ? kin[e] : 0u;
        vs[c] = e < n ? vin[e] : 0u;
    }
    // ---- tile bases: exclusive scan of the totals (thread t owns tiles per t .. per t + per - 1) ----
    uint32_t tot[PERMAX], pre[PERMAX];
    {
        const uint32_t* tp = totals + t * per;
        const uint32_t* rp = table + (size_t)blockIdx.x * nbins + t * per;
#pragma unroll
        for (int j = 0; j < PERMAX; j++) { tot[j] = j < per ? tp[j] : 0u; pre[j] = j < per ? rp[j] : 0u; }
    }
    uint32_t sum = 0;
#pragma unroll
    for (int j = 0; j < PERMAX; j++) sum += tot[j];
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wtot[wave] = incl;
#pragma unroll
    for (int w = 0; w < 4; w++)
        for (int i = t; i < nbins / 2; i += BLOCK) reinterpret_cast<uint32_t*>(&wcnt[w][0])[i] = 0u;
    __syncthreads();
    uint32_t run = incl - sum;
#pragma unroll
    for (int w = 0; w < 4; w++) run += w < wave ? wtot[w] : 0u;
#pragma unroll
    for (int j = 0; j < PERMAX; j++) {
        if (j < per) {
            const int tile = t * per + j;
            cur[tile] = run + pre[j];
            if (blockIdx.x == 0 && tile < T) {   // identifyTileRanges (rasterizer_impl.cu:116-138): [start, end), empty tiles stay (0, 0)
                ranges[2 * tile] = tot[j] ? run : 0u;
                ranges[2 * tile + 1] = tot[j] ? run + tot[j] : 0u;
            }
            run += tot[j];
        }
    }
    // ---- stable rank of every key among the keys of its WAVE with the same tile id ----
#pragma unroll
    for (int c = 0; c < ROUNDS; c++) {
        const bool valid = base + c * 64 + lane < n;
        const uint32_t d = ks[c] & (TS12_BINS - 1);
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 12; b++) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? m : ~m;
        }
        const uint32_t in_round = (uint32_t)__popcll(same & lt_mask);
        const uint32_t prior = wcnt[wave][d];   // (the DS operations of a wave execute in order: every lane reads before the leader writes)
        rk[c] = (uint16_t)(prior + in_round);
        if (valid && in_round == 0) wcnt[wave][d] = (uint16_t)(prior + (uint32_t)__popcll(same));
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < ROUNDS; c++) {
        if (base + c * 64 + lane < n) {
            const uint32_t d = ks[c] & (TS12_BINS - 1);
            uint32_t pos = cur[d] + rk[c];
            if (wave > 0) pos += wcnt[0][d];
            if (wave > 1) pos += wcnt[1][d];
            if (wave > 2) pos += wcnt[2][d];
            kout[pos] = ks[c];
            vout[pos] = vs[c];
        }
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

// (the counts gathered here are parked, in depth order, in `offsets`: the write kernel reads them back with coalesced loads instead of
// repeating the P random gathers, then overwrites them with the prefix)
__global__ void __launch_bounds__(BLOCK) offsets_reduce_kernel(const uint32_t* __restrict__ tiles,
                                                               const uint32_t* __restrict__ order, int n,
                                                               uint32_t* __restrict__ counts_out,
                                                               uint32_t* __restrict__ block_sums,
                                                               const uint32_t* __restrict__ key_top, int n_key_top,
                                                               uint32_t* __restrict__ block_key) {
    __shared__ uint32_t wsum[4];
    if (threadIdx.x < 64) {   // this block's slice of the preprocess waves' depth-key summaries {AND << 8 | OR}: SCAN_BLOCK_ELEMS / 64 = 32 entries
        const int j = blockIdx.x * (SCAN_BLOCK_ELEMS / 64) + (int)threadIdx.x;
        uint32_t kv = (threadIdx.x < SCAN_BLOCK_ELEMS / 64 && j < n_key_top) ? key_top[j] : 0xff00u;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)kv, d); kv = (kv & o & 0xff00u) | ((kv | o) & 0xffu); }
        if (threadIdx.x == 0) block_key[blockIdx.x] = kv;
    }
    uint32_t v[8];
    const int base = blockIdx.x * SCAN_BLOCK_ELEMS + threadIdx.x * 8;
    uint32_t oi[8];   // two rounds of independent loads instead of 8 dependent pairs
#pragma unroll
    for (int i = 0; i < 8; i++) oi[i] = order[min(base + i, n - 1)];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = tiles[2 * oi[i]];   // (common.hpp GeomLayout::tiles: {count, rectangle})
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = (base + i < n) ? v[i] : 0u;
    if (base + 8 <= n) {   // (P-sized arrays start 256-byte aligned: two 16-byte stores)
        reinterpret_cast<uint4*>(counts_out + base)[0] = make_uint4(v[0], v[1], v[2], v[3]);
        reinterpret_cast<uint4*>(counts_out + base)[1] = make_uint4(v[4], v[5], v[6], v[7]);
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++) if (base + i < n) counts_out[base + i] = v[i];
    }
    uint32_t total;
    block_exclusive_scan_2048(v, wsum, total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

__global__ void __launch_bounds__(BLOCK) offsets_write_kernel(int n,
                                                              const uint32_t* __restrict__ block_sums,
                                                              uint32_t* __restrict__ offsets, int nblocks,
                                                              uint32_t* __restrict__ total_out,
                                                              const uint32_t* __restrict__ block_key,
                                                              const uint32_t* __restrict__ violation,
                                                              unsigned long long* __restrict__ host_out, uint32_t host_tag) {
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t psum[4];
    __shared__ uint32_t ksum[4];
    uint32_t v[8];
    const int base = blockIdx.x * SCAN_BLOCK_ELEMS + threadIdx.x * 8;
    if (base + 8 <= n) {   // the counts the reduce kernel parked in `offsets` (depth order)
        const uint4 a4 = reinterpret_cast<const uint4*>(offsets + base)[0], b4 = reinterpret_cast<const uint4*>(offsets + base)[1];
        v[0] = a4.x; v[1] = a4.y; v[2] = a4.z; v[3] = a4.w; v[4] = b4.x; v[5] = b4.y; v[6] = b4.z; v[7] = b4.w;
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = (base + i < n) ? offsets[base + i] : 0u;
    }
    // sum of the preceding blocks' totals (the block sums are few: every block adds them up itself, no scan kernel)
    const bool last = blockIdx.x == (unsigned)(nblocks - 1);
    uint32_t pre = 0, kv = 0xff00u;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += BLOCK) pre += block_sums[b];
    if (last)   // the last block walks all the blocks anyway: it also folds their depth-key summaries {AND << 8 | OR}
        for (int b = threadIdx.x; b < nblocks; b += BLOCK) { const uint32_t x = block_key[b]; kv = (kv & x & 0xff00u) | ((kv | x) & 0xffu); }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        pre += __shfl_xor(pre, d);
        const uint32_t o = (uint32_t)__shfl_xor((int)kv, d);
        kv = (kv & o & 0xff00u) | ((kv | o) & 0xffu);
    }
    if ((threadIdx.x & 63) == 0) { psum[threadIdx.x >> 6] = pre; ksum[threadIdx.x >> 6] = kv; }
    uint32_t total;
    uint32_t run = block_exclusive_scan_2048(v, wsum, total);   // contains a __syncthreads()
    const uint32_t before = psum[0] + psum[1] + psum[2] + psum[3];
    run += before;
    uint32_t span = 0;   // 1 + position (in depth order) of this thread's last Gaussian that touches a tile
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (base + i < n) offsets[base + i] = run;
        run += v[i];
        if (v[i] != 0u) span = (uint32_t)(base + i + 1);
    }
    // total_out[3] = the visible span of the depth order: order[0 .. span) holds every Gaussian with tiles > 0 (they sort in front of
    // the culled ones) -- the working set of the fused shading (api.hip).  One atomic per wave that holds a visible Gaussian.
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) span = max(span, (uint32_t)__shfl_xor((int)span, d));
    if ((threadIdx.x & 63) == 0 && span != 0u) atomicMax(total_out + 3, span);
    if (last && threadIdx.x == 0) {
        const uint32_t R = before + total;
        const uint32_t summary = (ksum[0] & ksum[1] & ksum[2] & ksum[3] & 0xff00u) | ((ksum[0] | ksum[1] | ksum[2] | ksum[3]) & 0xffu);
        total_out[0] = R;
        total_out[2] = summary;
        // the host's copy: tagged 8-byte stores into pinned host memory -- no copy operation and no event on the stream; the host
        // recognises the values of THIS forward by the tag (api.hip): {R} and {prefilter violation << 16 | summary}
        if (host_out) {
            const uint32_t viol = violation ? (violation[0] != 0u ? 1u : 0u) : 0u;
            __hip_atomic_store(host_out, ((unsigned long long)host_tag << 32) | R, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(host_out + 1, ((unsigned long long)host_tag << 32) | (viol << 16) | summary, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---- emit --------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BLOCK) emit_kernel(int P, const uint32_t* __restrict__ order,
                                                     const uint32_t* __restrict__ tiles,
                                                     const uint32_t* __restrict__ offsets, float* __restrict__ rec,
                                                     const int32_t* __restrict__ radii, int gx, int gy,
                                                     uint32_t* __restrict__ tile_keys, uint32_t* __restrict__ vals,
                                                     uint32_t cap, uint32_t* __restrict__ ranges, int n_ranges,
                                                     uint32_t* __restrict__ seg_count, uint32_t* __restrict__ gtot,
                                                     int n_gtot, const uint32_t* __restrict__ span) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    // piggy-backed initialisation of three small tables used by later stages (saves three memset launches)
    for (int j = i; j < n_ranges; j += gridDim.x * BLOCK) ranges[j] = 0u;
    for (int j = i; j < n_gtot; j += gridDim.x * BLOCK) gtot[j] = 0u;
    if (i == 0) seg_count[0] = 0u;
    // One wave = 64 consecutive Gaussians of the depth order; their instances are one contiguous run of the output (offsets is the
    // exclusive scan in this order).  The run is written COOPERATIVELY: instance j of the wave goes to lane j & 63 of round j >> 6, which
    // finds its Gaussian by a binary search over the lanes' local offsets (6 steps through the LDS crossbar) -- every store instruction
    // writes 256 contiguous bytes and no lane idles while another one walks a large rectangle.  (A thread per Gaussian looping over its
    // rectangle wrote 64 dwords 24 bytes apart per instruction and ran as long as the wave's largest splat: 80 us for 5.5 M instances.)
    const int lane = threadIdx.x & 63;
    // (order[0 .. *span) holds every Gaussian that touches a tile -- the culled ones, more than half of a closed surface's surfels, sort
    // behind them)
    const uint32_t nvis = min((uint32_t)P, span[0]);
    if ((uint32_t)(i - lane) >= nvis) return;   // (uniform: the whole wave lies behind the visible span)
    const bool valid = (uint32_t)i < nvis;
    const uint32_t g = valid ? order[i] : 0u;
    const uint32_t off = valid ? offsets[i] : 0u;
    // (requested together, one latency behind `order`: the Gaussian's tile count and rectangle, as the preprocess stage computed them --
    // auxiliary.h:53-63 -- in one 8-byte gather)
    const uint2 tr = valid ? reinterpret_cast<const uint2*>(tiles)[g] : make_uint2(0u, 0u);
    const uint32_t n = tr.x, rect = tr.y;
    if (n != 0u) {   // for the backward's gradient rows: first instance index (emit order) and tile rectangle of this Gaussian
        rec[(size_t)g * REC + R_IBASE] = __builtin_bit_cast(float, off);
        rec[(size_t)g * REC + R_RECT] = __builtin_bit_cast(float, rect);
    }
    // local exclusive offsets of the wave's Gaussians and the run's length
    uint32_t incl = n;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= d) incl += o;
    }
    const uint32_t loc = incl - n;
    const uint32_t M = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(off - loc));   // (lane 0 is valid; off = base + loc on every valid lane)
    for (uint32_t j0 = 0; j0 < M; j0 += 64u) {
        const uint32_t j = j0 + (uint32_t)lane;
        // owner = the last lane whose local offset is <= j (lanes without instances share their successor's offset and lose to it)
        int lo = 0;
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1) {
            const int mid = lo + step;   // (<= 63)
            const uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute(mid << 2, (int)loc);
            lo = v <= j ? mid : lo;
        }
        const uint32_t t = j - (uint32_t)__builtin_amdgcn_ds_bpermute(lo << 2, (int)loc);
        const uint32_t go = (uint32_t)__builtin_amdgcn_ds_bpermute(lo << 2, (int)g);
        const uint32_t ro = (uint32_t)__builtin_amdgcn_ds_bpermute(lo << 2, (int)rect);
        const uint32_t w = max(ro >> 20, 1u), ty = t / w, tx = t - ty * w;
        const uint32_t pos = base + j;
        if (j < M && pos < cap) {   // (pos >= cap only ever for a speculative launch whose capacity guess was too small)
            tile_keys[pos] = (((ro >> 10) & 1023u) + ty) * (uint32_t)gx + ((ro & 1023u) + tx);
            vals[pos] = go;
        }
    }
}

__global__ void __launch_bounds__(BLOCK) ranges_kernel(int R_cap, const uint32_t* __restrict__ R_dev,
                                                       const uint32_t* __restrict__ tile_keys,
                                                       uint32_t* __restrict__ ranges) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    const int R = R_dev ? (int)min((uint32_t)R_cap, R_dev[0]) : R_cap;
    if (i >= R) return;
    const uint32_t cur = tile_keys[i];
    if (i == 0) ranges[2 * cur] = 0;
    else {
        const uint32_t prev = tile_keys[i - 1];
        if (cur != prev) { ranges[2 * prev + 1] = (uint32_t)i; ranges[2 * cur] = (uint32_t)i; }
    }
    if (i == R - 1) ranges[2 * cur + 1] = (uint32_t)R;
}

// One workgroup behind the cull: (a) counting sort of the n work items by descending size: order[] = item ids, largest first -- 1024
// size buckets, one per count below 1023 (exact order there; everything longer shares the first bucket -- those waves start first
// anyway); (b) exclusive prefix sums, in index order, of the counts (the first gradient row of every sub-tile) and of seg_slots(count)
// (its first dumped-state slot), and their totals (device + tagged host copy).  Every wave owns a contiguous chunk of the items and walks
// it 64 at a time; the counts are read ONCE (PIT rounds in registers) and serve the histogram, the wave totals, the prefixes and the
// scatter: one global latency and three barriers from the first load to the last store.
__global__ void __launch_bounds__(1024) order_desc_kernel(const uint32_t* __restrict__ counts, int n,
                                                          uint32_t* __restrict__ order, uint32_t* __restrict__ prefix,
                                                          uint32_t* __restrict__ slot_prefix, uint32_t* __restrict__ total,
                                                          unsigned long long* __restrict__ host_total, uint32_t host_tag,
                                                          uint32_t cap_R, long long cap_slots, uint32_t magic, int order_n) {
    __shared__ uint32_t hist[1024];
    __shared__ uint32_t wsum[16];
    __shared__ unsigned long long wsum2[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    hist[t] = 0;
    const int chunk = ((n + 15) / 16 + 63) / 64 * 64;   // items per wave
    const int c0 = wave * chunk, c1 = min(n, c0 + chunk);
    constexpr int PIT = 12;   // chunk rounds held in registers (n <= 12288 = an 880 x 880 image; beyond that the rounds re-read the counts)
    const bool in_regs = chunk <= PIT * 64;
    uint32_t cv[PIT];
#pragma unroll
    for (int r = 0; r < PIT; r++) cv[r] = (in_regs && c0 + r * 64 + lane < c1) ? counts[c0 + r * 64 + lane] : 0u;   // all loads in flight together
    auto both = [](uint32_t c) { return (unsigned long long)c | ((unsigned long long)seg_slots(c) << 32); };   // low word: counts, high word: slots
    __syncthreads();   // (hist is zero; the loads are still in flight)
    // ---- pass 1: size histogram + wave totals.  Items of size 0 (most of an image is usually empty) all land in the last bucket: they
    // are counted with one atomic per wave and round instead of one per item
    unsigned long long wtot = 0;
    auto count1 = [&](int i0, uint32_t len) {
        const bool valid = i0 + lane < c1;
        const unsigned long long zero = __ballot(valid && len == 0u);
        if (valid && len != 0u) atomicAdd(&hist[1023u - min(1023u, len)], 1u);
        if (lane == 0 && zero) atomicAdd(&hist[1023], (uint32_t)__popcll(zero));
        wtot += both(valid ? len : 0u);
    };
    if (in_regs) {
#pragma unroll
        for (int r = 0; r < PIT; r++)
            if (c0 + r * 64 < c1) count1(c0 + r * 64, cv[r]);   // (uniform)
    } else {
        for (int i0 = c0; i0 < c1; i0 += 64) count1(i0, i0 + lane < c1 ? counts[i0 + lane] : 0u);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wtot += (unsigned long long)__shfl_xor((long long)wtot, d);
    if (lane == 0) wsum2[wave] = wtot;
    __syncthreads();
    // ---- exclusive scan of hist over the 1024 threads -> bucket cursors
    const uint32_t hv = hist[t];
    uint32_t incl = hv;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    // ---- prefixes in index order (between the two barriers of the bucket scan: they only need wsum2)
    // inclusive wave scans on the VALU (DPP row shifts + row broadcasts; a __shfl_up scan is 6 LDS permutes per word)
    auto scan32 = [](uint32_t v) {
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
        return v;
    };
    auto wave_incl = [&](unsigned long long v) {   // (two independent 32-bit scans: neither half overflows into the other)
        return (unsigned long long)scan32((uint32_t)v) | ((unsigned long long)scan32((uint32_t)(v >> 32)) << 32);
    };
    unsigned long long run = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) { const unsigned long long x = wsum2[w]; run += w < wave ? x : 0ull; all += x; }
    if (prefix || slot_prefix) {
        auto round = [&](int i0, uint32_t c) {
            const int i = i0 + lane;
            const unsigned long long v = both(i < c1 ? c : 0u), inc = wave_incl(v);
            if (i < c1) {
                if (prefix) prefix[i] = (uint32_t)(run + inc - v);
                if (slot_prefix) slot_prefix[i] = (uint32_t)((run + inc - v) >> 32);
            }
            run += ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)inc, 63)) |
                   ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(inc >> 32), 63) << 32);
        };
        if (in_regs) {
#pragma unroll
            for (int r = 0; r < PIT; r++)
                if (c0 + r * 64 < c1) round(c0 + r * 64, cv[r]);   // (uniform)
        } else {
            for (int i0 = c0; i0 < c1; i0 += 64) round(i0, i0 + lane < c1 ? counts[i0 + lane] : 0u);
        }
    }
    if (t == 0) {
        if (total) {
            total[1] = (uint32_t)all; total[2] = (uint32_t)(all >> 32);
            // the capacities the forward laid the binning blob out for: the image blob describes its view by itself (api.hip view_from_blob)
            total[3] = magic; total[4] = cap_R; total[5] = (uint32_t)(unsigned long long)cap_slots; total[6] = (uint32_t)((unsigned long long)cap_slots >> 32);
        }
        // the host's copy: {sum, tag of this forward} as 8-byte stores into pinned host memory -- no copy operation and no event on the
        // stream; the host recognises the values by their tag (api.hip view_lookup)
        if (host_total) {
            // (third word: the number of non-empty items -- composite waves with work; the forward picks its occupancy variant from it)
            __hip_atomic_store(host_total + 2, ((unsigned long long)host_tag << 32) | (unsigned long long)((uint32_t)n - hist[1023]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(host_total, ((unsigned long long)host_tag << 32) | (all & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(host_total + 1, ((unsigned long long)host_tag << 32) | (all >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; w++) woff += wsum[w];
    hist[t] = woff + incl - hv;   // first output position of bucket t
    __syncthreads();
    // ---- pass 2: scatter (items of equal size in arrival order; the zero-size items at the very end)
    auto place = [&](int i0, uint32_t len) {
        const int i = i0 + lane;
        const bool valid = i < c1, z = valid && len == 0u;
        const unsigned long long zero = __ballot(z);
        if (valid && len != 0u) order[atomicAdd(&hist[1023u - min(1023u, len)], 1u)] = (uint32_t)i;
        uint32_t zbase = 0;
        if (lane == 0 && zero) zbase = atomicAdd(&hist[1023], (uint32_t)__popcll(zero));
        zbase = (uint32_t)__shfl((int)zbase, 0);
        if (z) order[zbase + (uint32_t)__popcll(zero & lt_mask)] = (uint32_t)i;
    };
    if (in_regs) {
#pragma unroll
        for (int r = 0; r < PIT; r++)
            if (c0 + r * 64 < c1) place(c0 + r * 64, cv[r]);   // (uniform)
    } else {
        for (int i0 = c0; i0 < c1; i0 += 64) place(i0, i0 + lane < c1 ? counts[i0 + lane] : 0u);
    }
    for (int i = n + t; i < order_n; i += 1024) order[i] = ORDER_NONE;   // padding (common.hpp ORDER_NONE)
}

// The same products for launches that fill the machine several times over (api.hip high_fill): ONE LONGEST-FIRST LIST PER XCD, and one
// workgroup per XCD to build it.  Workgroups are dealt to the eight XCDs round-robin (workgroup b of the composite runs on XCD b & 7), and
// each XCD has its own 4 MiB L2: with one global order the ~3 sub-tiles that gather a splat's record and vfeature rows run on three
// different XCDs, and each L2 fetches them again.  Here the image is cut into blocks of 4 x 4 tiles, block (bx, by) belongs to XCD
// (bx + 3 by) & 7 (common.hpp xcd_of_tile); workgroup c counting-sorts the sub-tiles of XCD c by descending count and writes its j-th item
// to order[8 j + c] (the lists differ in length by a few blocks: the tail is padded with ORDER_NONE up to order_n = order_entries()).
// With tens of thousands of waves the XCDs' sums are balanced to a few per cent; with one round of waves they are not (the exact order
// wins there: cfg4 render 220 vs 231 us).  The index-order prefixes are split eight ways too: workgroup c scans the c-th eighth of the
// items behind a reduction of everything in front of it (every workgroup reads all n counts twice: 40 loads per thread at 1600 x 1600),
// and the last one -- which has seen every item -- publishes the totals.  cfg5 (40 000 sub-tiles): 49 us for the single workgroup.
__global__ void __launch_bounds__(1024) order_xcd_kernel(const uint32_t* __restrict__ counts, int n, uint32_t* __restrict__ order,
                                                         uint32_t* __restrict__ prefix, uint32_t* __restrict__ slot_prefix,
                                                         uint32_t* __restrict__ total, unsigned long long* __restrict__ host_total,
                                                         uint32_t host_tag, uint32_t cap_R, long long cap_slots, uint32_t magic, int gx,
                                                         int order_n) {
    __shared__ uint32_t hist[1024];
    __shared__ uint32_t wsum[16];
    __shared__ unsigned long long wsum2[16];
    __shared__ uint32_t wzero[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t c = blockIdx.x;   // XCD whose list this workgroup builds; also its eighth of the prefixes
    hist[t] = 0;
    auto both = [](uint32_t v) { return (unsigned long long)v | ((unsigned long long)seg_slots(v) << 32); };
    const int per = ((n + 7) / 8 + 1023) / 1024 * 1024;       // items per workgroup (prefix part), a multiple of the workgroup size
    const int p0 = min(n, (int)c * per), p1 = min(n, p0 + per);
    __syncthreads();
    // ---- pass 1 over ALL items: histogram of this XCD's items; sum (counts | slots) and number of empty items in front of p0
    unsigned long long before = 0;
    uint32_t zeros = 0;
    for (int i = t; i < n; i += 1024) {
        const uint32_t len = counts[i];
        const int tile = i >> 2;
        if (xcd_of_tile(tile % gx, tile / gx) == c) atomicAdd(&hist[1023u - min(1023u, len)], 1u);
        if (i < p0) before += both(len);
        zeros += len == 0u ? 1u : 0u;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        before += (unsigned long long)__shfl_xor((long long)before, d);
        zeros += (uint32_t)__shfl_xor((int)zeros, d);
    }
    if (lane == 0) { wsum2[wave] = before; wzero[wave] = zeros; }
    __syncthreads();
    unsigned long long run = 0;
    uint32_t n_empty = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) { run += wsum2[w]; n_empty += wzero[w]; }
    // ---- bucket cursors of this XCD
    const uint32_t hv = hist[t];
    uint32_t incl = hv;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t woff = 0, n_mine = 0;
    for (int w = 0; w < 16; w++) { woff += w < wave ? wsum[w] : 0u; n_mine += wsum[w]; }
    hist[t] = woff + incl - hv;
    __syncthreads();
    // ---- prefixes of this workgroup's eighth, in index order: 1024 items per round (wave scans + a scan over the 16 wave totals)
    auto scan32 = [](uint32_t v) {
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
        return v;
    };
    for (int i0 = p0; i0 < p1; i0 += 1024) {
        const int i = i0 + t;
        const unsigned long long v = both(i < p1 ? counts[i] : 0u);
        const unsigned long long inc = (unsigned long long)scan32((uint32_t)v) | ((unsigned long long)scan32((uint32_t)(v >> 32)) << 32);
        __syncthreads();   // (wsum2 of the previous round has been read)
        if (lane == 63) wsum2[wave] = inc;
        __syncthreads();
        unsigned long long wo = 0, all = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) { const unsigned long long x = wsum2[w]; wo += w < wave ? x : 0ull; all += x; }
        if (i < p1) {
            const unsigned long long ex = run + wo + inc - v;
            if (prefix) prefix[i] = (uint32_t)ex;
            if (slot_prefix) slot_prefix[i] = (uint32_t)(ex >> 32);
        }
        run += all;
    }
    // (the workgroup that owns the last eighth has now summed every item: it publishes the totals; an eighth may be empty: p0 == p1 == n)
    if (c == 7 && t == 0) {
        if (total) {
            total[1] = (uint32_t)run; total[2] = (uint32_t)(run >> 32);
            total[3] = magic; total[4] = cap_R; total[5] = (uint32_t)(unsigned long long)cap_slots; total[6] = (uint32_t)((unsigned long long)cap_slots >> 32);
        }
        if (host_total) {
            __hip_atomic_store(host_total + 2, ((unsigned long long)host_tag << 32) | (unsigned long long)((uint32_t)n - n_empty), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(host_total, ((unsigned long long)host_tag << 32) | (run & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(host_total + 1, ((unsigned long long)host_tag << 32) | (run >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    // ---- pass 2: this XCD's items to order[8 j + c] (equal sizes in arrival order), then the padding behind the list
    for (int i = t; i < n; i += 1024) {
        const int tile = i >> 2;
        if (xcd_of_tile(tile % gx, tile / gx) != c) continue;
        const uint32_t len = counts[i];
        order[8u * atomicAdd(&hist[1023u - min(1023u, len)], 1u) + c] = (uint32_t)i;
    }
    for (uint32_t j = n_mine + (uint32_t)t; 8u * j + c < (uint32_t)order_n; j += 1024u) order[8u * j + c] = ORDER_NONE;
}

}  // namespace

void launch_order_desc(const uint32_t* counts, int n, uint32_t* order, uint32_t* prefix, uint32_t* slot_prefix, uint32_t* totals,
                       unsigned long long* host_totals, uint32_t host_tag, uint32_t cap_R, long long cap_slots, uint32_t magic, int gx, int order_n,
                       bool per_xcd, hipStream_t s) {
    if (per_xcd) hipLaunchKernelGGL(order_xcd_kernel, dim3(8), dim3(1024), 0, s, counts, n, order, prefix, slot_prefix, totals, host_totals,
                                    host_tag, cap_R, cap_slots, magic, gx, order_n);
    else hipLaunchKernelGGL(order_desc_kernel, dim3(1), dim3(1024), 0, s, counts, n, order, prefix, slot_prefix, totals, host_totals,
                            host_tag, cap_R, cap_slots, magic, order_n);
}

template <int ITEMS>
static void radix_sort_impl(uint32_t* const key[2], uint32_t* const val[2], int n, const uint32_t* n_dev, int total_bits,
                            int bits_per_pass, uint32_t* tbl, uint32_t* gtot, hipStream_t s) {
    const int per = BLOCK * ITEMS;
    const int nb = (n + per - 1) / per;
    const int ng = (nb + GS - 1) / GS;
    const int passes = (total_bits + bits_per_pass - 1) / bits_per_pass;
    for (int p = 0; p < passes; p++) {
        const int lo = p * bits_per_pass, nbits = std::min(bits_per_pass, total_bits - lo);
        uint32_t* gt = gtot + (size_t)p * ng * 256;
        hipLaunchKernelGGL((radix_hist_kernel<ITEMS>), dim3(nb), dim3(BLOCK), 0, s, key[p & 1], n, n_dev, lo,
                           (1u << nbits) - 1, tbl, gt);
        hipLaunchKernelGGL((radix_scatter_kernel<ITEMS>), dim3(nb), dim3(BLOCK), 0, s, key[p & 1], val[p & 1],
                           key[(p + 1) & 1], val[(p + 1) & 1], n, n_dev, lo, nbits, tbl, gt, ng);
    }
}

// Stable LSD radix sort of (u32 key, u32 value) pairs on bits [0, total_bits) in passes of bits_per_pass (<= 8);
// the result lands in slot (passes & 1) of the ping/pong buffers.  `n` sizes the launch and the scratch
// (`table`: radix_table_words(n) counters, its radix_gtot() part zeroed by the caller); if `n_dev` is not null the element count is min(n, *n_dev), read on the
// device (the count need not be known on the host at launch time).
void launch_radix_sort(uint32_t* const key[2], uint32_t* const val[2], int n, const uint32_t* n_dev, int total_bits,
                       int bits_per_pass, uint32_t* table, hipStream_t s) {
    if (n <= 0) return;
    uint32_t* gtot = radix_gtot(table, n);  // [passes <= 4][groups][256] group digit totals, zero on entry
    if (n <= (1 << 20)) radix_sort_impl<4>(key, val, n, n_dev, total_bits, bits_per_pass, table, gtot, s);
    else radix_sort_impl<16>(key, val, n, n_dev, total_bits, bits_per_pass, table, gtot, s);
}

void launch_offsets_scan(const uint32_t* tiles, const uint32_t* order, uint32_t* offsets, uint32_t* scan_tmp, int n,
                         uint32_t* total_out, const uint32_t* key_top, int n_key_top, const uint32_t* violation,
                         unsigned long long* host_out, uint32_t host_tag, hipStream_t s) {
    const int nb = scan_blocks(n);
    uint32_t* block_key = scan_tmp + nb + 1;
    hipLaunchKernelGGL(offsets_reduce_kernel, dim3(nb), dim3(BLOCK), 0, s, tiles, order, n, offsets, scan_tmp, key_top, n_key_top, block_key);
    hipLaunchKernelGGL(offsets_write_kernel, dim3(nb), dim3(BLOCK), 0, s, n, scan_tmp, offsets, nb,
                       total_out, block_key, violation, host_out, host_tag);
}

void launch_emit(int P, const uint32_t* order, const uint32_t* tiles, const uint32_t* offsets, float* rec,
                 const int32_t* radii, int gx, int gy, uint32_t* tile_keys, uint32_t* vals, int cap, uint32_t* ranges,
                 uint32_t* seg_count, uint32_t* sort_table, const uint32_t* span, hipStream_t s) {
    hipLaunchKernelGGL(emit_kernel, dim3((P + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, P, order, tiles, offsets, rec,
                       radii, gx, gy, tile_keys, vals, (uint32_t)cap, ranges, 2 * gx * gy + (gx * gy + 255) / 256 * SEG_BLOCK_STRIDE, seg_count,
                       radix_gtot(sort_table, cap), (int)radix_gtot_words(cap), span);
}

void launch_tile_sort12(uint32_t* const key[2], uint32_t* const val[2], int n, const uint32_t* n_dev, uint32_t* table, uint32_t* ranges, int T,
                        hipStream_t s) {
    if (n <= 0) return;
    const int nb = (n + TS12_KEYS - 1) / TS12_KEYS;
    const int nbins = (T + BLOCK - 1) / BLOCK * BLOCK;   // (<= TS12_BINS: tile_sort_plan)
    uint32_t* totals = table + (size_t)nb * nbins;
    hipLaunchKernelGGL(ts12_hist_kernel, dim3(nb), dim3(BLOCK), 0, s, key[0], n, n_dev, table, nbins);
    hipLaunchKernelGGL(ts12_colscan_kernel, dim3(nbins / 16), dim3(BLOCK), 0, s, table, nb, totals, nbins);
    hipLaunchKernelGGL(ts12_scatter_kernel, dim3(nb), dim3(BLOCK), 0, s, key[0], val[0], key[1], val[1], n, n_dev, table, totals, ranges, T, nbins);
}

// `ranges` must already be zero (launch_emit clears it)
void launch_ranges(int R, const uint32_t* R_dev, const uint32_t* tile_keys, uint32_t* ranges, int T, hipStream_t s) {
    if (R > 0) hipLaunchKernelGGL(ranges_kernel, dim3((R + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, R, R_dev, tile_keys, ranges);
}

}  // namespace svgir
