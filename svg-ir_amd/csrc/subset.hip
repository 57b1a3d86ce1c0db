// svg-ir_amd/csrc/subset.hip -- "shade only what the rasterizer reads": selection of a view's working set of surfels.
//
// The reference shades all P surfels of the model for every view (gaussian_renderer/svgss.py:116-141: rendering_equation4 over every
// point, chunked by 100 k) although the only consumer of the result is the rasterizer call behind it (svgss.py:143-182), which reads the
// packed rows of the surfels that survive its culls (svgss forward.cu:267-395 per Gaussian, and the alpha threshold per pixel).  Here the
// working set of a view is made explicit as a PARTITION of 0..P-1: list[0 .. n) = the selected surfels in index order, list[P-1-j] = the
// j-th unselected one; the shading kernels walk the front (csrc/shade.hip, svgir_shade_params.subset), a small kernel zero-fills the
// output rows of the back, so every output is still written completely.
//   forward : selected <=> the surfel is a candidate of at least one 8x8 sub-tile (flag byte set by cull_kernel);
//   backward: selected <=> out_weights > 0 (every other surfel has exactly-zero dL_dfeatures / dL_dvfeatures rows).
// Two launches, like the mask scan of csrc/optim.hip: per-block counts, then every block sums the counts in front of it and scatters.
#include <algorithm>

#include "common.hpp"

namespace svgir {

namespace {


template <bool FLAGS>
__device__ __forceinline__ bool part_pred(const uint8_t* __restrict__ flags, const float* __restrict__ values, int i) {
    return FLAGS ? flags[i] != 0 : values[i] > 0.f;
}

template <bool FLAGS>
__global__ void __launch_bounds__(BLOCK) part_count_kernel(const uint8_t* __restrict__ flags, const float* __restrict__ values, int P,
                                                           uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t wsum[BLOCK / 64];
    const int base = blockIdx.x * PART_ELEMS + threadIdx.x * 8;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) c += (base + i < P && part_pred<FLAGS>(flags, values, base + i)) ? 1u : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += (uint32_t)__shfl_xor((int)c, d);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t s = 0;
        for (int w = 0; w < BLOCK / 64; w++) s += wsum[w];
        block_sums[blockIdx.x] = s;
    }
}

template <bool FLAGS>
__global__ void __launch_bounds__(BLOCK) part_scatter_kernel(const uint8_t* __restrict__ flags, const float* __restrict__ values, int P,
                                                             const uint32_t* __restrict__ block_sums, int nblocks,
                                                             uint32_t* __restrict__ list, uint32_t* __restrict__ count_out) {
    __shared__ uint32_t wsum[BLOCK / 64];
    __shared__ uint32_t before_s;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    uint32_t pre = 0;
    for (int b = t; b < (int)blockIdx.x; b += BLOCK) pre += block_sums[b];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) pre += (uint32_t)__shfl_xor((int)pre, d);
    if (lane == 0) wsum[wave] = pre;
    __syncthreads();
    if (t == 0) { uint32_t s = 0; for (int w = 0; w < BLOCK / 64; w++) s += wsum[w]; before_s = s; }
    __syncthreads();
    const int base = blockIdx.x * PART_ELEMS + t * 8;
    uint32_t k[8], c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { k[i] = (base + i < P && part_pred<FLAGS>(flags, values, base + i)) ? 1u : 0u; c += k[i]; }
    uint32_t incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= d) incl += o;
    }
    const uint32_t before = before_s;
    __syncthreads();
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; w++) woff += wsum[w];
    uint32_t pos = before + woff + incl - c;            // selected surfels in front of this thread's first
    uint32_t npos = (uint32_t)base - pos;               // unselected ones in front of it
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (base + i >= P) break;
        if (k[i]) list[pos++] = (uint32_t)(base + i);
        else list[(uint32_t)P - 1u - npos++] = (uint32_t)(base + i);
    }
    if ((int)blockIdx.x == nblocks - 1 && t == BLOCK - 1) count_out[0] = pos;
}

struct ZeroRows {
    float* ptr[6]; int row_floats[6]; int n;
};
// one wave per unselected surfel (grid-stride): its row in each of the `n` tensors is zeroed with consecutive lanes
__global__ void __launch_bounds__(BLOCK) zero_rows_kernel(const uint32_t* __restrict__ list, const uint32_t* __restrict__ count, int P,
                                                          const ZeroRows z) {
    const int lane = threadIdx.x & 63;
    const uint32_t nsel = min(*count, (uint32_t)P);
    const uint32_t nrest = (uint32_t)P - nsel;
    const uint32_t waves = gridDim.x * (BLOCK / 64);
    for (uint32_t j = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6); j < nrest; j += waves) {
        const size_t g = list[(uint32_t)P - 1u - j];
        for (int t = 0; t < z.n; t++) {
            float* row = z.ptr[t] + g * (size_t)z.row_floats[t];
            for (int e = lane; e < z.row_floats[t]; e += 64) row[e] = 0.f;
        }
    }
}

}  // namespace

size_t partition_work_words(int P) { return (size_t)((P > 0 ? P : 1) + PART_ELEMS - 1) / PART_ELEMS + 2; }

void launch_partition(int P, const uint8_t* flags, const float* positive, uint32_t* list, uint32_t* work, uint32_t* count_dev, hipStream_t s) {
    if (P <= 0) return;
    const int nb = (P + PART_ELEMS - 1) / PART_ELEMS;
    if (flags) {
        hipLaunchKernelGGL(part_count_kernel<true>, dim3(nb), dim3(BLOCK), 0, s, flags, positive, P, work);
        hipLaunchKernelGGL(part_scatter_kernel<true>, dim3(nb), dim3(BLOCK), 0, s, flags, positive, P, work, nb, list, count_dev);
    } else {
        hipLaunchKernelGGL(part_count_kernel<false>, dim3(nb), dim3(BLOCK), 0, s, flags, positive, P, work);
        hipLaunchKernelGGL(part_scatter_kernel<false>, dim3(nb), dim3(BLOCK), 0, s, flags, positive, P, work, nb, list, count_dev);
    }
}

void launch_partition_scatter(int P, const float* positive, uint32_t* list, const uint32_t* work, uint32_t* count_dev, hipStream_t s) {
    if (P <= 0) return;
    const int nb = (P + PART_ELEMS - 1) / PART_ELEMS;
    hipLaunchKernelGGL(part_scatter_kernel<false>, dim3(nb), dim3(BLOCK), 0, s, (const uint8_t*)nullptr, positive, P, work, nb, list, count_dev);
}

void launch_zero_rows(int P, const uint32_t* list, const uint32_t* count_dev, float* const* tensors, const int* row_floats, int n, hipStream_t s) {
    ZeroRows z;
    z.n = 0;
    for (int i = 0; i < n && z.n < 6; i++)
        if (tensors[i] && row_floats[i] > 0) { z.ptr[z.n] = tensors[i]; z.row_floats[z.n] = row_floats[i]; z.n++; }
    if (P <= 0 || z.n == 0) return;
    const int blocks = std::min((P + BLOCK / 64 - 1) / (BLOCK / 64), 256 * 8);
    hipLaunchKernelGGL(zero_rows_kernel, dim3(blocks), dim3(BLOCK), 0, s, list, count_dev, P, z);
}

}  // namespace svgir
