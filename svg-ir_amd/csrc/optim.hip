// svg-ir_amd/csrc/optim.hip -- the consumer side of the rasterizer's gradients: the Adam step over the
// per-Gaussian parameter block, the densification statistics, and the row compaction behind pruning.
//
// Replaces, for the parameter groups the reference registers in GaussianModel.training_setup
// (scene/gaussian_model.py:737-773: torch.optim.Adam over 7 (+6 with PBR) tensors, per-group learning rates, eps = 1e-15),
//   * `optimizer.step()`            -> ONE multi-tensor kernel (torch's foreach Adam: ~10 launches per step),
//   * `add_densification_stats`     (scene/gaussian_model.py:1270-1276) -> one kernel,
//   * the boolean-mask indexing of `_prune_optimizer` / `prune_points` (scene/gaussian_model.py:1020-1062: `t[mask]` for every
//     parameter, both Adam moments and five bookkeeping arrays, ~45 index kernels each with its own mask scan) -> one scan of the
//     mask + one multi-tensor row gather.
// Arithmetic of the Adam step = torch.optim.Adam (amsgrad=False, weight_decay=0, maximize=False), single-tensor form:
//   m <- m + (1 - b1) (g - m);  v <- v b2 + (1 - b2) g g;  p <- p - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// with the bias corrections evaluated on the host in double precision like torch does.
#include "common.hpp"

namespace svgir {

namespace {

constexpr int ADAM_MAX = SVGIR_ADAM_MAX_TENSORS;
constexpr int ADAM_CHUNK = 4096;   // elements per workgroup

struct AdamTable {
    float* p[ADAM_MAX]; const float* g[ADAM_MAX]; float* m[ADAM_MAX]; float* v[ADAM_MAX];
    long long n[ADAM_MAX];
    int first_chunk[ADAM_MAX + 1];
    float neg_step_size[ADAM_MAX], bc2_sqrt[ADAM_MAX];
    int count;
};

__global__ void __launch_bounds__(BLOCK) adam_kernel(const AdamTable t, float w1, float b2, float w2, float eps) {
#pragma clang fp contract(off)
    int k = 0;
    while (k + 1 < t.count && (int)blockIdx.x >= t.first_chunk[k + 1]) k++;   // (<= 32 steps, scalar)
    const long long base = (long long)((int)blockIdx.x - t.first_chunk[k]) * ADAM_CHUNK;
    float* __restrict__ p = t.p[k];
    const float* __restrict__ g = t.g[k];
    float* __restrict__ m = t.m[k];
    float* __restrict__ v = t.v[k];
    const long long n = t.n[k];
    const float nss = t.neg_step_size[k], bs = t.bc2_sqrt[k];
    for (int j = threadIdx.x; j < ADAM_CHUNK; j += BLOCK) {
        const long long i = base + j;
        if (i >= n) break;
        const float gi = g[i];
        float mi = m[i], vi = v[i];
        mi = mi + w1 * (gi - mi);              // exp_avg.lerp_(grad, 1 - beta1)
        vi = vi * b2;                          // exp_avg_sq.mul_(beta2)
        vi = vi + (w2 * gi) * gi;              //            .addcmul_(grad, grad, value = 1 - beta2)
        const float denom = sqrtf(vi) / bs + eps;   // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
        m[i] = mi; v[i] = vi;
        p[i] = p[i] + nss * (mi / denom);      // param.addcdiv_(exp_avg, denom, value = -step_size)
    }
}

__global__ void __launch_bounds__(BLOCK) densify_stats_kernel(int P, const float* __restrict__ vgrad, int vstride,
                                                              const uint8_t* __restrict__ filter, const float* __restrict__ weights,
                                                              float* __restrict__ weights_accum, float* __restrict__ grad_accum,
                                                              float* __restrict__ denom) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= P) return;
    if (weights) weights_accum[i] += weights[i];
    if (filter[i]) {
        const float gx = vgrad[(size_t)i * vstride], gy = vgrad[(size_t)i * vstride + 1];
        grad_accum[i] += sqrtf(gx * gx + gy * gy);   // torch.norm(grad[:, :2], dim=-1)
        denom[i] += 1.f;
    }
}

// ---- mask -> list of kept rows (order preserving) ----------------------------------------------------------------
constexpr int SCAN_ELEMS = BLOCK * 8;
__global__ void __launch_bounds__(BLOCK) mask_count_kernel(const uint8_t* __restrict__ keep, int P, uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t wsum[BLOCK / 64];
    const int base = blockIdx.x * SCAN_ELEMS + threadIdx.x * 8;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) c += (base + i < P && keep[base + i]) ? 1u : 0u;
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
__global__ void __launch_bounds__(BLOCK) mask_scatter_kernel(const uint8_t* __restrict__ keep, int P, const uint32_t* __restrict__ block_sums,
                                                             int nblocks, uint32_t* __restrict__ kept, uint32_t* __restrict__ count_out) {
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
    const int base = blockIdx.x * SCAN_ELEMS + t * 8;
    uint32_t k[8], c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { k[i] = (base + i < P && keep[base + i]) ? 1u : 0u; c += k[i]; }
    uint32_t incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= d) incl += o;
    }
    __syncthreads();
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; w++) woff += wsum[w];
    uint32_t pos = before_s + woff + incl - c;
#pragma unroll
    for (int i = 0; i < 8; i++)
        if (k[i]) kept[pos++] = (uint32_t)(base + i);
    if ((int)blockIdx.x == nblocks - 1 && t == BLOCK - 1) count_out[0] = pos;
}

struct GatherTable {
    const uint32_t* src[ADAM_MAX]; uint32_t* dst[ADAM_MAX];
    int words[ADAM_MAX];              // 4-byte words per row
    long long first_word[ADAM_MAX + 1];   // prefix of count_max * words
    int count;
};
__global__ void __launch_bounds__(BLOCK) gather_rows_kernel(const GatherTable t, const uint32_t* __restrict__ kept,
                                                            const uint32_t* __restrict__ count_dev, long long rows_max) {
    const uint32_t rows = min((uint32_t)rows_max, count_dev[0]);
    const long long w = (long long)blockIdx.x * BLOCK + threadIdx.x;
    int k = 0;
    while (k + 1 < t.count && w >= t.first_word[k + 1]) k++;
    if (w >= t.first_word[t.count]) return;
    const long long r = w - t.first_word[k];
    const int words = t.words[k];
    const long long row = r / words;
    const int j = (int)(r - row * words);
    if (row >= (long long)rows) return;
    t.dst[k][row * words + j] = t.src[k][(size_t)kept[row] * words + j];
}

}  // namespace

}  // namespace svgir

extern "C" {

int svgir_adam_step(const svgir_adam_tensor* tensors, int32_t count, double beta1, double beta2, double eps, void* stream) {
    using namespace svgir;
    if (count < 0 || count > ADAM_MAX || (count > 0 && !tensors)) return SVGIR_ERR_INVALID;
    AdamTable t;
    t.count = 0;
    int chunks = 0;
    for (int i = 0; i < count; i++) {
        const svgir_adam_tensor& a = tensors[i];
        if (a.n <= 0) continue;
        if (!a.param || !a.grad || !a.exp_avg || !a.exp_avg_sq || a.step < 1) return SVGIR_ERR_INVALID;
        const int k = t.count++;
        t.p[k] = a.param; t.g[k] = a.grad; t.m[k] = a.exp_avg; t.v[k] = a.exp_avg_sq; t.n[k] = a.n;
        t.first_chunk[k] = chunks;
        chunks += (int)((a.n + ADAM_CHUNK - 1) / ADAM_CHUNK);
        // torch/optim/adam.py _single_tensor_adam: python-float (double) bias corrections
        const double bc1 = 1.0 - pow(beta1, (double)a.step), bc2 = 1.0 - pow(beta2, (double)a.step);
        t.neg_step_size[k] = (float)(-((double)a.lr / bc1));
        t.bc2_sqrt[k] = (float)sqrt(bc2);
    }
    if (t.count == 0) return SVGIR_OK;
    t.first_chunk[t.count] = chunks;
    // scalars are rounded to fp32 once, from the double values torch holds (1 - beta2 taken in double: 1 - fp32(0.999) would
    // be off by 1.3e-5 relative)
    hipLaunchKernelGGL(adam_kernel, dim3(chunks), dim3(BLOCK), 0, (hipStream_t)stream, t, (float)(1.0 - beta1), (float)beta2,
                       (float)(1.0 - beta2), (float)eps);
    return hipGetLastError() == hipSuccess ? SVGIR_OK : SVGIR_ERR_HIP;
}

int svgir_densify_stats(int32_t P, const float* viewspace_grad, int32_t grad_stride, const uint8_t* update_filter,
                        const float* weights, float* weights_accum, float* xyz_gradient_accum, float* denom, void* stream) {
    using namespace svgir;
    if (P < 0 || (P > 0 && (!viewspace_grad || !update_filter || !xyz_gradient_accum || !denom || grad_stride < 2)) ||
        (weights && !weights_accum))
        return SVGIR_ERR_INVALID;
    if (P == 0) return SVGIR_OK;
    hipLaunchKernelGGL(densify_stats_kernel, dim3((P + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, (hipStream_t)stream, P, viewspace_grad,
                       grad_stride, update_filter, weights, weights_accum, xyz_gradient_accum, denom);
    return hipGetLastError() == hipSuccess ? SVGIR_OK : SVGIR_ERR_HIP;
}

size_t svgir_mask_scan_work_words(int32_t P) { return (size_t)((P > 0 ? P : 1) + svgir::SCAN_ELEMS - 1) / svgir::SCAN_ELEMS + 1; }

int svgir_mask_scan(int32_t P, const uint8_t* keep, uint32_t* kept, uint32_t* work, uint32_t* count_dev, void* stream) {
    using namespace svgir;
    if (P < 0 || !count_dev || (P > 0 && (!keep || !kept || !work))) return SVGIR_ERR_INVALID;
    if (P == 0) return hipMemsetAsync(count_dev, 0, 4, (hipStream_t)stream) == hipSuccess ? SVGIR_OK : SVGIR_ERR_HIP;
    const int nb = (P + SCAN_ELEMS - 1) / SCAN_ELEMS;
    hipLaunchKernelGGL(mask_count_kernel, dim3(nb), dim3(BLOCK), 0, (hipStream_t)stream, keep, P, work);
    hipLaunchKernelGGL(mask_scatter_kernel, dim3(nb), dim3(BLOCK), 0, (hipStream_t)stream, keep, P, work, nb, kept, count_dev);
    return hipGetLastError() == hipSuccess ? SVGIR_OK : SVGIR_ERR_HIP;
}

int svgir_gather_rows(const svgir_row_tensor* tensors, int32_t count, const uint32_t* kept, const uint32_t* count_dev,
                      int32_t rows_max, void* stream) {
    using namespace svgir;
    if (count < 0 || count > ADAM_MAX || rows_max < 0 || (count > 0 && (!tensors || !kept || !count_dev))) return SVGIR_ERR_INVALID;
    if (count == 0 || rows_max == 0) return SVGIR_OK;
    GatherTable t;
    t.count = 0;
    long long words = 0;
    for (int i = 0; i < count; i++) {
        const svgir_row_tensor& a = tensors[i];
        if (a.row_bytes == 0) continue;
        if (!a.src || !a.dst || a.row_bytes % 4 != 0) return SVGIR_ERR_INVALID;
        const int k = t.count++;
        t.src[k] = (const uint32_t*)a.src; t.dst[k] = (uint32_t*)a.dst; t.words[k] = a.row_bytes / 4;
        t.first_word[k] = words;
        words += (long long)rows_max * t.words[k];
    }
    if (t.count == 0) return SVGIR_OK;
    t.first_word[t.count] = words;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((words + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream, t, kept,
                       count_dev, (long long)rows_max);
    return hipGetLastError() == hipSuccess ? SVGIR_OK : SVGIR_ERR_HIP;
}

}  // extern "C"
