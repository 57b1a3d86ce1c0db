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
#include <algorithm>

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
    float nan_value[ADAM_MAX]; int flags[ADAM_MAX];   // SVGIR_ADAM_SCRUB_NAN / SVGIR_ADAM_ZERO_GRAD
    float* gw[ADAM_MAX];                              // writable alias of g when a flag needs it
    int count;
};

__global__ void __launch_bounds__(BLOCK) adam_kernel(const AdamTable t, float w1, float b2, float w2, float eps) {
#pragma clang fp contract(off)
    int k = 0;
    while (k + 1 < t.count && (int)blockIdx.x >= t.first_chunk[k + 1]) k++;   // (<= 32 steps, scalar)
    const long long base = (long long)((int)blockIdx.x - t.first_chunk[k]) * ADAM_CHUNK;
    float* __restrict__ p = t.p[k];
    const float* g = t.g[k];   // (may alias gw)
    float* __restrict__ m = t.m[k];
    float* __restrict__ v = t.v[k];
    const long long n = t.n[k];
    const float nss = t.neg_step_size[k], bs = t.bc2_sqrt[k];
    const int flags = t.flags[k];
    const float nanv = t.nan_value[k];
    float* gw = t.gw[k];
    for (int j = threadIdx.x; j < ADAM_CHUNK; j += BLOCK) {
        const long long i = base + j;
        if (i >= n) break;
        float gi = g[i];
        // GaussianModel.step(): replace_nangrad_to_zero (NaN gradient entries -> the group's replacement value, in place),
        // optimizer.step(), optimizer.zero_grad() -- folded into the one pass over the gradient
        if ((flags & SVGIR_ADAM_SCRUB_NAN) && gi != gi) gi = nanv;
        if (flags & SVGIR_ADAM_ZERO_GRAD) gw[i] = 0.f;
        else if (flags & SVGIR_ADAM_SCRUB_NAN) gw[i] = gi;
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
    // grid.y = tensor (no per-thread table search); a thread copies one 4-byte word of one row
    const uint32_t rows = min((uint32_t)rows_max, count_dev[0]);
    const int k = blockIdx.y;
    const int words = t.words[k];
    const long long r = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (r >= (long long)rows * words) return;
    const long long row = r / words;
    const int j = (int)(r - row * words);
    t.dst[k][r] = t.src[k][(size_t)kept[row] * words + j];
}

// ---- densification: selection masks, fused append, split transform (scene/gaussian_model.py:1136-1248) ---------------------
// get_scaling = nan_to_num(exp(_scaling), nan = 1e-6) (scene/gaussian_model.py:270-272): NaN -> 1e-6, +inf -> FLT_MAX
__device__ __forceinline__ float scaling_act(float raw) {
    const float e = expf(raw);
    return e != e ? 1e-6f : fminf(e, 3.402823466e+38f);
}
__global__ void __launch_bounds__(BLOCK) densify_masks_kernel(int P, const float* __restrict__ grad_accum, const float* __restrict__ normal_accum,
                                                              const float* __restrict__ denom, const float* __restrict__ scaling_raw,
                                                              float grad_threshold, float normal_threshold, float size_limit,
                                                              uint8_t* __restrict__ clone_mask, uint8_t* __restrict__ split_mask) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= P) return;
    // densify_and_prune: grads = accum / denom, NaN -> 0; norm over the last (size-1) axis = |.|
    float g = grad_accum[i] / denom[i], gn = normal_accum ? normal_accum[i] / denom[i] : 0.f;
    g = g != g ? 0.f : fabsf(g);
    gn = gn != gn ? 0.f : fabsf(gn);
    const bool sel = g >= grad_threshold || gn >= normal_threshold;
    // max of get_scaling over the three axes (NaN axes count as 1e-6, like the reference's nan_to_num)
    const float s = fmaxf(fmaxf(scaling_act(scaling_raw[3 * i]), scaling_act(scaling_raw[3 * i + 1])), scaling_act(scaling_raw[3 * i + 2]));
    clone_mask[i] = sel && s <= size_limit;
    split_mask[i] = sel && s > size_limit;
}

struct AppendTable {
    const uint32_t* src[ADAM_MAX]; uint32_t* dst[ADAM_MAX];
    int words[ADAM_MAX]; int flags[ADAM_MAX];
    long long first_word[ADAM_MAX + 1];
    int count;
};
// dst = cat(src[0 : rows_old], repeat(src[list], repeat))  (new rows zero for SVGIR_APPEND_ZERO_NEW: the Adam moments of
// cat_tensors_to_optimizer); one launch for every parameter, both moments and the bookkeeping arrays
__global__ void __launch_bounds__(BLOCK) append_rows_kernel(const AppendTable t, const uint32_t* __restrict__ list,
                                                            const uint32_t* __restrict__ count_dev, long long rows_old, long long n_sel_max,
                                                            int repeat) {
    const long long n_sel = n_sel_max > 0 ? min((long long)count_dev[0], n_sel_max) : 0;
    const int k = blockIdx.y;   // grid.y = tensor
    const int words = t.words[k];
    const long long r = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (r >= (rows_old + n_sel_max * repeat) * words) return;
    const long long row = r / words;
    const int j = (int)(r - row * words);
    if (row < rows_old) { t.dst[k][r] = t.src[k][r]; return; }
    const long long q = row - rows_old;
    if (q >= n_sel * repeat) return;
    t.dst[k][r] = (t.flags[k] & SVGIR_APPEND_ZERO_NEW) ? 0u : t.src[k][(size_t)list[q % n_sel] * words + j];
}

// densify_and_split on the freshly appended copies (rows [rows_old, rows_old + n_new)): xyz <- R(q) (z * exp(s)) + xyz,
// scaling <- log(exp(s) / (0.8 N)) with the last axis at -1e10 (scene/gaussian_model.py:1152-1161); z = standard normal
// draws supplied by the caller ([n_new, 3], torch.normal(mean = 0, std = stds) = stds * z)
__global__ void __launch_bounds__(BLOCK) split_transform_kernel(long long n_new, int N, const float* __restrict__ z,
                                                                float* __restrict__ xyz, float* __restrict__ scaling,
                                                                const float* __restrict__ rotation) {
#pragma clang fp contract(off)
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n_new) return;
    const float s0 = scaling_act(scaling[3 * i]), s1 = scaling_act(scaling[3 * i + 1]), s2 = scaling_act(scaling[3 * i + 2]);   // get_scaling
    const float v[3] = {z[3 * i] * s0, z[3 * i + 1] * s1, z[3 * i + 2] * s2};
    float q0 = rotation[4 * i], q1 = rotation[4 * i + 1], q2 = rotation[4 * i + 2], q3 = rotation[4 * i + 3];
    const float nrm = sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);   // build_rotation (utils/general_utils.py:82-103)
    q0 /= nrm; q1 /= nrm; q2 /= nrm; q3 /= nrm;
    const float r = q0, x = q1, y = q2, w = q3;
    const float R[3][3] = {{1 - 2 * (y * y + w * w), 2 * (x * y - r * w), 2 * (x * w + r * y)},
                           {2 * (x * y + r * w), 1 - 2 * (x * x + w * w), 2 * (y * w - r * x)},
                           {2 * (x * w - r * y), 2 * (y * w + r * x), 1 - 2 * (x * x + y * y)}};
#pragma unroll
    for (int c = 0; c < 3; c++) xyz[3 * i + c] = ((R[c][0] * v[0] + R[c][1] * v[1]) + R[c][2] * v[2]) + xyz[3 * i + c];
    const float k = 0.8f * (float)N;
    scaling[3 * i] = logf(s0 / k); scaling[3 * i + 1] = logf(s1 / k); scaling[3 * i + 2] = -1e10f;
}

}  // namespace

}  // namespace svgir

extern "C" {

int svgir_densify_masks(int32_t P, const float* xyz_gradient_accum, const float* normal_gradient_accum, const float* denom,
                        const float* scaling_raw, float grad_threshold, float normal_threshold, float size_limit,
                        uint8_t* clone_mask, uint8_t* split_mask, void* stream) {
    using namespace svgir;
    if (P < 0 || (P > 0 && (!xyz_gradient_accum || !denom || !scaling_raw || !clone_mask || !split_mask))) return SVGIR_ERR_INVALID;
    if (P == 0) return SVGIR_OK;
    hipLaunchKernelGGL(densify_masks_kernel, dim3((P + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, (hipStream_t)stream, P, xyz_gradient_accum,
                       normal_gradient_accum, denom, scaling_raw, grad_threshold, normal_threshold, size_limit, clone_mask, split_mask);
    return hipGetLastError() == hipSuccess ? SVGIR_OK : SVGIR_ERR_HIP;
}

int svgir_append_rows(const svgir_append_tensor* tensors, int32_t count, int64_t rows_old, const uint32_t* list,
                      const uint32_t* count_dev, int64_t n_sel_max, int32_t repeat, void* stream) {
    using namespace svgir;
    if (count < 0 || count > ADAM_MAX || rows_old < 0 || n_sel_max < 0 || repeat < 1 || (count > 0 && !tensors) ||
        (n_sel_max > 0 && (!list || !count_dev)))
        return SVGIR_ERR_INVALID;
    if (count == 0) return SVGIR_OK;
    AppendTable t;
    t.count = 0;
    long long words = 0;
    const long long rows_new = rows_old + n_sel_max * repeat;
    for (int i = 0; i < count; i++) {
        const svgir_append_tensor& a = tensors[i];
        if (a.row_bytes == 0) continue;
        if (!a.dst || (!a.src && rows_old > 0) || a.row_bytes % 4 != 0) return SVGIR_ERR_INVALID;
        const int k = t.count++;
        t.src[k] = (const uint32_t*)a.src; t.dst[k] = (uint32_t*)a.dst; t.words[k] = a.row_bytes / 4; t.flags[k] = a.flags;
        t.first_word[k] = words;
        words += rows_new * t.words[k];
    }
    if (t.count == 0 || words == 0) return SVGIR_OK;
    t.first_word[t.count] = words;
    int wmax = 0;
    for (int k = 0; k < t.count; k++) wmax = std::max(wmax, t.words[k]);
    hipLaunchKernelGGL(append_rows_kernel, dim3((unsigned)((rows_new * wmax + BLOCK - 1) / BLOCK), (unsigned)t.count), dim3(BLOCK), 0,
                       (hipStream_t)stream, t, list, count_dev, (long long)rows_old, (long long)n_sel_max, repeat);
    return hipGetLastError() == hipSuccess ? SVGIR_OK : SVGIR_ERR_HIP;
}

int svgir_split_transform(int64_t n_new, int32_t N, const float* z, float* xyz_new, float* scaling_new, const float* rotation_new,
                          void* stream) {
    using namespace svgir;
    if (n_new < 0 || N < 1 || (n_new > 0 && (!z || !xyz_new || !scaling_new || !rotation_new))) return SVGIR_ERR_INVALID;
    if (n_new == 0) return SVGIR_OK;
    hipLaunchKernelGGL(split_transform_kernel, dim3((unsigned)((n_new + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream,
                       (long long)n_new, N, z, xyz_new, scaling_new, rotation_new);
    return hipGetLastError() == hipSuccess ? SVGIR_OK : SVGIR_ERR_HIP;
}


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
        t.flags[k] = a.flags; t.nan_value[k] = a.nan_value; t.gw[k] = const_cast<float*>(a.grad);
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
    int wmax = 0;
    for (int k = 0; k < t.count; k++) wmax = std::max(wmax, t.words[k]);
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)(((long long)rows_max * wmax + BLOCK - 1) / BLOCK), (unsigned)t.count), dim3(BLOCK),
                       0, (hipStream_t)stream, t, kept, count_dev, (long long)rows_max);
    return hipGetLastError() == hipSuccess ? SVGIR_OK : SVGIR_ERR_HIP;
}

}  // extern "C"
