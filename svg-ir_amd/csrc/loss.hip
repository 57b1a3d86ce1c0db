// svg-ir_amd/csrc/loss.hip -- image losses right behind the rasterizer: L1 and SSIM with the
// reference's 11 x 11 Gaussian window, forward and backward.
//
// Replaces `F.l1_loss(image, gt)` + `ssim(image, gt)` as the reference calls them after render_view
// (gaussian_renderer/svgss.py:281-289, render.py:150-151; utils/loss_utils.py:21-64: `gaussian(11, 1.5)`, zero-padded
// depthwise conv2d of img1, img2, img1^2, img2^2, img1 img2, the SSIM map with C1 = 0.01^2, C2 = 0.03^2, its mean).
// The reference runs 5 convolutions + ~15 element-wise kernels over [3,H,W] planes forward and their autograd adjoints
// backward (~40 launches, every intermediate map through HBM).  Here:
//   forward : one kernel, one 16 x 16 output tile per workgroup: both images are read once (26 x 26 tile with halo), the five
//             windowed moments are a separable pass through LDS, and the kernel emits per-workgroup partial sums of the SSIM
//             map and of |img1 - img2| plus -- for the backward -- the three partial-derivative maps of the SSIM map w.r.t. the
//             windowed moments of img1 (mu1, E[x^2], E[x y]);
//   backward: one kernel, the same tiling: dL/dimg1 = w * dmu1 + 2 img1 (w * de11) + img2 (w * de12) (the window is
//             symmetric and the padding is zero, so the adjoint of the convolution is the convolution) + the L1 sign term.
// Only img1 (the rendered image) gets a gradient; img2 is the ground truth.
#include "common.hpp"

namespace svgir {

namespace {

constexpr int LT = 16, LR = 5, LW = LT + 2 * LR;   // output tile, window radius, input tile with halo

struct SsimWindow { float g[11]; };
inline SsimWindow ssim_window() {
    // utils/loss_utils.py:21-23: fp32 tensor of exp(-(x - 5)^2 / (2 * 1.5^2)), divided by its fp32 sum
    SsimWindow w;
    float s = 0.f;
    for (int i = 0; i < 11; i++) { w.g[i] = (float)exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); s += w.g[i]; }
    for (int i = 0; i < 11; i++) w.g[i] /= s;
    return w;
}

__global__ void __launch_bounds__(LT * LT) ssim_fwd_kernel(const float* __restrict__ img1, const float* __restrict__ img2, int H, int W,
                                                           SsimWindow win, float* __restrict__ partial, float* __restrict__ dmaps) {
    __shared__ float sA[LW][LW + 1], sB[LW][LW + 1];
    __shared__ float sH[5][LW][LT + 1];
    __shared__ float sRed[2][4];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4, t = threadIdx.x;
    const int ch = blockIdx.z;
    const size_t N = (size_t)H * W;
    const float* a = img1 + ch * N;
    const float* b = img2 + ch * N;
    const int x0 = blockIdx.x * LT - LR, y0 = blockIdx.y * LT - LR;
    for (int i = t; i < LW * LW; i += LT * LT) {
        const int r = i / LW, c = i - r * LW;
        const int y = y0 + r, x = x0 + c;
        const bool in = y >= 0 && y < H && x >= 0 && x < W;   // zero padding (loss_utils.py:45: padding = window_size // 2)
        sA[r][c] = in ? a[(size_t)y * W + x] : 0.f;
        sB[r][c] = in ? b[(size_t)y * W + x] : 0.f;
    }
    __syncthreads();
    for (int i = t; i < LW * LT; i += LT * LT) {   // horizontal pass: rows of the halo tile, 16 output columns
        const int r = i / LT, c = i - r * LT;
        float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float u = sA[r][c + k], v = sB[r][c + k], g = win.g[k];
            m1 += g * u; m2 += g * v; e11 += g * (u * u); e22 += g * (v * v); e12 += g * (u * v);
        }
        sH[0][r][c] = m1; sH[1][r][c] = m2; sH[2][r][c] = e11; sH[3][r][c] = e22; sH[4][r][c] = e12;
    }
    __syncthreads();
    float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
    for (int k = 0; k < 11; k++) {
        const float g = win.g[k];
        mu1 += g * sH[0][ty + k][tx]; mu2 += g * sH[1][ty + k][tx];
        e11 += g * sH[2][ty + k][tx]; e22 += g * sH[3][ty + k][tx]; e12 += g * sH[4][ty + k][tx];
    }
    const int x = blockIdx.x * LT + tx, y = blockIdx.y * LT + ty;
    const bool valid = x < W && y < H;
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    const float s1 = e11 - mu1 * mu1, s2 = e22 - mu2 * mu2, s12 = e12 - mu1 * mu2;
    const float A1 = 2.f * mu1 * mu2 + C1, A2 = 2.f * s12 + C2, B1 = mu1 * mu1 + mu2 * mu2 + C1, B2 = s1 + s2 + C2;
    const float iB1 = 1.f / B1, iB2 = 1.f / B2;
    const float S = (A1 * A2) * (iB1 * iB2);
    float ssim = valid ? S : 0.f;
    float l1 = valid ? fabsf(sA[ty + LR][tx + LR] - sB[ty + LR][tx + LR]) : 0.f;
    if (dmaps && valid) {
        // d S / d(mu1, E[x^2], E[xy]) with sigma1 = E[x^2] - mu1^2, sigma12 = E[xy] - mu1 mu2
        const float de11 = -S * iB2;
        const float de12 = 2.f * A1 * (iB1 * iB2);
        const float dmu1 = 2.f * mu2 * A2 * (iB1 * iB2) - S * 2.f * mu1 * iB1 - de12 * mu2 - de11 * 2.f * mu1;
        const size_t o = (size_t)ch * N + (size_t)y * W + x;
        const size_t CN = (size_t)gridDim.z * N;
        dmaps[o] = dmu1; dmaps[CN + o] = de11; dmaps[2 * CN + o] = de12;
    }
    // workgroup partial sums (fixed order: wave DPP sums, then the four waves in order)
    ssim = wave_sum(ssim); l1 = wave_sum(l1);
    if ((t & 63) == 0) { sRed[0][t >> 6] = ssim; sRed[1][t >> 6] = l1; }
    __syncthreads();
    if (t == 0) {
        const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partial[2 * blk] = (sRed[0][0] + sRed[0][1]) + (sRed[0][2] + sRed[0][3]);
        partial[2 * blk + 1] = (sRed[1][0] + sRed[1][1]) + (sRed[1][2] + sRed[1][3]);
    }
}

__global__ void __launch_bounds__(LT * LT) ssim_bwd_kernel(const float* __restrict__ img1, const float* __restrict__ img2,
                                                           const float* __restrict__ dmaps, int H, int W, SsimWindow win,
                                                           float g_ssim, float g_l1, const float* __restrict__ g_dev,
                                                           float* __restrict__ dL_dimg1) {
    if (g_dev) { g_ssim *= g_dev[0]; g_l1 *= g_dev[1]; }   // upstream scalars still on the device (no host read-back)
    __shared__ float sM[3][LW][LW + 1];
    __shared__ float sH[3][LW][LT + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4, t = threadIdx.x;
    const int ch = blockIdx.z;
    const size_t N = (size_t)H * W, CN = (size_t)gridDim.z * N;
    const int x0 = blockIdx.x * LT - LR, y0 = blockIdx.y * LT - LR;
    for (int i = t; i < LW * LW; i += LT * LT) {
        const int r = i / LW, c = i - r * LW;
        const int y = y0 + r, x = x0 + c;
        const bool in = y >= 0 && y < H && x >= 0 && x < W;
        const size_t o = (size_t)ch * N + (size_t)(in ? y : 0) * W + (in ? x : 0);
#pragma unroll
        for (int q = 0; q < 3; q++) sM[q][r][c] = in ? dmaps[q * CN + o] : 0.f;
    }
    __syncthreads();
    for (int i = t; i < LW * LT; i += LT * LT) {
        const int r = i / LT, c = i - r * LT;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) { const float g = win.g[k]; v0 += g * sM[0][r][c + k]; v1 += g * sM[1][r][c + k]; v2 += g * sM[2][r][c + k]; }
        sH[0][r][c] = v0; sH[1][r][c] = v1; sH[2][r][c] = v2;
    }
    __syncthreads();
    const int x = blockIdx.x * LT + tx, y = blockIdx.y * LT + ty;
    if (x >= W || y >= H) return;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int k = 0; k < 11; k++) { const float g = win.g[k]; c0 += g * sH[0][ty + k][tx]; c1 += g * sH[1][ty + k][tx]; c2 += g * sH[2][ty + k][tx]; }
    const size_t o = (size_t)ch * N + (size_t)y * W + x;
    const float u = img1[o], v = img2[o];
    const float d = u - v;
    const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);   // torch: sign(0) = 0
    dL_dimg1[o] = g_ssim * (c0 + 2.f * u * c1 + v * c2) + g_l1 * sgn;
}

// the two means from the per-tile partial sums: one workgroup, fixed order, double accumulation
__global__ void __launch_bounds__(256) ssim_reduce_kernel(const float* __restrict__ partial, int nblk, double inv, float* __restrict__ out2) {
    __shared__ double red[2][4];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) { a += (double)partial[2 * i]; b += (double)partial[2 * i + 1]; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d); b += __shfl_xor(b, d); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out2[0] = (float)(((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) * inv);
        out2[1] = (float)(((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) * inv);
    }
}

}  // namespace

}  // namespace svgir

extern "C" {

size_t svgir_l1_ssim_partials(int32_t C, int32_t H, int32_t W) {
    return (size_t)C * ((H + svgir::LT - 1) / svgir::LT) * ((W + svgir::LT - 1) / svgir::LT);
}

int svgir_l1_ssim_forward(const float* img1, const float* img2, int32_t C, int32_t H, int32_t W, float* partial, float* dmaps,
                          float* means2, void* stream) {
    if (C <= 0 || H <= 0 || W <= 0 || !img1 || !img2 || !partial) return SVGIR_ERR_INVALID;
    const dim3 grid((W + svgir::LT - 1) / svgir::LT, (H + svgir::LT - 1) / svgir::LT, C);
    hipLaunchKernelGGL(svgir::ssim_fwd_kernel, grid, dim3(svgir::LT * svgir::LT), 0, (hipStream_t)stream, img1, img2, H, W,
                       svgir::ssim_window(), partial, dmaps);
    if (means2)
        hipLaunchKernelGGL(svgir::ssim_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, (int)(grid.x * grid.y * grid.z),
                           1.0 / ((double)C * (double)H * (double)W), means2);
    return hipGetLastError() == hipSuccess ? SVGIR_OK : SVGIR_ERR_HIP;
}

int svgir_l1_ssim_backward(const float* img1, const float* img2, const float* dmaps, int32_t C, int32_t H, int32_t W,
                           float g_ssim_mean, float g_l1_mean, const float* g_dev, float* dL_dimg1, void* stream) {
    if (C <= 0 || H <= 0 || W <= 0 || !img1 || !img2 || !dmaps || !dL_dimg1) return SVGIR_ERR_INVALID;
    const float inv = 1.f / ((float)C * (float)H * (float)W);   // both losses are means over all C H W elements
    const dim3 grid((W + svgir::LT - 1) / svgir::LT, (H + svgir::LT - 1) / svgir::LT, C);
    hipLaunchKernelGGL(svgir::ssim_bwd_kernel, grid, dim3(svgir::LT * svgir::LT), 0, (hipStream_t)stream, img1, img2, dmaps, H, W,
                       svgir::ssim_window(), g_ssim_mean * inv, g_l1_mean * inv, g_dev, dL_dimg1);
    return hipGetLastError() == hipSuccess ? SVGIR_OK : SVGIR_ERR_HIP;
}

}  // extern "C"
