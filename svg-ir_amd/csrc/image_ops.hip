// svg-ir_amd/csrc/image_ops.hip -- rgss image-space epilogue kernels.
//
// Replaces renderSurfaceXYZCUDA (rgss forward.cu:538-565) and renderPseudoNormalCUDA (rgss forward.cu:567-631):
// view-space xyz from depth / max(opacity, 1e-7) and the principal point, then a 3x3 Sobel-style cross-product
// normal rotated to world space.  Only run when `computer_pseudo_normal` is set; not differentiated.
// Pure streaming stencils: one lane per pixel, rows of 64 consecutive pixels per wave (coalesced 256-B rows).
#include "common.hpp"

#pragma clang fp contract(off)

namespace svgir {

namespace {

__global__ void __launch_bounds__(BLOCK) surface_xyz_kernel(int W, int H, float fx, float fy, float cx, float cy,
                                                            const float* __restrict__ opac,
                                                            const float* __restrict__ depth, float* __restrict__ xyz) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t N = (size_t)W * H, id = (size_t)W * y + x;
    const float d = depth[id] / fmaxf(opac[id], 0.0000001f);
    xyz[id] = ((float)x - cx) / fx * d;
    xyz[N + id] = ((float)y - cy) / fy * d;
    xyz[2 * N + id] = d;
}

__global__ void __launch_bounds__(BLOCK) pseudo_normal_kernel(int W, int H, const float* __restrict__ V,
                                                              const float* __restrict__ xyz, float* __restrict__ nrm) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t N = (size_t)W * H;
    const int ym = y == 0 ? 0 : y - 1, yp = y == H - 1 ? H - 1 : y + 1;
    const int xm = x == 0 ? 0 : x - 1, xp = x == W - 1 ? W - 1 : x + 1;
    float ga[3], gb[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float* p = xyz + i * N;
        const float v00 = p[(size_t)W * ym + xm], v01 = p[(size_t)W * ym + x], v02 = p[(size_t)W * ym + xp];
        const float v10 = p[(size_t)W * y + xm], v12 = p[(size_t)W * y + xp];
        const float v20 = p[(size_t)W * yp + xm], v21 = p[(size_t)W * yp + x], v22 = p[(size_t)W * yp + xp];
        ga[i] = -0.125f * v00 + 0.125f * v02 - 0.25f * v10 + 0.25f * v12 - 0.125f * v20 + 0.125f * v22;
        gb[i] = -0.125f * v00 - 0.25f * v01 - 0.125f * v02 + 0.125f * v20 + 0.25f * v21 + 0.125f * v22;
    }
    float n[3] = {ga[1] * gb[2] - ga[2] * gb[1], -ga[0] * gb[2] + ga[2] * gb[0], ga[0] * gb[1] - ga[1] * gb[0]};
    const float nn = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    const size_t id = (size_t)W * y + x;
    if (nn <= 0.00000f) {   // degenerate stencil: the reference leaves its zero-initialised output (forward.cu:620-622)
        nrm[id] = 0.f; nrm[N + id] = 0.f; nrm[2 * N + id] = 0.f;
        return;
    }
    n[0] = -n[0] / nn; n[1] = -n[1] / nn; n[2] = -n[2] / nn;
    nrm[id] = V[0] * n[0] + V[1] * n[1] + V[2] * n[2];
    nrm[N + id] = V[4] * n[0] + V[5] * n[1] + V[6] * n[2];
    nrm[2 * N + id] = V[8] * n[0] + V[9] * n[1] + V[10] * n[2];
}

}  // namespace

void launch_image_ops(int W, int H, const float* view, float focal_x, float focal_y, float cx, float cy,
                      const float* opacity, const float* depth, float* pseudo_normal, float* surface_xyz,
                      hipStream_t s) {
    const dim3 grid((W + 63) / 64, (H + 3) / 4);
    hipLaunchKernelGGL(surface_xyz_kernel, grid, dim3(BLOCK), 0, s, W, H, focal_x, focal_y, cx, cy, opacity, depth,
                       surface_xyz);
    hipLaunchKernelGGL(pseudo_normal_kernel, grid, dim3(BLOCK), 0, s, W, H, view, surface_xyz, pseudo_normal);
}

}  // namespace svgir
