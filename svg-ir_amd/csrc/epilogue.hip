// svg-ir_amd/csrc/epilogue.hip -- image-space epilogue right after the svgss rasterizer.
//
// Replaces the PyTorch tail of `render_view` (gaussian_renderer/svgss.py:187-246): division of the feature / vfeature
// planes by the rendered opacity (clamp_min 1e-5), channel split, `rgb_to_srgb` (utils/graphics_utils.py:198-215) and
// compositing over the background, and `depth2normal` (utils/image_utils.py:61-125).  The reference runs ~40 elementwise
// torch kernels over [C,H,W] planes (and autograd keeps their inputs); here one pass reads the rasterizer's planes once
// and writes every result plane, and one pass maps the upstream gradients of those planes back to
// dL/d(opacity, feature, vfeature) -- both pure HBM streams: 4 (1 + S + VS/4) bytes in, 4 NOUT bytes out per pixel.
//
// Result planes [NOUT,H,W] in this order (each group 3 planes; one-channel quantities are broadcast over the three
// background channels exactly like `r * opacity + (1 - opacity) * bg[:, None, None]` does in the reference):
//   training (S=4: visibility, local light; VS/4=13: pbr, base colour, normal, roughness, diffuse)     NOUT = 21
//     pbr | normal | base_color | roughness | diffuse | local_lights | visibility
//   evaluation (S=7: light, local light, visibility; VS/4=16: pbr, base, normal, roughness, direct, indirect)  NOUT = 27
//     pbr | normal | base_color | roughness | direct | indirect | lights | local_lights | visibility
// with x = plane / max(opacity, 1e-5):
//   pbr = srgb(x o + (1 - o) bg)      normal = x      direct, indirect = srgb(x)
//   base_color, diffuse, lights, local_lights = srgb(x) o + (1 - o) bg      roughness, visibility = x o + (1 - o) bg
#include "common.hpp"

namespace svgir {

namespace {

enum PlaneKind { K_SRGB_OF_OVER = 0, K_PLAIN = 1, K_OVER_SRGB = 2, K_OVER_LIN = 3, K_SRGB = 4 };
struct PlaneGroup { int kind, src, vf, n; };   // src: first source channel; vf: source is vfeature; n: source channels (1 or 3)

__device__ __forceinline__ float srgb(float x) {
    const float y = x > 0.0031308f ? powf(fmaxf(x, 0.0031308f), 1.0f / 2.4f) * 1.055f - 0.055f : 12.92f * x;
    return fminf(1.f, fmaxf(0.f, y));
}
// d srgb / dx (0 where the final clip to [0,1] is active)
__device__ __forceinline__ float dsrgb(float x) {
    const float y = x > 0.0031308f ? powf(fmaxf(x, 0.0031308f), 1.0f / 2.4f) * 1.055f - 0.055f : 12.92f * x;
    if (y < 0.f || y > 1.f) return 0.f;
    return x > 0.0031308f ? (1.055f / 2.4f) * powf(fmaxf(x, 0.0031308f), 1.0f / 2.4f - 1.0f) : 12.92f;
}

template <bool TRAINING> struct Groups;
template <> struct Groups<true> {
    static constexpr int N = 7, S = 4, VC = 13;
    static constexpr PlaneGroup g[7] = {{K_SRGB_OF_OVER, 0, 1, 3}, {K_PLAIN, 6, 1, 3}, {K_OVER_SRGB, 3, 1, 3}, {K_OVER_LIN, 9, 1, 1},
                                        {K_OVER_SRGB, 10, 1, 3}, {K_OVER_SRGB, 1, 0, 3}, {K_OVER_LIN, 0, 0, 1}};
};
template <> struct Groups<false> {
    static constexpr int N = 9, S = 7, VC = 16;
    static constexpr PlaneGroup g[9] = {{K_SRGB_OF_OVER, 0, 1, 3}, {K_PLAIN, 6, 1, 3}, {K_OVER_SRGB, 3, 1, 3}, {K_OVER_LIN, 9, 1, 1},
                                        {K_SRGB, 10, 1, 3}, {K_SRGB, 13, 1, 3}, {K_OVER_SRGB, 0, 0, 3}, {K_OVER_SRGB, 3, 0, 3},
                                        {K_OVER_LIN, 6, 0, 1}};
};

struct UnpackArgs {
    int W, H, training;
    const float *bg, *opacity, *feature, *vfeature;
    float* out;                                        // forward
    const float* g_out; float *d_opacity, *d_feature, *d_vfeature;   // backward
};

template <bool BWD, bool TRAINING>
__global__ void __launch_bounds__(BLOCK) unpack_kernel(const UnpackArgs a) {
    using G = Groups<TRAINING>;
    const size_t N = (size_t)a.W * a.H;
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= N) return;
    constexpr int S = G::S, VC = G::VC;
    const float o = a.opacity[i];
    const bool clamped = !(o > 1e-5f);
    const float inv = 1.f / fmaxf(o, 1e-5f);
    const float dinv = clamped ? 0.f : -inv * inv;     // d inv / d o
    const float bg[3] = {a.bg[0], a.bg[1], a.bg[2]};
    float d_o = 0.f;
    float dS[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, dV[16];
#pragma unroll
    for (int c = 0; c < 16; c++) dV[c] = 0.f;
#pragma unroll
    for (int k = 0; k < G::N; k++) {
        constexpr const PlaneGroup* g = G::g;
        const float* src = g[k].vf ? a.vfeature : a.feature;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int sc = g[k].src + (g[k].n == 3 ? c : 0);
            const float raw = src[(size_t)sc * N + i];
            const float x = raw * inv;
            const size_t oi = (size_t)(3 * k + c) * N + i;
            if (!BWD) {
                float y;
                switch (g[k].kind) {
                    case K_SRGB_OF_OVER: y = srgb(x * o + (1.f - o) * bg[c]); break;
                    case K_PLAIN: y = x; break;
                    case K_OVER_SRGB: y = srgb(x) * o + (1.f - o) * bg[c]; break;
                    case K_OVER_LIN: y = x * o + (1.f - o) * bg[c]; break;
                    default: y = srgb(x); break;
                }
                a.out[oi] = y;
            } else {
                const float gy = a.g_out[oi];
                float dx = 0.f, dopac = 0.f;    // dy/dx, explicit dy/do
                switch (g[k].kind) {
                    case K_SRGB_OF_OVER: { const float ds = dsrgb(x * o + (1.f - o) * bg[c]); dx = ds * o; dopac = ds * (x - bg[c]); break; }
                    case K_PLAIN: dx = 1.f; break;
                    case K_OVER_SRGB: dx = dsrgb(x) * o; dopac = srgb(x) - bg[c]; break;
                    case K_OVER_LIN: dx = o; dopac = x - bg[c]; break;
                    default: dx = dsrgb(x); break;
                }
                const float draw = gy * dx * inv;
                d_o += gy * (dopac + dx * raw * dinv);
                if (g[k].vf) dV[sc] += draw; else dS[sc] += draw;
            }
        }
    }
    if (BWD) {
        a.d_opacity[i] = d_o;
#pragma unroll
        for (int c = 0; c < 7; c++) if (c < S) a.d_feature[(size_t)c * N + i] = dS[c];
#pragma unroll
        for (int c = 0; c < 16; c++) if (c < VC) a.d_vfeature[(size_t)c * N + i] = dV[c];
    }
}

// depth2normal (utils/image_utils.py:61-125): back-project the pixel and its four neighbours (replicate padding) with
// the reference's intrinsics -- K = diag(focal(FoVy, H), focal(FoVx, W)), i.e. x is divided by the y focal length and
// vice versa, as in the reference --, mask, sum of the four cross products of neighbouring differences, normalise, mask.
__global__ void __launch_bounds__(BLOCK) depth2normal_kernel(const float* __restrict__ depth, const float* __restrict__ mask,
                                                             int W, int H, float k00, float k11, float ppx, float ppy,
                                                             float* __restrict__ normal) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= W * H) return;
    const int x = i % W, y = i / W;
    auto cam = [&](int xx, int yy, float* p, float& m) {
        xx = min(max(xx, 0), W - 1); yy = min(max(yy, 0), H - 1);
        const float d = depth[yy * W + xx];
        m = mask[yy * W + xx] != 0.f ? 1.f : 0.f;
        p[0] = ((float)xx - ppx) * d / k00; p[1] = ((float)yy - ppy) * d / k11; p[2] = d;
    };
    float pc[3], pu[3], pl[3], pb[3], pr[3], mc, mu, ml, mb, mr;
    cam(x, y, pc, mc); cam(x, y - 1, pu, mu); cam(x - 1, y, pl, ml); cam(x, y + 1, pb, mb); cam(x + 1, y, pr, mr);
    float c[3], u[3], l[3], b[3], r[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        c[j] = pc[j] * mc;
        u[j] = (pu[j] - c[j]) * mu; l[j] = (pl[j] - c[j]) * ml; b[j] = (pb[j] - c[j]) * mb; r[j] = (pr[j] - c[j]) * mr;
    }
    auto cross = [](const float* a, const float* b_, float* o) {
        o[0] = a[1] * b_[2] - a[2] * b_[1]; o[1] = a[2] * b_[0] - a[0] * b_[2]; o[2] = a[0] * b_[1] - a[1] * b_[0];
    };
    float n1[3], n2[3], n3[3], n4[3];
    cross(u, l, n1); cross(r, u, n2); cross(b, r, n3); cross(l, b, n4);
    float n[3] = {n1[0] + n2[0] + n3[0] + n4[0], n1[1] + n2[1] + n3[1] + n4[1], n1[2] + n2[2] + n3[2] + n4[2]};
    const float len = fmaxf(sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]), 1e-12f);
    const size_t N = (size_t)W * H;
#pragma unroll
    for (int j = 0; j < 3; j++) normal[(size_t)j * N + i] = n[j] / len * mc;
}

// adjoint of depth2normal: every pixel recomputes its five camera points and scatters d(loss)/d(depth) of its own normal to
// the (clamped) depths it read -- the depth-normal consistency loss differentiates through the pseudo normal
// (gaussian_renderer/render.py:158-160).  dL_ddepth must be zero on entry.
__global__ void __launch_bounds__(BLOCK) depth2normal_bwd_kernel(const float* __restrict__ depth, const float* __restrict__ mask,
                                                                 const float* __restrict__ g_normal, int W, int H, float k00, float k11,
                                                                 float ppx, float ppy, float* __restrict__ dL_ddepth) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= W * H) return;
    const int x = i % W, y = i / W;
    float ray[5][3], m[5];
    int at[5];
    float p[5][3];
    const int ox[5] = {0, 0, -1, 0, 1}, oy[5] = {0, -1, 0, 1, 0};   // centre, up, left, bottom, right
#pragma unroll
    for (int q = 0; q < 5; q++) {
        const int xx = min(max(x + ox[q], 0), W - 1), yy = min(max(y + oy[q], 0), H - 1);
        at[q] = yy * W + xx;
        const float d = depth[at[q]];
        m[q] = mask[at[q]] != 0.f ? 1.f : 0.f;
        ray[q][0] = ((float)xx - ppx) / k00; ray[q][1] = ((float)yy - ppy) / k11; ray[q][2] = 1.f;
        p[q][0] = ((float)xx - ppx) * d / k00; p[q][1] = ((float)yy - ppy) * d / k11; p[q][2] = d;
    }
    float c[3], e[5][3];   // e[1..4] = u, l, b, r
#pragma unroll
    for (int j = 0; j < 3; j++) {
        c[j] = p[0][j] * m[0];
#pragma unroll
        for (int q = 1; q < 5; q++) e[q][j] = (p[q][j] - c[j]) * m[q];
    }
    auto cross = [](const float* a, const float* b_, float* o) {
        o[0] = a[1] * b_[2] - a[2] * b_[1]; o[1] = a[2] * b_[0] - a[0] * b_[2]; o[2] = a[0] * b_[1] - a[1] * b_[0];
    };
    const float *u = e[1], *l = e[2], *b = e[3], *r = e[4];
    float n1[3], n2[3], n3[3], n4[3], n[3];
    cross(u, l, n1); cross(r, u, n2); cross(b, r, n3); cross(l, b, n4);
#pragma unroll
    for (int j = 0; j < 3; j++) n[j] = n1[j] + n2[j] + n3[j] + n4[j];
    const size_t N = (size_t)W * H;
    // normal = n / max(|n|, 1e-12) * mask_c
    float g[3] = {g_normal[i] * m[0], g_normal[N + i] * m[0], g_normal[2 * N + i] * m[0]};
    const float len = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    float gn[3];
    if (len > 1e-12f) {
        const float dot = (n[0] * g[0] + n[1] * g[1] + n[2] * g[2]) / (len * len);
#pragma unroll
        for (int j = 0; j < 3; j++) gn[j] = (g[j] - n[j] * dot) / len;
    } else {
#pragma unroll
        for (int j = 0; j < 3; j++) gn[j] = g[j] / 1e-12f;
    }
    // n = u x l + r x u + b x r + l x b;  for a x b: d/da = b x g, d/db = g x a
    float t1[3], t2[3], ge[5][3];
    cross(l, gn, t1); cross(gn, r, t2);
#pragma unroll
    for (int j = 0; j < 3; j++) ge[1][j] = t1[j] + t2[j];   // u
    cross(gn, u, t1); cross(b, gn, t2);
#pragma unroll
    for (int j = 0; j < 3; j++) ge[2][j] = t1[j] + t2[j];   // l
    cross(r, gn, t1); cross(gn, l, t2);
#pragma unroll
    for (int j = 0; j < 3; j++) ge[3][j] = t1[j] + t2[j];   // b
    cross(u, gn, t1); cross(gn, b, t2);
#pragma unroll
    for (int j = 0; j < 3; j++) ge[4][j] = t1[j] + t2[j];   // r
    float gc[3] = {0.f, 0.f, 0.f};
    float gd[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 1; q < 5; q++) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const float gp = ge[q][j] * m[q];       // e = (p_q - c) m_q
            gd[q] += gp * ray[q][j];
            gc[j] -= gp;
        }
    }
#pragma unroll
    for (int j = 0; j < 3; j++) gd[0] += gc[j] * m[0] * ray[0][j];   // c = p_c m_c
#pragma unroll
    for (int q = 0; q < 5; q++)
        if (gd[q] != 0.f) atomic_add_f32(&dL_ddepth[at[q]], gd[q]);
}

// ---- stage 1 (rgss) feature packing and image-space tail (gaussian_renderer/render.py:83-91, 107-114) ----------
// pack: features[P,5] = [geometric normal (world), view depth d, d^2], d = (xyz1 @ viewmatrix).z
__global__ void __launch_bounds__(BLOCK) pack_rgss_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ normals,
                                                          const float* __restrict__ V, float* __restrict__ features,
                                                          const float* __restrict__ g_features, float* __restrict__ d_means3D,
                                                          float* __restrict__ d_normals) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= P) return;
    const float x = means3D[3 * i], y = means3D[3 * i + 1], z = means3D[3 * i + 2];
    const float d = x * V[2] + y * V[6] + z * V[10] + V[14];
    if (features) {
        float* f = features + (size_t)i * 5;
        f[0] = normals[3 * i]; f[1] = normals[3 * i + 1]; f[2] = normals[3 * i + 2]; f[3] = d; f[4] = d * d;
    }
    if (g_features) {
        const float* g = g_features + (size_t)i * 5;
        const float gd = g[3] + 2.f * d * g[4];
        d_normals[3 * i] = g[0]; d_normals[3 * i + 1] = g[1]; d_normals[3 * i + 2] = g[2];
        d_means3D[3 * i] = gd * V[2]; d_means3D[3 * i + 1] = gd * V[6]; d_means3D[3 * i + 2] = gd * V[10];
    }
}

// unpack: x_c = feature_c / max(opacity, 1e-5) * (num_contrib > 0); out = [x_0..x_4 | depth_var = x_4 - depth^2]
template <bool BWD>
__global__ void __launch_bounds__(BLOCK) unpack_rgss_kernel(int N, const int32_t* __restrict__ num_contrib,
                                                            const float* __restrict__ opacity, const float* __restrict__ depth,
                                                            const float* __restrict__ feature, float* __restrict__ out,
                                                            const float* __restrict__ g_out, float* __restrict__ d_opacity,
                                                            float* __restrict__ d_depth, float* __restrict__ d_feature) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= N) return;
    const float o = opacity[i];
    const float m = num_contrib[i] > 0 ? 1.f : 0.f;
    const float inv = m / fmaxf(o, 1e-5f);
    const float dinv = o > 1e-5f ? -inv / o : 0.f;
    const float dep = depth[i];
    if (!BWD) {
#pragma unroll
        for (int c = 0; c < 5; c++) out[(size_t)c * N + i] = feature[(size_t)c * N + i] * inv;
        out[(size_t)5 * N + i] = feature[(size_t)4 * N + i] * inv - dep * dep;
    } else {
        float d_o = 0.f;
        const float gv = g_out[(size_t)5 * N + i];
#pragma unroll
        for (int c = 0; c < 5; c++) {
            const float g = g_out[(size_t)c * N + i] + (c == 4 ? gv : 0.f);
            d_feature[(size_t)c * N + i] = g * inv;
            d_o += g * feature[(size_t)c * N + i] * dinv;
        }
        d_opacity[i] = d_o;
        d_depth[i] = -2.f * dep * gv;
    }
}

}  // namespace

}  // namespace svgir

using namespace svgir;

extern "C" {

int svgir_unpack_planes(int32_t training) { return training ? 21 : 27; }

int svgir_unpack_forward(int32_t W, int32_t H, int32_t training, const float* bg, const float* opacity, const float* feature,
                         const float* vfeature, float* out, void* stream) {
    if (W <= 0 || H <= 0 || !bg || !opacity || !feature || !vfeature || !out) return SVGIR_ERR_INVALID;
    UnpackArgs a{};
    a.W = W; a.H = H; a.training = training; a.bg = bg; a.opacity = opacity; a.feature = feature; a.vfeature = vfeature; a.out = out;
    const size_t N = (size_t)W * H;
    hipStream_t s = (hipStream_t)stream;
    StageMarks tm = stage_begin(s);
    if (training) hipLaunchKernelGGL((unpack_kernel<false, true>), dim3((unsigned)((N + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, a);
    else hipLaunchKernelGGL((unpack_kernel<false, false>), dim3((unsigned)((N + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, a);
    stage_mark(tm, "unpack_fwd");
    return hipGetLastError() == hipSuccess ? 0 : SVGIR_ERR_HIP;
}

int svgir_unpack_backward(int32_t W, int32_t H, int32_t training, const float* bg, const float* opacity, const float* feature,
                          const float* vfeature, const float* dL_dout, float* dL_dopacity, float* dL_dfeature,
                          float* dL_dvfeature, void* stream) {
    if (W <= 0 || H <= 0 || !bg || !opacity || !feature || !vfeature || !dL_dout || !dL_dopacity || !dL_dfeature || !dL_dvfeature)
        return SVGIR_ERR_INVALID;
    UnpackArgs a{};
    a.W = W; a.H = H; a.training = training; a.bg = bg; a.opacity = opacity; a.feature = feature; a.vfeature = vfeature;
    a.g_out = dL_dout; a.d_opacity = dL_dopacity; a.d_feature = dL_dfeature; a.d_vfeature = dL_dvfeature;
    const size_t N = (size_t)W * H;
    hipStream_t s = (hipStream_t)stream;
    StageMarks tm = stage_begin(s);
    if (training) hipLaunchKernelGGL((unpack_kernel<true, true>), dim3((unsigned)((N + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, a);
    else hipLaunchKernelGGL((unpack_kernel<true, false>), dim3((unsigned)((N + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, a);
    stage_mark(tm, "unpack_bwd");
    return hipGetLastError() == hipSuccess ? 0 : SVGIR_ERR_HIP;
}

int svgir_pack_rgss_forward(int32_t P, const float* means3D, const float* normals, const float* viewmatrix, float* features,
                            void* stream) {
    if (P < 0 || (P > 0 && (!means3D || !normals || !viewmatrix || !features))) return SVGIR_ERR_INVALID;
    if (P == 0) return 0;
    hipLaunchKernelGGL(pack_rgss_kernel, dim3((P + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, (hipStream_t)stream, P, means3D, normals,
                       viewmatrix, features, (const float*)nullptr, (float*)nullptr, (float*)nullptr);
    return hipGetLastError() == hipSuccess ? 0 : SVGIR_ERR_HIP;
}

int svgir_pack_rgss_backward(int32_t P, const float* means3D, const float* viewmatrix, const float* dL_dfeatures,
                             float* dL_dmeans3D, float* dL_dnormals, void* stream) {
    if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !dL_dfeatures || !dL_dmeans3D || !dL_dnormals))) return SVGIR_ERR_INVALID;
    if (P == 0) return 0;
    hipLaunchKernelGGL(pack_rgss_kernel, dim3((P + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, (hipStream_t)stream, P, means3D,
                       (const float*)nullptr, viewmatrix, (float*)nullptr, dL_dfeatures, dL_dmeans3D, dL_dnormals);
    return hipGetLastError() == hipSuccess ? 0 : SVGIR_ERR_HIP;
}

int svgir_unpack_rgss_forward(int32_t W, int32_t H, const int32_t* num_contrib, const float* opacity, const float* depth,
                              const float* feature, float* out, void* stream) {
    if (W <= 0 || H <= 0 || !num_contrib || !opacity || !depth || !feature || !out) return SVGIR_ERR_INVALID;
    const int N = W * H;
    hipLaunchKernelGGL(unpack_rgss_kernel<false>, dim3((N + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, (hipStream_t)stream, N, num_contrib,
                       opacity, depth, feature, out, (const float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr);
    return hipGetLastError() == hipSuccess ? 0 : SVGIR_ERR_HIP;
}

int svgir_unpack_rgss_backward(int32_t W, int32_t H, const int32_t* num_contrib, const float* opacity, const float* depth,
                               const float* feature, const float* dL_dout, float* dL_dopacity, float* dL_ddepth,
                               float* dL_dfeature, void* stream) {
    if (W <= 0 || H <= 0 || !num_contrib || !opacity || !depth || !feature || !dL_dout || !dL_dopacity || !dL_ddepth || !dL_dfeature)
        return SVGIR_ERR_INVALID;
    const int N = W * H;
    hipLaunchKernelGGL(unpack_rgss_kernel<true>, dim3((N + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, (hipStream_t)stream, N, num_contrib,
                       opacity, depth, feature, (float*)nullptr, dL_dout, dL_dopacity, dL_ddepth, dL_dfeature);
    return hipGetLastError() == hipSuccess ? 0 : SVGIR_ERR_HIP;
}

int svgir_depth2normal(int32_t W, int32_t H, const float* depth, const float* mask, float fovx, float fovy, float prcp_x,
                       float prcp_y, float* normal, void* stream) {
    if (W <= 0 || H <= 0 || !depth || !mask || !normal) return SVGIR_ERR_INVALID;
    const float k00 = (float)H / (2.f * tanf(fovy * 0.5f)), k11 = (float)W / (2.f * tanf(fovx * 0.5f));
    hipStream_t s = (hipStream_t)stream;
    StageMarks tm = stage_begin(s);
    hipLaunchKernelGGL(depth2normal_kernel, dim3((W * H + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, depth, mask, W, H, k00, k11,
                       prcp_x * (float)W, prcp_y * (float)H, normal);
    stage_mark(tm, "depth2normal");
    return hipGetLastError() == hipSuccess ? 0 : SVGIR_ERR_HIP;
}

int svgir_depth2normal_backward(int32_t W, int32_t H, const float* depth, const float* mask, const float* dL_dnormal, float fovx,
                                float fovy, float prcp_x, float prcp_y, float* dL_ddepth, void* stream) {
    if (W <= 0 || H <= 0 || !depth || !mask || !dL_dnormal || !dL_ddepth) return SVGIR_ERR_INVALID;
    const float k00 = (float)H / (2.f * tanf(fovy * 0.5f)), k11 = (float)W / (2.f * tanf(fovx * 0.5f));
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(dL_ddepth, 0, (size_t)W * H * 4, s) != hipSuccess) return SVGIR_ERR_HIP;
    hipLaunchKernelGGL(depth2normal_bwd_kernel, dim3((W * H + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, depth, mask, dL_dnormal, W, H, k00,
                       k11, prcp_x * (float)W, prcp_y * (float)H, dL_ddepth);
    return hipGetLastError() == hipSuccess ? 0 : SVGIR_ERR_HIP;
}

}  // extern "C"
