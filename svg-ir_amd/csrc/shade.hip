// svg-ir_amd/csrc/shade.hip -- fused per-splat spatially-varying BRDF shading (forward + backward).
//
// Replaces the PyTorch rendering_equation4 + GGX_specular4 (gaussian_renderer/svgss.py:537-631), the env lookup of
// DirectLightMap.direct_light / EnvLight.direct_light (scene/direct_light_map.py:70-83, scene/envmap.py:53-72) and the
// feature packing of svgss.py:143-166.  The reference materialises dozens of [P,Ns,4,3] temporaries (and their
// autograd copies); here every per-sample quantity lives in registers / LDS and HBM sees the inputs once:
// 32*P*Ns bytes of incident data (+ P*~130 B) in, P*(70+S+VS)*4 bytes out => an HBM-streaming kernel.
//
// CDNA4 mapping: one wave64 per Gaussian (4 per workgroup).
//   phase 1, lane = incident sample: coalesced loads of dirs / radiance / visibility / area (64 consecutive samples),
//            per-sample terms that do not depend on the corner (L, H, Schlick term, env lookup -> global light) are
//            written to a 17-float LDS record per sample; light means are accumulated.
//   phase 2, lane = (corner k = lane/16, sample group = lane%16): every lane walks Ns/16 samples of ONE corner
//            (LDS reads: conflict-free stride-17 rows, broadcast across the four corner rows), evaluates the GGX
//            term and accumulates its 15 outputs; a 4-step DPP butterfly over the 16-lane row finishes the sums.
//   epilogue: results go through a 70-float LDS strip so that `reduced`, `features` and `vfeatures` rows are
//            written with consecutive lanes.
// The backward re-runs phase 1/2 with the adjoint arithmetic; per-sample light gradients are exchanged between the
// four corner rows through LDS, env-texel gradients are accumulated in a workgroup-private LDS image by persistent
// workgroups and flushed with one global atomic per texel per workgroup.
#include <algorithm>

#include "common.hpp"
#include "shade_tables.hpp"

#include "dev_trace.hpp"

namespace svgir {

namespace {

constexpr int SREC = 17;   // floats per staged sample: d(3) L(3) H(3) frac0 Lg(3) Ll(3) area
constexpr int NRED = SVGIR_SHADE_REDUCED;
constexpr float kPi = 3.14159265358979323846f;
constexpr float kInvPi = 0.31830988618379067154f;

struct ShadeArgs {
    svgir_shade_params p;
    float* reduced; float* features; float* vfeatures;
};

__device__ __forceinline__ float row16_sum(float v) {  // sum over the 16 lanes of a DPP row, result in all lanes
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));  // row_mirror
    return v;
}

__device__ __forceinline__ float quad_sum(float v) {  // sum over the 4 lanes of a quad, result in all 4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    return v;
}

#ifndef SHADE_PRECISE
#define SHADE_PRECISE 2
#endif
// 1 / max(sqrt(x2), 1e-12) for the normalisations.  At the glossy end of the reference's roughness range (0.09: alpha^2 = 6.6e-5) the GGX
// denominator N.H^2 (alpha^2 - 1) + 1 amplifies an error of N.H 15 000 times, so the unit vectors must be as good as the reference's
// (torch: correctly rounded sqrt and division): the hardware's 1-ulp rsq gets one Newton step.
__device__ __forceinline__ float inv_norm(float x2) {
    const float y = fminf(__builtin_amdgcn_rsqf(x2), 1e12f);
#if SHADE_PRECISE >= 1
    return x2 > 1e-24f ? y * fmaf(-0.5f * x2 * y, y, 1.5f) : y;
#else
    return y;
#endif
}


// (the f(env) table -- one float4 per texel {f(r), f(g), f(b), 0}: a bilinear tap is ONE 16-byte gather instead of three dwords -- is
// built by shade_prologue_kernel below)

// Lat-long bilinear lookup (grid_sample, align_corners=True, zero padding) of direction d.
struct EnvTap { int idx[4]; float w[4]; };
__device__ __forceinline__ void env_taps(const float* d, int He, int We, EnvTap& t) {
    const float phi = acosf(d[2]) - 1e-6f;
    const float theta = atan2f(d[1], d[0]);
    const float gy = phi * (2.f * kInvPi) - 1.f;
    const float gx = -theta * kInvPi;
    const float x = (gx + 1.f) * 0.5f * (float)(We - 1);
    const float y = (gy + 1.f) * 0.5f * (float)(He - 1);
    const float x0f = floorf(x), y0f = floorf(y);
    const float fx = x - x0f, fy = y - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int xi = x0 + (j & 1), yi = y0 + (j >> 1);
        const bool ok = xi >= 0 && xi < We && yi >= 0 && yi < He;
        t.idx[j] = ok ? yi * We + xi : -1;   // texel index (the table holds one float4 per texel)
        t.w[j] = ((j & 1) ? fx : 1.f - fx) * ((j >> 1) ? fy : 1.f - fy);
    }
}

// ---- incident-direction lattice -----------------------------------------------------------
// `fibonacci_sphere_sampling` (utils/graphics_utils.py:9-37) + `rotation_between_z` (utils/sh_utils.py:36-68), as
// called by `sample_incident_rays` (scene/gaussian_model.py:23-31): sample i of a surfel = R(n) * (sin t_i rad_i,
// cos t_i rad_i, z_i), re-normalised, with z_i = max(1 - 2 i / (2 Ns - 1), sin 10 deg), rad_i = sqrt(1 - z_i^2),
// t_i = delta * i (+ the surfel's random azimuth offset in training), areas = 2 pi.  The reference materialises
// [P,Ns,3] + [P,Ns,1] tensors (16 of the 32 bytes per sample the shading kernels stream); here the per-index part
// {sin t_i, cos t_i, z_i, rad_i} is a tiny table (one launch per call) and the kernels build the directions in
// registers from 3 (+1) floats per SURFEL.
// The reference evaluates t_i in fp32 (one ulp is 6e-5 rad at t = 900); the table is built from exactly that rounded
// product, and the offset is added with the rounding error of the fp32 sum carried to first order.
constexpr float kTwoPi = 6.28318530717958647692f;

__global__ void __launch_bounds__(BLOCK) lattice_table_kernel(int Ns, float4* __restrict__ tab) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= Ns) return;
    const float fi = (float)i;
    const float z = fmaxf(1.f - 2.f * fi / (float)(2 * Ns - 1), 0.17364817766693033f);   // sin(10 deg)
    const float rad = sqrtf(1.f - z * z);
    const float t = kLatticeDelta * fi;
    float sn, cs;
    sincosf(t, &sn, &cs);
    tab[i] = make_float4(sn, cs, z, rad);
}

// One launch in front of a shading kernel: the f(env) table, the lattice table (when the directions are generated in-kernel) and, for
// the backward, the clear of the env-gradient accumulator -- three tiny jobs that used to be three launches (shade_tables.hpp).
__global__ void __launch_bounds__(BLOCK) shade_prologue_kernel(const ShadeTables t) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    shade_table_entry(t, i);
    for (int j = i; j < t.nzero; j += gridDim.x * BLOCK) t.zero[j] = 0.f;
}
static ShadeTables shade_tables_of(const svgir_shade_params* p, float* zero, int nzero) {
    ShadeTables t;
    t.env = p->env; t.env_tab = (float4*)p->env_work; t.ntexel = p->env_h * p->env_w; t.softplus = p->env_softplus;
    t.Ns = p->Ns; t.lat_tab = p->incident_dirs ? (float4*)nullptr : (float4*)p->lattice_work;
    t.zero = zero; t.nzero = nzero;
    return t;
}
static void launch_shade_prologue(const svgir_shade_params* p, float* zero, int nzero, hipStream_t s) {
    const ShadeTables t = shade_tables_of(p, zero, nzero);
    hipLaunchKernelGGL(shade_prologue_kernel, dim3((t.entries() + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, t);
}

struct LatticeFrame { float R[9]; float so, co, off; };   // rotation taking +z to the surfel's normal; offset angle
__device__ __forceinline__ LatticeFrame lattice_frame(const float* __restrict__ normals, const float* __restrict__ offsets,
                                                      size_t g) {
    LatticeFrame f;
    const float nx = normals[g * 3], ny = normals[g * 3 + 1], nz = normals[g * 3 + 2];
    const float v1 = -ny, v2 = nx;
    const float ic = 1.f / fmaxf(nz + 1.f, 1e-7f);
    const bool ok = nz + 1.f > 0.f;
    f.R[0] = ok ? 1.f - v2 * v2 * ic : -1.f; f.R[1] = ok ? v1 * v2 * ic : 0.f; f.R[2] = ok ? v2 : 0.f;
    f.R[3] = ok ? v1 * v2 * ic : 0.f; f.R[4] = ok ? 1.f - v1 * v1 * ic : -1.f; f.R[5] = ok ? -v1 : 0.f;
    f.R[6] = ok ? -v2 : 0.f; f.R[7] = ok ? v1 : 0.f; f.R[8] = ok ? 1.f - (v2 * v2 + v1 * v1) * ic : -1.f;
    f.off = offsets ? offsets[g] : 0.f;
    f.so = 0.f; f.co = 1.f;
    if (offsets) sincosf(f.off, &f.so, &f.co);
    return f;
}
__device__ __forceinline__ void lattice_dir(const LatticeFrame& f, const float4 t, int i, bool has_offset, float* d) {
    float sn = t.x, cs = t.y;
    if (has_offset) {
        // the reference takes sin / cos of th = fl(offset + t_i).  With e = (offset + t_i) - th exactly (two-sum):
        // sin th = sin(offset + t_i) - e cos(offset + t_i) + O(e^2), and the sine / cosine of the exact sum come from the
        // table and the surfel's (sin, cos)(offset) by the addition theorems.
        float ti, th, e;
        {
#pragma clang fp contract(off)
            ti = kLatticeDelta * (float)i;
            th = f.off + ti;
            const float bb = th - f.off;
            e = (f.off - (th - bb)) + (ti - bb);
        }
        const float se = f.so * t.y + f.co * t.x, ce = f.co * t.y - f.so * t.x;
        sn = se - e * ce; cs = ce + e * se;
    }
    const float x = sn * t.w, y = cs * t.w, z = t.z;
    float v[3] = {f.R[0] * x + f.R[1] * y + f.R[2] * z, f.R[3] * x + f.R[4] * y + f.R[5] * z, f.R[6] * x + f.R[7] * y + f.R[8] * z};
    const float il = inv_norm(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);   // = 1 / max(|.|, 1e-12)
    d[0] = v[0] * il; d[1] = v[1] * il; d[2] = v[2] * il;
}

// materialises the lattice: the drop-in for fibonacci_sphere_sampling's return values
__global__ void __launch_bounds__(BLOCK) incident_dirs_kernel(int P, int Ns, const float* __restrict__ normals,
                                                              const float* __restrict__ offsets,
                                                              const float4* __restrict__ tab, float* __restrict__ dirs,
                                                              float* __restrict__ areas) {
    const size_t o = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (o >= (size_t)P * Ns) return;
    const size_t g = o / Ns;
    const int i = (int)(o - g * Ns);
    const LatticeFrame f = lattice_frame(normals, offsets, g);
    float d[3];
    lattice_dir(f, tab[i], i, offsets != nullptr, d);
    if (dirs) { dirs[o * 3] = d[0]; dirs[o * 3 + 1] = d[1]; dirs[o * 3 + 2] = d[2]; }
    if (areas) areas[o] = kTwoPi;
}

// F.interpolate(mode='bilinear', align_corners=False) of an [H,W,C] image to [oh,ow,C] (EnvLight.direct_light's 32x64
// down-sample, scene/envmap.py:62-63): source coordinate (dst + 0.5) * scale - 0.5 clamped at 0, neighbours clamped
__global__ void __launch_bounds__(BLOCK) resample_kernel(const float* __restrict__ src, int H, int W, int C,
                                                         float* __restrict__ dst, int oh, int ow) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= oh * ow * C) return;
    const int c = i % C, x = (i / C) % ow, y = i / (C * ow);
    const float sy = fmaxf(((float)y + 0.5f) * ((float)H / (float)oh) - 0.5f, 0.f);
    const float sx = fmaxf(((float)x + 0.5f) * ((float)W / (float)ow) - 0.5f, 0.f);
    const int y0 = min((int)sy, H - 1), x0 = min((int)sx, W - 1);
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float fy = sy - (float)y0, fx = sx - (float)x0;
    const float a = src[((size_t)y0 * W + x0) * C + c], b = src[((size_t)y0 * W + x1) * C + c];
    const float e = src[((size_t)y1 * W + x0) * C + c], f = src[((size_t)y1 * W + x1) * C + c];
    dst[i] = (a * (1.f - fx) + b * fx) * (1.f - fy) + (e * (1.f - fx) + f * fx) * fy;
}

struct GaussConst {  // per-(Gaussian, corner) constants of a phase-2 lane
    float nraw[3], Nh[3], a2, kk, nom1, fd[3], r;
    float sgn, inv_len, NoV_raw;  // for the backward
};

// The GGX denominator term  nom0 = clamp(N.H, 1e-6, 1)^2 (alpha^2 - 1) + 1  (svgss.py:612-617) for unit N (flipped to the viewer) and
// unit H.  Evaluated literally, 1 - N.H^2 cancels when N is near H -- exactly where a glossy lobe (alpha^2 down to 6.6e-5) has its
// weight -- and one ulp of N.H moves nom0 by 2e-3.  With  1 - N.H^2 = |H - (N.H) N|^2  (the part of H perpendicular to N: three FMAs
// whose results are small and carry ~1e-7 ABSOLUTE error, i.e. ~1e-5 of their own size at the lobe's width; an error of N.H itself
// enters squared)  nom0 = alpha^2 + |H - (N.H) N|^2 (1 - alpha^2)  is good to ~1e-5 where the literal form -- the reference's own fp32
// code included -- is good to ~2e-3; the lower clamp (N.H < 1e-6: H behind the surface) keeps the reference's value.
__device__ __forceinline__ float ggx_nom0(const float* N, const float* H, float a2) {
    const float noh = N[0] * H[0] + N[1] * H[1] + N[2] * H[2];
#if SHADE_PRECISE >= 2
    const float px = fmaf(-noh, N[0], H[0]), py = fmaf(-noh, N[1], H[1]), pz = fmaf(-noh, N[2], H[2]);
    const float s2 = fminf(px * px + py * py + pz * pz, 1.f);
    return noh >= 1e-6f ? fmaf(s2, 1.f - a2, a2) : 1e-12f * (a2 - 1.f) + 1.f;
#else
    const float NoH = fminf(1.f, fmaxf(1e-6f, noh));
    return NoH * NoH * (a2 - 1.f) + 1.f;
#endif
}

struct CornerIn { float v[3], n[3], r, base[3]; };   // what the constants of one (Gaussian, corner) are made from
__device__ __forceinline__ CornerIn load_corner_in(const svgir_shade_params& p, size_t g, int k) {
    CornerIn ci;
#pragma unroll
    for (int j = 0; j < 3; j++) { ci.v[j] = p.viewdirs[g * 3 + j]; ci.n[j] = p.normals[g * 12 + k * 3 + j]; ci.base[j] = p.base_color[g * 12 + j * 4 + k]; }
    ci.r = p.roughness[g * 4 + k];
    return ci;
}
__device__ __forceinline__ void corner_consts(const CornerIn& ci, const float* V, GaussConst& c) {
    const float* n = ci.n;
    c.nraw[0] = n[0]; c.nraw[1] = n[1]; c.nraw[2] = n[2];
    const float len = fmaxf(sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]), 1e-12f);
#if SHADE_PRECISE >= 1
    c.inv_len = 1.0f / len;
#else
    c.inv_len = __builtin_amdgcn_rcpf(len);
#endif
    float Nn[3] = {n[0] * c.inv_len, n[1] * c.inv_len, n[2] * c.inv_len};
    const float nov = V[0] * Nn[0] + V[1] * Nn[1] + V[2] * Nn[2];
    c.sgn = nov > 0.f ? 1.f : (nov < 0.f ? -1.f : 0.f);
    c.Nh[0] = Nn[0] * c.sgn; c.Nh[1] = Nn[1] * c.sgn; c.Nh[2] = Nn[2] * c.sgn;
    c.NoV_raw = c.Nh[0] * V[0] + c.Nh[1] * V[1] + c.Nh[2] * V[2];
    const float NoV = fminf(1.f, fmaxf(1e-6f, c.NoV_raw));
    c.r = ci.r;
    const float a = c.r * c.r;
    c.a2 = a * a;
    c.kk = (a + 2.f * c.r + 1.0f) / 8.0f;
    c.nom1 = NoV * (1.f - c.kk) + c.kk;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) c.fd[ch] = ci.base[ch] * kInvPi;
}
__device__ __forceinline__ void load_corner(const svgir_shade_params& p, size_t g, int k, const float* V, GaussConst& c) {
    corner_consts(load_corner_in(p, g, k), V, c);
}

// get_radiances (scene/gaussian_model.py:323-324): the incident radiance is nan_to_num(_radiances.detach() * _radiance_ratio, nan = 0)
// -- with svgir_shade_params.radiance_ratio the product and its clean-up happen here, on the value just loaded
struct RadRatio { float ratio; bool on; };
__device__ __forceinline__ RadRatio rad_ratio(const svgir_shade_params& p) {   // (the RATIO kernels: launched only with the pointer set)
    return RadRatio{*p.radiance_ratio, true};
}
__device__ __forceinline__ bool ratio_finite(float raw, const RadRatio& rr) {   // isfinite(raw * ratio): where nan_to_num passes its gradient
    const float v = raw * rr.ratio;
    return fabsf(v) <= 3.402823466e+38f;   // (false for NaN)
}
__device__ __forceinline__ float incident_of(float raw, const RadRatio& rr) {
    if (!rr.on) return raw;
    const float v = raw * rr.ratio;
    // torch.nan_to_num(v, nan = 0.0): +-inf -> the largest finite floats (the median of three), NaN -> 0
    const float c = __builtin_amdgcn_fmed3f(v, -3.402823466e+38f, 3.402823466e+38f);
    return v != v ? 0.f : c;
}

// the raw inputs of one incident sample (lane = sample), loadable a chunk / a Gaussian ahead of their use
struct RawSample { float d[3], rad[3], vis, area; };

__device__ __forceinline__ RawSample load_raw(const svgir_shade_params& p, size_t g, int s, int lane, int cnt) {
    const int si = s + (lane < cnt ? lane : 0);
    const size_t o = g * (size_t)p.Ns + (size_t)si;
    RawSample r;
    if (p.incident_dirs) {
#pragma unroll
        for (int j = 0; j < 3; j++) r.d[j] = p.incident_dirs[o * 3 + j];
    } else {   // directions from the lattice (3 (+1) floats per surfel instead of 12 bytes per sample)
        const LatticeFrame lf = lattice_frame(p.lattice_normals, p.lattice_offsets, g);
        lattice_dir(lf, reinterpret_cast<const float4*>(p.lattice_work)[si], si, p.lattice_offsets != nullptr, r.d);
    }
#pragma unroll
    for (int j = 0; j < 3; j++) r.rad[j] = p.radiance[o * 3 + j];
    r.vis = p.visibility[o]; r.area = p.incident_areas ? p.incident_areas[o] : kTwoPi;
    return r;
}

// phase 1 for one chunk of <= 64 samples [s0, s0+cnt) of one Gaussian: fills the wave's sample records (slot =
// sample - s0) and adds this lane's sample to the per-lane light sums m[10].
__device__ __forceinline__ void stage_samples(const svgir_shade_params& p, size_t g, int lane, const float* V,
                                              float* __restrict__ sS, float* m, int s0, int cnt, const LatticeFrame& lf, const RadRatio& rr) {
    const int Ns = p.Ns;
    if (lane < cnt) {
        const int s = s0 + lane;
        const size_t o = g * Ns + s;
        float d[3];
        if (p.incident_dirs) { d[0] = p.incident_dirs[o * 3]; d[1] = p.incident_dirs[o * 3 + 1]; d[2] = p.incident_dirs[o * 3 + 2]; }
        else lattice_dir(lf, reinterpret_cast<const float4*>(p.lattice_work)[s], s, p.lattice_offsets != nullptr, d);
        const float raw3[3] = {p.radiance[o * 3], p.radiance[o * 3 + 1], p.radiance[o * 3 + 2]};   // (one 96-bit load)
        const float rad[3] = {incident_of(raw3[0], rr), incident_of(raw3[1], rr), incident_of(raw3[2], rr)};
        const float vis = p.visibility[o], area = p.incident_areas ? p.incident_areas[o] : kTwoPi;
        const float il = inv_norm(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);   // = 1 / max(|.|, 1e-12)
        const float L[3] = {d[0] * il, d[1] * il, d[2] * il};
        float H[3] = {(L[0] + V[0]) * 0.5f, (L[1] + V[1]) * 0.5f, (L[2] + V[2]) * 0.5f};
        const float ih = inv_norm(H[0] * H[0] + H[1] * H[1] + H[2] * H[2]);   // = 1 / max(|.|, 1e-12)
        H[0] *= ih; H[1] *= ih; H[2] *= ih;
        const float VoH = fminf(1.f, fmaxf(1e-6f, V[0] * H[0] + V[1] * H[1] + V[2] * H[2]));
        const float frac0 = 0.04f + (1.f - 0.04f) * __builtin_amdgcn_exp2f((-5.55473f * VoH - 6.98316f) * VoH);
        EnvTap t;
        float dl[3] = {d[0], d[1], d[2]};
        if (p.env_transform) {
            const float* m = p.env_transform;
#pragma unroll
            for (int j = 0; j < 3; j++) dl[j] = m[3 * j] * d[0] + m[3 * j + 1] * d[1] + m[3 * j + 2] * d[2];
        }
        env_taps(dl, p.env_h, p.env_w, t);
        float E[3] = {0.f, 0.f, 0.f};
        {   // 12 unconditional gathers (out-of-range taps: texel 0 with weight 0), one memory latency for all of them
            float tex[4][3];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float4 tq = reinterpret_cast<const float4*>(p.env_work)[t.idx[j] >= 0 ? t.idx[j] : 0];
                tex[j][0] = tq.x; tex[j][1] = tq.y; tex[j][2] = tq.z;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float w = t.idx[j] >= 0 ? t.w[j] : 0.f;
#pragma unroll
                for (int ch = 0; ch < 3; ch++) E[ch] += w * tex[j][ch];
            }
        }
        float* r = sS + lane * SREC;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const float Lg = fminf(64.f, fmaxf(0.f, E[ch] * p.env_scale)) * vis;
            r[ch] = d[ch]; r[3 + ch] = L[ch]; r[6 + ch] = H[ch];
            r[10 + ch] = Lg; r[13 + ch] = rad[ch];
            m[3 + ch] += rad[ch]; m[6 + ch] += Lg;
        }
        r[9] = frac0; r[16] = area;
        m[9] += vis;
    }
}

// Subset launches: the packed rows of the surfels OUTSIDE the subset are zero-filled by the same kernel -- every wave takes a strided share
// of the partition's back (svgir_shade_params.subset) before its own work; stores nobody waits for.
__device__ __forceinline__ void zero_rest_rows(const ShadeArgs& a, size_t wave_id, size_t nwaves, size_t nsel, int lane) {
    const svgir_shade_params& p = a.p;
    const size_t nrest = (size_t)p.P - nsel;
    const int nf = p.training ? 4 : 7, nvf = p.training ? 52 : 64;
    for (size_t j = wave_id; j < nrest; j += nwaves) {
        const size_t g = p.subset[(size_t)p.P - 1 - j];
        if (a.reduced) { a.reduced[g * NRED + lane] = 0.f; if (lane + 64 < NRED) a.reduced[g * NRED + 64 + lane] = 0.f; }
        if (a.features && lane < nf) a.features[g * nf + lane] = 0.f;
        if (a.vfeatures && lane < nvf) a.vfeatures[g * nvf + lane] = 0.f;
    }
}

#ifndef SHADE_FWPE
#define SHADE_FWPE 5   // 96 VGPRs without spills; the default heuristic settles on 107 (4 waves per SIMD): 285 -> 252 us at P = 200k, Ns = 64
#endif
// One wave per Gaussian.  (A persistent variant that loads the next Gaussian's samples and corner data while the current one
// is processed was built and measured: 282 us at 4 waves/SIMD, 332 us at 5 (spills) against 250 us for this one.)
// (RATIO: svgir_shade_params.radiance_ratio is set -- a template parameter so that the plain kernels are exactly the code they were)
template <bool RATIO>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(SHADE_FWPE, SHADE_FWPE))) shade_fwd_kernel(const ShadeArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const svgir_shade_params& p = a.p;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int Ns = p.Ns;
    const RadRatio rr = RATIO ? rad_ratio(p) : RadRatio{1.f, false};
    float* sS = smem + (size_t)wave * (64 * SREC + 80 + 32);
    float* sOut = sS + 64 * SREC;
    float* sIn = sOut + 80;   // base_color[12] | normals[12] | roughness[4] of this Gaussian, for the packing in the epilogue
    // work item -> surfel: all P of them, or the front of the caller's partition (svgir_shade_params.subset)
    const size_t wi = (size_t)blockIdx.x * 4 + wave;
    const size_t nwork = p.subset ? (size_t)min(*p.subset_count, (uint32_t)p.P) : (size_t)p.P;
    if (p.subset) zero_rest_rows(a, wi, (size_t)gridDim.x * 4, nwork, lane);
    if ((size_t)blockIdx.x * 4 >= nwork) return;   // (whole workgroup past the end of the list)
    const bool valid = wi < nwork;
    const size_t g = valid ? (p.subset ? (size_t)p.subset[wi] : wi) : 0;
    const size_t gg = g;
    const float inv_ns = 1.f / (float)Ns;

    float V[3] = {p.viewdirs[gg * 3], p.viewdirs[gg * 3 + 1], p.viewdirs[gg * 3 + 2]};
    {
        const float iv = inv_norm(V[0] * V[0] + V[1] * V[1] + V[2] * V[2]);   // = 1 / max(|.|, 1e-12)
        V[0] *= iv; V[1] *= iv; V[2] *= iv;
    }
    DEV_TRACE_DECL();
    if (a.vfeatures && lane < 28) {   // fetched with the rest of the set-up loads: the epilogue then waits for nothing
        const float* src = lane < 12 ? p.base_color + gg * 12 + lane : (lane < 24 ? p.normals + gg * 12 + (lane - 12) : p.roughness + gg * 4 + (lane - 24));
        sIn[lane] = *src;
    }
    float m[10];
#pragma unroll
    for (int i = 0; i < 10; i++) m[i] = 0.f;
    const int k = lane >> 4, sg = lane & 15;
    GaussConst c;
    load_corner(p, gg, k, V, c);
    // per channel four sums: A_d = sum(Lg*ge), A_l = sum(Ll*ge), B_d = sum(fs*Lg*ge), B_l = sum(fs*Ll*ge); with
    // f = f_d + fs the five outputs are  diffuse = A_d + A_l, specular = B_d + B_l, direct = f_d*A_d + B_d,
    // indirect = f_d*A_l + B_l, pbr = direct + indirect
    float Ad[3] = {0, 0, 0}, Al[3] = {0, 0, 0}, Bd[3] = {0, 0, 0}, Bl[3] = {0, 0, 0};
    LatticeFrame lf = {};
    if (!p.incident_dirs) lf = lattice_frame(p.lattice_normals, p.lattice_offsets, gg);
    for (int s0 = 0; s0 < Ns; s0 += 64) {
      const int cnt = min(64, Ns - s0);
      wave_lds_sync();   // previous chunk consumed
      DEV_TRACE_MARK(0);
      stage_samples(p, gg, lane, V, sS, m, s0, cnt, lf, rr);
      wave_lds_sync();
      DEV_TRACE_MARK(1);
      for (int s = sg; s < cnt; s += 16) {
        const float* r = sS + s * SREC;
        const float ndi = fmaxf(c.nraw[0] * r[0] + c.nraw[1] * r[1] + c.nraw[2] * r[2], 0.f);
        const float NoL = fminf(1.f, fmaxf(1e-6f, c.Nh[0] * r[3] + c.Nh[1] * r[4] + c.Nh[2] * r[5]));
        const float nom0 = ggx_nom0(c.Nh, r + 6, c.a2);
        const float nom = fminf(4.f * kPi, fmaxf(1e-6f, 4.f * kPi * nom0 * nom0 * c.nom1 * (NoL * (1.f - c.kk) + c.kk)));
        const float fs = r[9] * c.a2 * __builtin_amdgcn_rcpf(nom);
        const float ge = r[16] * ndi;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const float td = r[10 + ch] * ge, tl = r[13 + ch] * ge;
            Ad[ch] += td; Al[ch] += tl; Bd[ch] += fs * td; Bl[ch] += fs * tl;
        }
      }
    }
    DEV_TRACE_MARK(2);
#pragma unroll
    for (int i = 3; i < 10; i++) {   // local(3), global(3), visibility; incident = local + global
        const float t = wave_scan_last(m[i]);
        if (lane == 63) sOut[60 + i] = t * inv_ns;
    }
    if (lane == 63) {
#pragma unroll
        for (int i = 0; i < 3; i++) sOut[60 + i] = sOut[63 + i] + sOut[66 + i];
    }
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        const float ad = row16_sum(Ad[ch]), al = row16_sum(Al[ch]), bd = row16_sum(Bd[ch]), bl = row16_sum(Bl[ch]);
        const float v1 = ad + al, v2 = bd + bl, v3 = c.fd[ch] * ad + bd, v4 = c.fd[ch] * al + bl, v0 = v3 + v4;
        if (sg == 0) {
            sOut[0 + ch * 4 + k] = v0 * inv_ns; sOut[12 + ch * 4 + k] = v1 * inv_ns; sOut[24 + ch * 4 + k] = v2 * inv_ns;
            sOut[36 + ch * 4 + k] = v3 * inv_ns; sOut[48 + ch * 4 + k] = v4 * inv_ns;
        }
    }
    wave_lds_sync();
    if (!valid) return;
    // ---- epilogue: consecutive lanes write consecutive floats ----
    if (a.reduced) {
        a.reduced[g * NRED + lane] = sOut[lane];
        if (lane + 64 < NRED) a.reduced[g * NRED + 64 + lane] = sOut[64 + lane];
    }
    if (a.features) {
        if (p.training) {
            if (lane < 4) a.features[g * 4 + lane] = lane == 0 ? sOut[69] : sOut[63 + (lane - 1)];
        } else {
            if (lane < 7) a.features[g * 7 + lane] = lane < 3 ? sOut[60 + lane] : (lane < 6 ? sOut[63 + (lane - 3)] : sOut[69]);
        }
    }
    if (a.vfeatures) {
        const int VS = p.training ? 52 : 64;
        if (lane < VS) {
            float v;
            if (lane < 12) v = sOut[lane];
            else if (lane < 24) v = sIn[lane - 12];
            else if (lane < 36) {
                const int e = lane - 24, ch = e >> 2, kc = e & 3;  // channel*4 + corner
                const float* n = sIn + 12 + kc * 3;
                v = n[0] * p.viewmatrix[0 * 4 + ch] + n[1] * p.viewmatrix[1 * 4 + ch] + n[2] * p.viewmatrix[2 * 4 + ch];
            } else if (lane < 40) v = sIn[24 + (lane - 36)];
            else if (lane < 52) v = p.training ? sOut[12 + (lane - 40)] : sOut[36 + (lane - 40)];
            else v = sOut[48 + (lane - 52)];
            a.vfeatures[g * VS + lane] = v;
        }
    }
    DEV_TRACE_MARK(3);
    DEV_TRACE_END(1, 1u, (unsigned)Ns, 0u);
}

// ----------------------------------------------------------------------------------------------------------------
// forward, quad layout (round 4): lane = (surfel q = lane / 4 of the wave's 16 consecutive surfels, corner k = lane % 4).
//
// The one-wave-per-surfel kernel above pays its fixed costs 200 k times: every lane repeats the surfel's prologue (view vector,
// lattice frame incl. a sincos, corner constants 16x redundantly), and every surfel ends in 7 wave reductions + 12 row reductions
// + a strip epilogue -- PMC: 885 VALU instructions per surfel at Ns = 64, of which the corner x sample loop is ~6 % of the wave's
// life.  Here a lane owns ONE (surfel, corner) for all of its samples:
//   * corner constants and the 12 light sums (A_d, A_l, B_d, B_l per channel) live in the lane's registers: no cross-lane
//     reduction for any of the 60 corner outputs;
//   * the corner-independent part of a sample (direction from the lattice, |d|, half vector, Schlick term, env lookup with
//     acos / atan2 and 12 gathers) is computed ONCE per sample: in every step of 4 samples lane k of the quad stages sample
//     4 i + k, then the quad walks the four samples, each staged record (12 floats) broadcast from its lane with quad_perm DPP;
//   * H is never formed: N.H = (N.L + N.V) / (2 |(L + V) / 2|) needs the scalar 1 / |(L + V) / 2| only;
//   * the per-surfel light means are four partial sums per quad (two DPP adds), and the outputs of the wave's 16 surfels go through
//     one LDS strip so that `reduced`, `features` and `vfeatures` -- contiguous over consecutive surfels -- leave as full rows.
// ~400 instructions per surfel at Ns = 64 instead of 885; 16 surfels per wave keep 16 x (48 + 16) bytes of samples in flight per step.
#ifndef SHADE_FQ_WPE
#define SHADE_FQ_WPE 4
#endif
#ifndef SHADE_FQ_WPE_LO
#define SHADE_FQ_WPE_LO 3   // "3 to 4 waves per SIMD": the allocator takes the 145-148 VGPRs the kernel wants (three waves, nothing spilled); held to 128
                            // it spills 8-11 dwords inside the sample loop: 141 -> 129 us on the 88.7 k-surfel working set of cfg3_train.  (The
                            // one-wave-per-surfel kernel above is the opposite case: 5 waves with 2 spilled dwords beat 4 without, 277 vs 315 us.)
#endif
constexpr int FQ_SURF = 16;                      // surfels per wave
constexpr int FQ_ROW = NRED + 2;                 // LDS floats per surfel: the 70 reduced outputs (+ pad: rows of a quad group on distinct banks)
constexpr int FQ_IN = 28;                        // base_color[12] | normals[12] | roughness[4] of a surfel (for the packing)
template <int J>
__device__ __forceinline__ float quad_bcast(float v) {   // value of lane (quad base + J) in all four lanes of the quad
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), J | (J << 2) | (J << 4) | (J << 6), 0xf, 0xf, false));
}

struct FqSample { float d[3], il, ih, frac0, LgA[3], LlA[3]; };   // the staged, corner-independent part of one incident sample

template <bool RATIO>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(SHADE_FQ_WPE_LO, SHADE_FQ_WPE))) shade_fwd_quad_kernel(const ShadeArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const svgir_shade_params& p = a.p;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int Ns = p.Ns;
    float* sOut = smem + (size_t)wave * (FQ_SURF * (FQ_ROW + FQ_IN) + FQ_SURF);
    float* sIn = sOut + FQ_SURF * FQ_ROW;
    uint32_t* sId = reinterpret_cast<uint32_t*>(sIn + FQ_SURF * FQ_IN);   // surfel of each of the wave's work items (subset launches)
    const int q = lane >> 2, k = lane & 3;
    // work items -> surfels: all P of them, or the front of the caller's partition (svgir_shade_params.subset)
    const size_t nwork = p.subset ? (size_t)min(*p.subset_count, (uint32_t)p.P) : (size_t)p.P;
    const size_t g0 = ((size_t)blockIdx.x * 4 + wave) * FQ_SURF;       // first work item of the wave
    if (p.subset) zero_rest_rows(a, (size_t)blockIdx.x * 4 + wave, (size_t)gridDim.x * 4, nwork, lane);
    if (g0 >= nwork) return;
    const int nsurf = (int)min((size_t)FQ_SURF, nwork - g0);
    const bool valid = q < nsurf;
    const size_t wi = g0 + (valid ? q : 0);
    const size_t gg = p.subset ? (size_t)p.subset[wi] : wi;
    if (p.subset && k == 0) sId[q] = (uint32_t)gg;
    const float inv_ns = 1.f / (float)Ns;

    float V[3] = {p.viewdirs[gg * 3], p.viewdirs[gg * 3 + 1], p.viewdirs[gg * 3 + 2]};
    {
        const float iv = inv_norm(V[0] * V[0] + V[1] * V[1] + V[2] * V[2]);   // = 1 / max(|.|, 1e-12)
        V[0] *= iv; V[1] *= iv; V[2] *= iv;
    }
    GaussConst c;
    {
        const CornerIn ci = load_corner_in(p, gg, k);
        corner_consts(ci, V, c);
        {   // the quad holds exactly the surfel's 28 packing inputs (also the epilogue's source of the diffuse albedo)
            float* si = sIn + q * FQ_IN;
#pragma unroll
            for (int j = 0; j < 3; j++) { si[j * 4 + k] = ci.base[j]; si[12 + k * 3 + j] = ci.n[j]; }
            si[24 + k] = ci.r;
        }
    }
    const bool lattice = !p.incident_dirs;
    const bool has_off = p.lattice_offsets != nullptr;
    LatticeFrame lf = {};
    if (lattice) lf = lattice_frame(p.lattice_normals, p.lattice_offsets, gg);
    const float4* ltab = reinterpret_cast<const float4*>(p.lattice_work);
    float m[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};     // this lane's samples: local light (3), global light (3), visibility
    float Ad[3] = {0, 0, 0}, Al[3] = {0, 0, 0}, Bd[3] = {0, 0, 0}, Bl[3] = {0, 0, 0};
    const float omk = 1.f - c.kk, a2m1 = c.a2 - 1.f;

    // raw inputs of the lane's sample of a step, fetched one step ahead
    struct Raw { float d[3], rad[3], vis, area; };
    const RadRatio rrq = RATIO ? rad_ratio(p) : RadRatio{1.f, false};
    auto load_raw_q = [&](int s) -> Raw {
        Raw r;
        const int sc = min(s, Ns - 1);
        const size_t o = gg * (size_t)Ns + (size_t)sc;
        if (!lattice) { r.d[0] = p.incident_dirs[o * 3]; r.d[1] = p.incident_dirs[o * 3 + 1]; r.d[2] = p.incident_dirs[o * 3 + 2]; }
        else { r.d[0] = r.d[1] = r.d[2] = 0.f; }
        r.rad[0] = p.radiance[o * 3]; r.rad[1] = p.radiance[o * 3 + 1]; r.rad[2] = p.radiance[o * 3 + 2];
        r.vis = p.visibility[o];
        r.area = p.incident_areas ? p.incident_areas[o] : kTwoPi;
        return r;
    };
    Raw raw = load_raw_q(k);
    for (int s0 = 0; s0 < Ns; s0 += 4) {
        const int s = s0 + k;
        const bool act = s < Ns;
        Raw x = raw;
        raw = load_raw_q(s + 4);
        if (RATIO) {   // (at the values' USE, not at their load: the loads above are a step ahead)
#pragma unroll
            for (int ch = 0; ch < 3; ch++) x.rad[ch] = incident_of(x.rad[ch], rrq);
        }
        // ---- stage the lane's sample (stage_samples above, without the record) ----
        FqSample f;
        {
            float d[3] = {x.d[0], x.d[1], x.d[2]};
            if (lattice) lattice_dir(lf, ltab[min(s, Ns - 1)], min(s, Ns - 1), has_off, d);
            const float il = inv_norm(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);   // = 1 / max(|.|, 1e-12)
            const float L[3] = {d[0] * il, d[1] * il, d[2] * il};
            float H[3] = {(L[0] + V[0]) * 0.5f, (L[1] + V[1]) * 0.5f, (L[2] + V[2]) * 0.5f};
            const float ih = inv_norm(H[0] * H[0] + H[1] * H[1] + H[2] * H[2]);   // = 1 / max(|.|, 1e-12)
            const float VoH = fminf(1.f, fmaxf(1e-6f, (V[0] * H[0] + V[1] * H[1] + V[2] * H[2]) * ih));
            f.frac0 = 0.04f + (1.f - 0.04f) * __builtin_amdgcn_exp2f((-5.55473f * VoH - 6.98316f) * VoH);
            EnvTap t;
            float dl[3] = {d[0], d[1], d[2]};
            if (p.env_transform) {
                const float* mt = p.env_transform;
#pragma unroll
                for (int j = 0; j < 3; j++) dl[j] = mt[3 * j] * d[0] + mt[3 * j + 1] * d[1] + mt[3 * j + 2] * d[2];
            }
            env_taps(dl, p.env_h, p.env_w, t);
            float E[3] = {0.f, 0.f, 0.f};
            {   // 12 unconditional gathers (out-of-range taps: texel 0 with weight 0), one memory latency for all of them
                float tex[4][3];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float4 tq = reinterpret_cast<const float4*>(p.env_work)[t.idx[j] >= 0 ? t.idx[j] : 0];
                    tex[j][0] = tq.x; tex[j][1] = tq.y; tex[j][2] = tq.z;
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float w = t.idx[j] >= 0 ? t.w[j] : 0.f;
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) E[ch] += w * tex[j][ch];
                }
            }
            const float area = act ? x.area : 0.f;   // samples beyond Ns (Ns not a multiple of 4) carry weight 0
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const float Lg = fminf(64.f, fmaxf(0.f, E[ch] * p.env_scale)) * x.vis;
                f.d[ch] = d[ch];
                f.LgA[ch] = Lg * area; f.LlA[ch] = x.rad[ch] * area;
                if (act) { m[ch] += x.rad[ch]; m[3 + ch] += Lg; }
            }
            if (act) m[6] += x.vis;
            f.il = il; f.ih = ih;
        }
        // ---- the quad's four samples against this lane's corner ----
        auto corner = [&](const FqSample& r) {
            const float ndi = fmaxf(c.nraw[0] * r.d[0] + c.nraw[1] * r.d[1] + c.nraw[2] * r.d[2], 0.f);
            const float nl = (c.Nh[0] * r.d[0] + c.Nh[1] * r.d[1] + c.Nh[2] * r.d[2]) * r.il;     // Nh . L
            const float NoL = fminf(1.f, fmaxf(1e-6f, nl));
            const float nohr = (nl + c.NoV_raw) * 0.5f * r.ih;                                      // Nh . H,  H = (L + V) / 2 * ih
#if SHADE_PRECISE >= 2
            // 1 - (Nh . H)^2 = |H - (Nh . H) Nh|^2 with 2 H / ih = d il + V (see ggx_nom0)
            const float nohu = nl + c.NoV_raw;
            const float px = fmaf(-nohu, c.Nh[0], fmaf(r.d[0], r.il, V[0])), py = fmaf(-nohu, c.Nh[1], fmaf(r.d[1], r.il, V[1])),
                        pz = fmaf(-nohu, c.Nh[2], fmaf(r.d[2], r.il, V[2]));
            const float hs = 0.5f * r.ih;
            const float s2 = fminf((px * px + py * py + pz * pz) * (hs * hs), 1.f);
            const float nom0 = nohr >= 1e-6f ? fmaf(s2, -a2m1, c.a2) : 1e-12f * a2m1 + 1.f;
#else
            const float NoH = fminf(1.f, fmaxf(1e-6f, nohr));
            const float nom0 = NoH * NoH * a2m1 + 1.f;
#endif
            const float nom = fminf(4.f * kPi, fmaxf(1e-6f, 4.f * kPi * nom0 * nom0 * c.nom1 * (NoL * omk + c.kk)));
            const float fs = r.frac0 * c.a2 * __builtin_amdgcn_rcpf(nom);
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const float td = r.LgA[ch] * ndi, tl = r.LlA[ch] * ndi;
                Ad[ch] += td; Al[ch] += tl; Bd[ch] += fs * td; Bl[ch] += fs * tl;
            }
        };
#define FQ_STEP(J)                                                                                                       \
        {                                                                                                                \
            FqSample r;                                                                                                  \
            r.d[0] = quad_bcast<J>(f.d[0]); r.d[1] = quad_bcast<J>(f.d[1]); r.d[2] = quad_bcast<J>(f.d[2]);              \
            r.il = quad_bcast<J>(f.il); r.ih = quad_bcast<J>(f.ih); r.frac0 = quad_bcast<J>(f.frac0);                    \
            r.LgA[0] = quad_bcast<J>(f.LgA[0]); r.LgA[1] = quad_bcast<J>(f.LgA[1]); r.LgA[2] = quad_bcast<J>(f.LgA[2]);  \
            r.LlA[0] = quad_bcast<J>(f.LlA[0]); r.LlA[1] = quad_bcast<J>(f.LlA[1]); r.LlA[2] = quad_bcast<J>(f.LlA[2]);  \
            corner(r);                                                                                                   \
        }
        FQ_STEP(0) FQ_STEP(1) FQ_STEP(2) FQ_STEP(3)
#undef FQ_STEP
    }
    // ---- per-surfel results into the wave's LDS strip ----
    {
        float* so = sOut + q * FQ_ROW;
#pragma unroll
        for (int i = 0; i < 7; i++) m[i] = quad_sum(m[i]);
        if (k == 0) {
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const float loc = m[ch] * inv_ns, glo = m[3 + ch] * inv_ns;
                so[60 + ch] = loc + glo; so[63 + ch] = loc; so[66 + ch] = glo;
            }
            so[69] = m[6] * inv_ns;
        }
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const float ad = Ad[ch], al = Al[ch], bd = Bd[ch], bl = Bl[ch];
            const float fd = sIn[q * FQ_IN + ch * 4 + k] * kInvPi;   // (= c.fd[ch], not held across the sample loop)
            const float v1 = ad + al, v2 = bd + bl, v3 = fd * ad + bd, v4 = fd * al + bl, v0 = v3 + v4;
            so[0 + ch * 4 + k] = v0 * inv_ns; so[12 + ch * 4 + k] = v1 * inv_ns; so[24 + ch * 4 + k] = v2 * inv_ns;
            so[36 + ch * 4 + k] = v3 * inv_ns; so[48 + ch * 4 + k] = v4 * inv_ns;
        }
    }
    wave_lds_sync();
    // ---- epilogue: the rows of the wave's surfels are contiguous in every output (all-P launches; a subset's rows are scattered,
    // each still written by consecutive lanes): consecutive lanes write consecutive floats ----
    const bool scat = p.subset != nullptr;
    auto row_of = [&](int qq) -> size_t { return scat ? (size_t)sId[qq] : g0 + (size_t)qq; };
    if (a.reduced) {
        for (int i = lane; i < nsurf * NRED; i += 64) { const int qq = i / NRED, e = i - qq * NRED; a.reduced[row_of(qq) * NRED + e] = sOut[qq * FQ_ROW + e]; }
    }
    if (a.features) {
        if (p.training) {
            if (lane < nsurf * 4) { const int qq = lane >> 2, e = lane & 3; a.features[row_of(qq) * 4 + e] = sOut[qq * FQ_ROW + (e == 0 ? 69 : 63 + (e - 1))]; }
        } else {
            for (int i = lane; i < nsurf * 7; i += 64) {
                const int qq = i / 7, e = i - qq * 7;
                a.features[row_of(qq) * 7 + e] = sOut[qq * FQ_ROW + (e < 3 ? 60 + e : (e < 6 ? 63 + (e - 3) : 69))];
            }
        }
    }
    if (a.vfeatures) {
        auto pack = [&](int qq, int e, bool training) -> float {
            const float* so = sOut + qq * FQ_ROW;
            const float* si = sIn + qq * FQ_IN;
            if (e < 12) return so[e];
            if (e < 24) return si[e - 12];
            if (e < 36) {
                const int ee = e - 24, ch = ee >> 2, kc = ee & 3;  // channel*4 + corner
                const float* n = si + 12 + kc * 3;
                return n[0] * p.viewmatrix[0 * 4 + ch] + n[1] * p.viewmatrix[1 * 4 + ch] + n[2] * p.viewmatrix[2 * 4 + ch];
            }
            if (e < 40) return si[24 + (e - 36)];
            if (e < 52) return training ? so[12 + (e - 40)] : so[36 + (e - 40)];
            return so[48 + (e - 52)];
        };
        if (p.training) {
            for (int i = lane; i < nsurf * 52; i += 64) { const int qq = i / 52, e = i - qq * 52; a.vfeatures[row_of(qq) * 52 + e] = pack(qq, e, true); }
        } else {
            for (int i = lane; i < nsurf * 64; i += 64) a.vfeatures[row_of(i >> 6) * 64 + (i & 63)] = pack(i >> 6, i & 63, false);
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------
// backward
// ----------------------------------------------------------------------------------------------------------------
struct ShadeBwdArgs {
    svgir_shade_params p;
    const float* g_red;          // may be null when g_feat / g_vfeat carry the upstream gradient
    const float *g_feat, *g_vfeat;  // optional: gradients w.r.t. the packed features [P,S] / vfeatures [P,VS]
    float *d_base, *d_rough, *d_normals, *d_radiance, *d_envtab;  // d_envtab: gradient w.r.t. the f(env) table; d_radiance may be null (radiance_ratio)
    float* ratio_part;   // [gridDim.x] or null: per-workgroup partial sums of dL/d(radiance_ratio)
    int zero_rest;   // subset launches: the waves also zero-fill the gradient rows of the surfels OUTSIDE the subset (a few rows per processed
                     // surfel: stores nobody waits for, in a kernel that is bound by instruction issue)
};

// ---- backward sample record (BREC = 20 floats = five float4, written and read with 128-bit LDS instructions):
//   d(3) il | H(3) frac0 | Lg(3) area | Ll(3) ve | footprint, fx, fy, pad
// il = 1 / |d| (L = d * il is three multiplies in the consumer instead of three floats here), ve = visibility * env_scale =
// d(global light) / d(env lookup) where the lookup is inside its clamp -- the three per-channel clamp flags ride in the
// footprint word {x0 + 1 : 14 bits | y0 + 1 : 14 bits | flags : 3 bits}, so the adjoint never re-evaluates acos / atan2.
// (20 floats instead of 24: 435 -> 427 us.)
constexpr int BREC = 20;
#ifndef SHADE_BWAVES
#define SHADE_BWAVES 12   // one workgroup per CU: 12 x (5 KB of sample records + 3.3 KB of surfel records) + the fp64 env-gradient image (48 KB at 32x64); 16 waves with KB_G = 2 fit as well and are slower (467 vs 427 us)
#endif
#ifndef SHADE_BWPE
#define SHADE_BWPE 3
#endif
#ifndef SHADE_KB_G
#define SHADE_KB_G 4
#endif
constexpr int BWAVES = SHADE_BWAVES;   // waves per backward workgroup
constexpr int KB_G = SHADE_KB_G, KREC = 52;   // Gaussians prepared per batch; floats per (Gaussian, corner) record (13 float4)

__device__ __forceinline__ void stage_raw_bwd(const svgir_shade_params& p, const RawSample& x, int lane, const float* V,
                                              float* __restrict__ sS, const RadRatio& rr) {
    const float* d = x.d;
    const float il = inv_norm(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);   // = 1 / max(|.|, 1e-12)
    const float L[3] = {d[0] * il, d[1] * il, d[2] * il};
    float H[3] = {(L[0] + V[0]) * 0.5f, (L[1] + V[1]) * 0.5f, (L[2] + V[2]) * 0.5f};
    const float ih = inv_norm(H[0] * H[0] + H[1] * H[1] + H[2] * H[2]);   // = 1 / max(|.|, 1e-12)
    H[0] *= ih; H[1] *= ih; H[2] *= ih;
    const float VoH = fminf(1.f, fmaxf(1e-6f, V[0] * H[0] + V[1] * H[1] + V[2] * H[2]));
    const float frac0 = 0.04f + (1.f - 0.04f) * __builtin_amdgcn_exp2f((-5.55473f * VoH - 6.98316f) * VoH);
    // bilinear footprint (same arithmetic as env_taps)
    const int He = p.env_h, We = p.env_w;
    float dl[3] = {d[0], d[1], d[2]};
    if (p.env_transform) {
        const float* m = p.env_transform;
#pragma unroll
        for (int j = 0; j < 3; j++) dl[j] = m[3 * j] * d[0] + m[3 * j + 1] * d[1] + m[3 * j + 2] * d[2];
    }
    const float phi = acosf(dl[2]) - 1e-6f;
    const float theta = atan2f(dl[1], dl[0]);
    const float gy = phi * (2.f * kInvPi) - 1.f;
    const float gx = -theta * kInvPi;
    const float xx = (gx + 1.f) * 0.5f * (float)(We - 1);
    const float yy = (gy + 1.f) * 0.5f * (float)(He - 1);
    const float x0f = floorf(xx), y0f = floorf(yy);
    const float fx = xx - x0f, fy = yy - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
    float E[3] = {0.f, 0.f, 0.f};
    {   // 12 unconditional gathers (out-of-range taps: texel 0 with weight 0), one memory latency for all of them
        float tex[4][3], tw[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int xi = x0 + (j & 1), yi = y0 + (j >> 1);
            const bool ok = xi >= 0 && xi < We && yi >= 0 && yi < He;
            tw[j] = ok ? ((j & 1) ? fx : 1.f - fx) * ((j >> 1) ? fy : 1.f - fy) : 0.f;
            const float4 tq = reinterpret_cast<const float4*>(p.env_work)[ok ? yi * We + xi : 0];
            tex[j][0] = tq.x; tex[j][1] = tq.y; tex[j][2] = tq.z;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
            for (int ch = 0; ch < 3; ch++) E[ch] += tw[j] * tex[j][ch];
        }
    }
    float r[BREC];
    uint32_t flags = 0;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        const float Es = E[ch] * p.env_scale;
        r[ch] = d[ch]; r[4 + ch] = H[ch];
        r[8 + ch] = fminf(64.f, fmaxf(0.f, Es)) * x.vis; r[12 + ch] = incident_of(x.rad[ch], rr);
        flags |= (Es >= 0.f && Es <= 64.f) ? (1u << ch) : 0u;
    }
    r[3] = il; r[7] = frac0; r[11] = x.area; r[15] = x.vis * p.env_scale;
    // footprint origin, biased by +1 (x0, y0 >= -1 by construction), 14 bits each (the launcher checks the map's size)
    const int xb = min(max(x0 + 1, 0), 16383), yb = min(max(y0 + 1, 0), 16383);
    r[16] = __builtin_bit_cast(float, (uint32_t)xb | ((uint32_t)yb << 14) | (flags << 28));
    r[17] = fx; r[18] = fy;
    {   // radiance_ratio: which channels' products were not finite (nan_to_num passes no gradient there), bits 0..2
        uint32_t nf = 0;
        if (rr.on) {
#pragma unroll
            for (int ch = 0; ch < 3; ch++) nf |= ratio_finite(x.rad[ch], rr) ? 0u : (1u << ch);
        }
        r[19] = __builtin_bit_cast(float, nf);
    }
    float4* o = reinterpret_cast<float4*>(sS + lane * BREC);
#pragma unroll
    for (int i = 0; i < BREC / 4; i++) o[i] = make_float4(r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]);
}

// sum over the 16 lanes of the wave that share (lane & 3); result in all of them
__device__ __forceinline__ float stride4_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));  // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// Persistent workgroups of BWAVES waves; every wave owns one Gaussian at a time and never synchronises with the other
// waves (its sample records are wave-private LDS).  Lane = (sample group sg = lane / 4, corner k = lane % 4): the
// four corners of one incident sample sit in one DPP quad, so the per-sample light gradients are summed with two
// quad_perm adds and the quad then shares the adjoint of the sample: lane k stores channel k of dL/dradiance and
// scatters bilinear tap k of the env-lookup gradient into the workgroup-private LDS image (flushed with one global
// atomic per texel per workgroup).  That image is fp64: on gfx950 ds_add_f64 runs at the rate of the integer LDS atomics
// (16 cycles per wave instruction per CU, scripts/probes/lds_atomic_probe.hip) while ds_add_f32 takes 128 -- with fp32
// the 12 atomic instructions per 64 samples were ~500 of the kernel's 552 us of LDS time, hidden behind nothing.  (And
// the per-workgroup partial sums are exact to fp32 precision whatever the order of the adds.)  The next chunk of samples
// is prefetched into registers while the current one is processed.
template <bool RATIO>
__global__ void __launch_bounds__(BWAVES * 64) __attribute__((amdgpu_waves_per_eu(SHADE_BWPE, SHADE_BWPE))) shade_bwd_kernel(const ShadeBwdArgs a, int env_in_lds) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const svgir_shade_params& p = a.p;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int Ns = p.Ns, We = p.env_w, He = p.env_h;
    const int ntex = He * We * 3;
    float* sS = smem + (size_t)wave * (64 * BREC);
    float* sK = smem + (size_t)BWAVES * (64 * BREC) + (size_t)wave * (4 * KB_G * KREC);   // this wave's batch of per-(Gaussian, corner) records
    double* sEnv = reinterpret_cast<double*>(smem + (size_t)BWAVES * (64 * BREC + 4 * KB_G * KREC));   // [ntex] (only when env_in_lds)
    if (env_in_lds) {
        for (int i = threadIdx.x; i < ntex; i += BWAVES * 64) sEnv[i] = 0.0;
    }
    __syncthreads();
    const float inv_ns = 1.f / (float)Ns;
    const int k = lane & 3, sg = lane >> 2;
    // work items: all P surfels, or the front of the caller's partition (svgir_shade_params.subset); `g` counts work items, sid(g) is
    // the surfel
    const int P = p.subset ? (int)min(*p.subset_count, (uint32_t)p.P) : p.P;
    const uint32_t* __restrict__ sub = p.subset;
    auto sid = [&](int w) -> size_t { return sub ? (size_t)sub[w] : (size_t)w; };
    const int gstep = (int)gridDim.x * BWAVES;
    int g = (int)blockIdx.x * BWAVES + wave;   // wave-uniform

    const bool tr = p.training != 0;
    const int nvf = tr ? 52 : 64, nf = tr ? 4 : 7;
    // zero rows of the partition's back (see ShadeBwdArgs::zero_rest): this wave's share is every gstep-th of them
    const uint32_t nrest = (sub && a.zero_rest) ? (uint32_t)(p.P - P) : 0u;
    uint32_t zj = (uint32_t)g;
    const int zquota = (int)((nrest / (uint32_t)gstep + 1u) / (uint32_t)max(1, P / gstep) + 1u);   // rows per processed surfel
    auto zero_rows = [&](int n) {
        for (int i = 0; i < n && zj < nrest; i++, zj += (uint32_t)gstep) {
            const size_t gz = (size_t)sub[(uint32_t)p.P - 1u - zj];
            if (lane < 12) { a.d_base[gz * 12 + lane] = 0.f; a.d_normals[gz * 12 + lane] = 0.f; }
            if (lane < 4) a.d_rough[gz * 4 + lane] = 0.f;
            if (a.d_radiance) {
                float* row = a.d_radiance + gz * (size_t)(3 * Ns);
                for (int e = lane; e < 3 * Ns; e += 64) row[e] = 0.f;
            }
        }
    };
    DEV_TRACE_DECL();
    [[maybe_unused]] unsigned dev_n = 0;
    // (the scalar is re-read where it is used -- twice per chunk, scalar-cache hits -- instead of living in an SGPR across the kernel:
    // the kernel sits at its SGPR limit)
    auto ratio_now = [&]() -> RadRatio {
        if (!RATIO) return RadRatio{1.f, false};
        const float* rp = p.radiance_ratio;
        asm volatile("" : "+s"(rp));
        return RadRatio{*rp, true};
    };
    float ratio_acc = 0.f;   // this lane's share of dL/d(radiance_ratio) = sum dL/d(incident) * isfinite(raw * ratio) * raw
    const bool ratio_zero = RATIO && ratio_now().ratio == 0.f;
    RawSample raw;
    if (g < P) raw = load_raw(p, sid(g), 0, lane, min(64, Ns));
    // The per-(Gaussian, corner) constants -- unit view vector, corner frame, and the upstream gradients folded into the
    // coefficients of the four light sums -- are ~130 scattered loads and ~400 instructions per corner.  With lane = (sample
    // group, corner) all 64 lanes would repeat them for ONE Gaussian; instead the wave prepares its next KB_G Gaussians at once,
    // lane = (Gaussian of the batch, corner), parks the KREC-float records in LDS and picks them up Gaussian by Gaussian.
    for (int gb = g; gb < P; gb += KB_G * gstep) {
    wave_lds_sync();   // the previous batch's records have been read
    if (lane < 4 * KB_G) {
        const int gj = gb + (lane >> 2) * gstep;
        if (gj < P) {
        const size_t gg = sid(gj);
        float V[3] = {p.viewdirs[gg * 3], p.viewdirs[gg * 3 + 1], p.viewdirs[gg * 3 + 2]};
        {
            const float iv = inv_norm(V[0] * V[0] + V[1] * V[1] + V[2] * V[2]);   // = 1 / max(|.|, 1e-12)
            V[0] *= iv; V[1] *= iv; V[2] *= iv;
        }
        GaussConst c;
        load_corner(p, gg, k, V, c);
        // Upstream gradients of this (Gaussian, corner): dL_dreduced plus the rows of the packed features / vfeatures
        // that alias it (see upstream()).
        float gp[3] = {0, 0, 0}, gd[3] = {0, 0, 0}, gs[3] = {0, 0, 0}, gdi[3] = {0, 0, 0}, gin[3] = {0, 0, 0};
        float gmi[3] = {0, 0, 0}, gml[3] = {0, 0, 0}, gmg[3] = {0, 0, 0};
        float dir_b[3] = {0, 0, 0}, dir_n[3] = {0, 0, 0}, dir_r = 0.f;   // direct terms of the packing
        if (a.g_red) {
            const float* gr = a.g_red + gg * NRED;
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                gp[ch] = gr[ch * 4 + k]; gd[ch] = gr[12 + ch * 4 + k]; gs[ch] = gr[24 + ch * 4 + k];
                gdi[ch] = gr[36 + ch * 4 + k]; gin[ch] = gr[48 + ch * 4 + k];
                gmi[ch] = gr[60 + ch]; gml[ch] = gr[63 + ch]; gmg[ch] = gr[66 + ch];
            }
        }
        if (a.g_vfeat) {
            const float* vf = a.g_vfeat + gg * nvf;
            float t0[3], t1[3], t2[3], nv[3];
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                t0[ch] = vf[ch * 4 + k]; t1[ch] = vf[40 + ch * 4 + k]; t2[ch] = tr ? 0.f : vf[52 + ch * 4 + k];
                dir_b[ch] = vf[12 + ch * 4 + k]; nv[ch] = vf[24 + ch * 4 + k];
            }
            dir_r = vf[36 + k];
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                gp[ch] += t0[ch];
                if (tr) gd[ch] += t1[ch]; else { gdi[ch] += t1[ch]; gin[ch] += t2[ch]; }
            }
            // vfeatures[24:36] = (normals @ view[:3,:3])^T  =>  dn[j] += sum_ch g[ch] * view[j][ch]
#pragma unroll
            for (int j = 0; j < 3; j++)
                dir_n[j] = nv[0] * p.viewmatrix[j * 4 + 0] + nv[1] * p.viewmatrix[j * 4 + 1] + nv[2] * p.viewmatrix[j * 4 + 2];
        }
        if (a.g_feat) {
            const float* f = a.g_feat + gg * nf;
            if (tr) {
#pragma unroll
                for (int ch = 0; ch < 3; ch++) gml[ch] += f[1 + ch];
            } else {
#pragma unroll
                for (int ch = 0; ch < 3; ch++) { gmi[ch] += f[ch]; gml[ch] += f[3 + ch]; }
            }
        }
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            gp[ch] *= inv_ns; gd[ch] *= inv_ns; gs[ch] *= inv_ns; gdi[ch] *= inv_ns; gin[ch] *= inv_ns;
            gmi[ch] *= inv_ns; gml[ch] *= inv_ns; gmg[ch] *= inv_ns;
        }
        float* o = sK + lane * KREC;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const float qd = gdi[ch] + gp[ch], ql = gin[ch] + gp[ch];            // d/df_d = qd * A_d + ql * A_l
            o[ch] = V[ch]; o[3 + ch] = c.nraw[ch]; o[6 + ch] = c.Nh[ch]; o[12 + ch] = c.fd[ch];
            o[20 + ch] = gd[ch] + c.fd[ch] * qd; o[23 + ch] = gd[ch] + c.fd[ch] * ql;   // kAd, kAl
            o[26 + ch] = gs[ch] + qd; o[29 + ch] = gs[ch] + ql;                           // kBd, kBl
            o[32 + ch] = qd; o[35 + ch] = ql;
            o[38 + ch] = gmi[ch] + gmg[ch];   // constant part of the env gradient
            o[41 + ch] = gmi[ch] + gml[ch];   // constant part of dL/dradiance
            o[44 + ch] = dir_b[ch]; o[47 + ch] = dir_n[ch];
        }
        o[9] = c.a2; o[10] = c.kk; o[11] = c.nom1; o[15] = c.r; o[16] = c.sgn; o[17] = c.inv_len; o[18] = c.NoV_raw; o[19] = dir_r;
        }
    }
    wave_lds_sync();
    for (int jb = 0; jb < KB_G; jb++) {
        g = gb + jb * gstep;
        if (g >= P) break;
        const size_t gg = sid(g);
        zero_rows(zquota);
        float V[3], kAd[3], kAl[3], kBd[3], kBl[3], qd[3], ql[3], gmig[3], dir_b[3], dir_n[3], dir_r, grad_const;
        GaussConst c;
        {
            const float4* q = reinterpret_cast<const float4*>(sK + (jb * 4 + k) * KREC);
            const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4], q5 = q[5], q6 = q[6], q7 = q[7], q8 = q[8], q9 = q[9], q10 = q[10],
                         q11 = q[11], q12 = q[12];
            V[0] = q0.x; V[1] = q0.y; V[2] = q0.z; c.nraw[0] = q0.w; c.nraw[1] = q1.x; c.nraw[2] = q1.y;
            c.Nh[0] = q1.z; c.Nh[1] = q1.w; c.Nh[2] = q2.x; c.a2 = q2.y; c.kk = q2.z; c.nom1 = q2.w;
            c.fd[0] = q3.x; c.fd[1] = q3.y; c.fd[2] = q3.z; c.r = q3.w; c.sgn = q4.x; c.inv_len = q4.y; c.NoV_raw = q4.z; dir_r = q4.w;
            kAd[0] = q5.x; kAd[1] = q5.y; kAd[2] = q5.z; kAl[0] = q5.w; kAl[1] = q6.x; kAl[2] = q6.y;
            kBd[0] = q6.z; kBd[1] = q6.w; kBd[2] = q7.x; kBl[0] = q7.y; kBl[1] = q7.z; kBl[2] = q7.w;
            qd[0] = q8.x; qd[1] = q8.y; qd[2] = q8.z; ql[0] = q8.w; ql[1] = q9.x; ql[2] = q9.y;
            gmig[0] = q9.z; gmig[1] = q9.w; gmig[2] = q10.x;
            grad_const = k == 0 ? q10.y : (k == 1 ? q10.z : q10.w);   // lane k owns channel k (< 3) of dL/dradiance
            dir_b[0] = q11.x; dir_b[1] = q11.y; dir_b[2] = q11.z; dir_n[0] = q11.w; dir_n[1] = q12.x; dir_n[2] = q12.y;
        }
        DEV_TRACE_MARK(0);   // inputs -> per-(Gaussian, corner) constants
        dev_n++;
        float d_fd[3] = {0, 0, 0}, d_n[3] = {0, 0, 0}, d_Nh[3] = {0, 0, 0};
        float s_a2 = 0.f, s_nom1 = 0.f, s_kk2 = 0.f, s_nov = 0.f;   // per-Gaussian sums whose chain rule is applied once, below
        const float NoV = fminf(1.f, fmaxf(1e-6f, c.NoV_raw));
        const bool nov_in = c.NoV_raw >= 1e-6f && c.NoV_raw <= 1.f;
        for (int s0 = 0; s0 < Ns; s0 += 64) {
            const int cnt = min(64, Ns - s0);
            wave_lds_sync();   // previous chunk consumed
            if (lane < cnt) stage_raw_bwd(p, raw, lane, V, sS, ratio_now());
            {   // prefetch the next chunk (of this Gaussian or of the wave's next one)
                const bool more = s0 + 64 < Ns;
                const int gn = more ? g : g + gstep;
                const int sn = more ? s0 + 64 : 0;
                if (gn < P) raw = load_raw(p, more ? gg : sid(gn), sn, lane, min(64, Ns - sn));
            }
            wave_lds_sync();
            DEV_TRACE_MARK(1);   // staging of a chunk (+ issue of the prefetches)
#pragma unroll 1
            for (int it = 0; it < 4; it++) {
                const int s = sg + 16 * it;
                if (16 * it >= cnt) break;   // wave-uniform
                const bool act = s < cnt;
                float r[BREC];
                {
                    const float4* rq = reinterpret_cast<const float4*>(sS + (act ? s : 0) * BREC);
#pragma unroll
                    for (int i = 0; i < BREC / 4; i++) { const float4 q = rq[i]; r[4 * i] = q.x; r[4 * i + 1] = q.y; r[4 * i + 2] = q.z; r[4 * i + 3] = q.w; }
                }
                const float ndr = c.nraw[0] * r[0] + c.nraw[1] * r[1] + c.nraw[2] * r[2];
                const float ndi = fmaxf(ndr, 0.f);
                const float Lv[3] = {r[0] * r[3], r[1] * r[3], r[2] * r[3]};
                const float NoLr = c.Nh[0] * Lv[0] + c.Nh[1] * Lv[1] + c.Nh[2] * Lv[2];
                const float NoHr = c.Nh[0] * r[4] + c.Nh[1] * r[5] + c.Nh[2] * r[6];
                const float NoL = fminf(1.f, fmaxf(1e-6f, NoLr)), NoH = fminf(1.f, fmaxf(1e-6f, NoHr));
                const float nom0 = ggx_nom0(c.Nh, r + 4, c.a2);   // (the VALUE; its derivative below is the literal form's)
                const float nom2 = NoL * (1.f - c.kk) + c.kk;
                const float nomr = 4.f * kPi * nom0 * nom0 * c.nom1 * nom2;
                const float nom = fminf(4.f * kPi, fmaxf(1e-6f, nomr));
                const float inv_nom = __builtin_amdgcn_rcpf(nom);
                const float fs = r[7] * c.a2 * inv_nom;
                const float area = act ? r[11] : 0.f, ge = area * ndi;
                float d_fs = 0.f, d_ndi = 0.f;
                float xg[3], xl[3];   // d/d(global light), d/d(local light) of this (corner, sample)
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    // The five outputs are linear in  A_d = Lg*ge, A_l = Ll*ge, B_d = fs*A_d, B_l = fs*A_l  (see the
                    // forward): dL/dA_d = kAd, dL/dA_l = kAl, dL/dB_d = kBd, dL/dB_l = kBl are per-(corner, channel)
                    // constants computed once per Gaussian.
                    const float Lg = r[8 + ch], Ll = r[12 + ch];
                    const float td = Lg * ge, tl = Ll * ge;
                    const float cg = kAd[ch] + fs * kBd[ch], cl = kAl[ch] + fs * kBl[ch];
                    xg[ch] = cg * ge; xl[ch] = cl * ge;
                    d_fd[ch] += qd[ch] * td + ql[ch] * tl;
                    d_fs += kBd[ch] * td + kBl[ch] * tl;
                    d_ndi += (cg * Lg + cl * Ll) * area;
                }
                {
                    const float dm = ndr > 0.f ? d_ndi : 0.f;
#pragma unroll
                    for (int j = 0; j < 3; j++) d_n[j] += dm * r[j];
                }
                // fs = frac0 * a2 / nom
                const float d_nom = (nomr >= 1e-6f && nomr <= 4.f * kPi) ? -d_fs * fs * inv_nom : 0.f;
                const float t4 = 4.f * kPi;
                const float d_nom0 = d_nom * t4 * 2.f * nom0 * c.nom1 * nom2;
                const float d_nom1 = d_nom * t4 * nom0 * nom0 * nom2;
                const float d_nom2 = d_nom * t4 * nom0 * nom0 * c.nom1;
                // a2 enters frac (fs/a2) and nom0; kk enters nom1, nom2
                s_a2 += d_fs * r[7] * inv_nom + d_nom0 * NoH * NoH;   // dL/da2
                s_nom1 += d_nom1;                                      // nom1 = NoV (1 - kk) + kk is per-Gaussian
                s_kk2 += d_nom2 * (1.f - NoL);                         // kk through nom2
                const float d_NoH = (NoHr >= 1e-6f && NoHr <= 1.f) ? d_nom0 * 2.f * NoH * (c.a2 - 1.f) : 0.f;
                const float d_NoL = (NoLr >= 1e-6f && NoLr <= 1.f) ? d_nom2 * (1.f - c.kk) : 0.f;
#pragma unroll
                for (int j = 0; j < 3; j++) d_Nh[j] += d_NoH * r[4 + j] + d_NoL * Lv[j];

                // ---- adjoint of the sample: sum the four corners (one quad), then lane k takes channel / tap k ----
#pragma unroll
                for (int ch = 0; ch < 3; ch++) { xg[ch] = quad_sum(xg[ch]); xl[ch] = quad_sum(xl[ch]); }
                {
                    {   // parked in the sample's (consumed) local-light slots, written out below (lane k = 3: the pad
                        // slot; lanes beyond the chunk: records nobody reads)
                        const float v = k == 0 ? xl[0] : (k == 1 ? xl[1] : xl[2]);
                        float gk = v + grad_const;   // dL/d(incident radiance), channel k of this sample
                        if (RATIO) {
                            // dL/d(ratio) = sum gk * isfinite(raw * ratio) * raw.  The record holds incident = nan_to_num(raw * ratio): with
                            // ratio != 0 the sum is taken over gk * incident and divided by the ratio once, at the end (where the product
                            // was not finite -- flagged in the record -- there is no gradient, as for nan_to_num; torch then adds
                            // 0 * inf = NaN to the scalar's gradient, here the entry adds nothing); with ratio == 0 (uniform, rare) the
                            // raw values are read again where the chunk's rows leave (below).
                            const float inc = k == 0 ? r[12] : (k == 1 ? r[13] : r[14]);
                            const bool fin = ((__builtin_bit_cast(uint32_t, r[19]) >> k) & 1u) == 0u;   // (stage_raw_bwd)
                            gk = fin ? gk : 0.f;
                            ratio_acc += (act && k < 3) ? gk * inc : 0.f;
                        }
                        sS[s * BREC + (k < 3 ? 12 + k : BREC - 1)] = gk;
                    }
                    const uint32_t xy = __builtin_bit_cast(uint32_t, r[16]);
                    const int tx = (int)(xy & 0x3fffu) - 1 + (k & 1), ty = (int)((xy >> 14) & 0x3fffu) - 1 + (k >> 1);
                    const float fx = r[17], fy = r[18];
                    const float w = ((k & 1) ? fx : 1.f - fx) * ((k >> 1) ? fy : 1.f - fy) * r[15];
                    const int idx = (ty * We + tx) * 3;
                    float dt[3];
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) dt[ch] = ((xy >> (28 + ch)) & 1u) ? (xg[ch] + gmig[ch]) * w : 0.f;
                    // ONE test per tap: inside the chunk and the image, and not all three channels zero (they vanish together:
                    // zero weight or an occluded / clamped sample)
                    if (act && tx >= 0 && tx < We && ty >= 0 && ty < He && (dt[0] != 0.f || dt[1] != 0.f || dt[2] != 0.f)) {
#pragma unroll
                        for (int ch = 0; ch < 3; ch++) {
                            if (env_in_lds) atomicAdd(&sEnv[idx + ch], (double)dt[ch]);   // ds_add_f64
                            else atomic_add_f32(&a.d_envtab[idx + ch], dt[ch]);
                        }
                    }
                }
            }
            DEV_TRACE_MARK(2);   // corner x sample loop
            // dL/dradiance of the chunk: the values parked in the records leave as whole rows (3 cnt consecutive floats)
            wave_lds_sync();
            __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): the next Gaussian's inputs (LDS-DMA) and the next chunk's samples, both issued an inner loop ago
            if (RATIO && ratio_zero) {   // (uniform, rare: incident = 0 everywhere, the sum above is 0 -- the scalar's gradient from the raw cache)
                const float* rawp = p.radiance + (gg * Ns + s0) * 3;
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const int i = lane + 64 * j;
                    if (i < 3 * cnt) ratio_acc += sS[(i / 3) * BREC + 12 + (i % 3)] * rawp[i];
                }
            }
            if (a.d_radiance) {
                float* out = a.d_radiance + (gg * Ns + s0) * 3;
                const float rsc = RATIO ? ratio_now().ratio : 1.f;   // (RATIO: the parked values are masked dL/d(incident); d(incident)/d(raw) = ratio)
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const int i = lane + 64 * j;
                    if (i < 3 * cnt) out[i] = RATIO ? sS[(i / 3) * BREC + 12 + (i % 3)] * rsc : sS[(i / 3) * BREC + 12 + (i % 3)];
                }
            }
        }
        // per-Gaussian chain rule of the sums: a2 = r^4, kk = (r^2 + 2r + 1)/8 (through nom1 and nom2), NoV -> Nh
        float d_r;
        {
            const float d_kk = s_nom1 * (1.f - NoV) + s_kk2;
            d_r = s_a2 * 4.f * c.r * c.r * c.r + d_kk * (2.f * c.r + 2.f) / 8.f;
            const float d_NoV = nov_in ? s_nom1 * (1.f - c.kk) : 0.f;
#pragma unroll
            for (int j = 0; j < 3; j++) d_Nh[j] += d_NoV * V[j];
        }
        // Nh = sgn * n / |n|  =>  dn += sgn/|n| * (dNh - Nn (Nn . dNh)),  Nn = n/|n|
        {
            const float Nn[3] = {c.nraw[0] * c.inv_len, c.nraw[1] * c.inv_len, c.nraw[2] * c.inv_len};
            const float dot = Nn[0] * d_Nh[0] + Nn[1] * d_Nh[1] + Nn[2] * d_Nh[2];
#pragma unroll
            for (int j = 0; j < 3; j++) d_n[j] += c.sgn * c.inv_len * (d_Nh[j] - Nn[j] * dot);
        }
#pragma unroll
        for (int j = 0; j < 3; j++) { d_n[j] = stride4_sum(d_n[j]); d_fd[j] = stride4_sum(d_fd[j]); }
        d_r = stride4_sum(d_r);
        if (sg == 0) {
#pragma unroll
            for (int j = 0; j < 3; j++) {
                a.d_normals[gg * 12 + k * 3 + j] = d_n[j] + dir_n[j];
                a.d_base[gg * 12 + j * 4 + k] = d_fd[j] * kInvPi + dir_b[j];
            }
            a.d_rough[gg * 4 + k] = d_r + dir_r;
        }
        DEV_TRACE_MARK(3);   // row stores + per-Gaussian chain rule and stores
    }
    }
    zero_rows(0x7fffffff);   // (what is left of this wave's share: waves with few or no surfels of their own)
    DEV_TRACE_END(0, dev_n, (unsigned)Ns, 0u);
    __syncthreads();
    if (RATIO && a.ratio_part) {   // the workgroup's partial sum, waves in order (the epilogue adds the workgroups in a fixed order: reproducible)
        const float rz = ratio_now().ratio;
        const float ws = wave_sum(ratio_acc) * (rz != 0.f ? 1.f / rz : 1.f);   // (sum over gk * incident -> sum over gk * raw)
        if (lane == 0) smem[wave] = ws;   // (the sample records are free: every wave passed the barrier above)
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.f;
            for (int w = 0; w < BWAVES; w++) t += smem[w];
            a.ratio_part[blockIdx.x] = t;
        }
        __syncthreads();
    }
    if (env_in_lds) {
        for (int i = threadIdx.x; i < ntex; i += BWAVES * 64) {
            const float v = (float)sEnv[i];
            if (v != 0.f) atomic_add_f32(&a.d_envtab[i], v);
        }
    }
}

// dL/d env_raw = dL/d f(env) * f'(env);  softplus' = sigmoid
__global__ void __launch_bounds__(BLOCK) env_grad_kernel(const float* __restrict__ env, const float* __restrict__ dtab,
                                                         float* __restrict__ denv, int n, int softplus,
                                                         const float* __restrict__ ratio_part, int nparts, float* __restrict__ d_ratio) {
    if (d_ratio && blockIdx.x == 0) {   // dL/d(radiance_ratio): the workgroups' partial sums (<= 256), a fixed tree
        __shared__ float part[BLOCK];
        part[threadIdx.x] = (int)threadIdx.x < nparts ? ratio_part[threadIdx.x] : 0.f;
        __syncthreads();
        for (int h = BLOCK / 2; h >= 1; h >>= 1) {
            if ((int)threadIdx.x < h) part[threadIdx.x] += part[threadIdx.x + h];
            __syncthreads();
        }
        if (threadIdx.x == 0) *d_ratio = part[0];
    }
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const float x = env[i];
    denv[i] = softplus ? dtab[i] * (x > 20.f ? 1.f : 1.f / (1.f + expf(-x))) : dtab[i];
}

}  // namespace

}  // namespace svgir

using namespace svgir;

#ifdef SVGIR_DEV
extern "C" int svgir_dev_trace_read_shade(unsigned long long* out, int cap_records, int slot) {   // slot 0: backward, 1: forward
    unsigned int n[2] = {0, 0};
    if (hipMemcpyFromSymbol(n, HIP_SYMBOL(svgir::g_dev_trace_n), sizeof(n)) != hipSuccess) return -1;
    const int cnt = (int)std::min<unsigned>(n[slot & 1], (unsigned)std::min(cap_records, svgir::DEV_TRACE_CAP));
    if (cnt > 0 && hipMemcpyFromSymbol(out, HIP_SYMBOL(svgir::g_dev_trace), (size_t)cnt * svgir::DEV_TRACE_WORDS * 8,
                                       (size_t)(slot & 1) * svgir::DEV_TRACE_CAP * svgir::DEV_TRACE_WORDS * 8) != hipSuccess) return -1;
    n[slot & 1] = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(svgir::g_dev_trace_n), n, sizeof(n));
    return cnt;
}
#endif

extern "C" {

int svgir_shade_forward(const svgir_shade_params* p, float* reduced, float* features, float* vfeatures, void* stream) {
    return svgir::shade_forward_impl(p, reduced, features, vfeatures, false, stream);
}

}  // extern "C"

svgir::ShadeTables svgir::shade_tables(const svgir_shade_params* p, float* zero, int nzero) { return shade_tables_of(p, zero, nzero); }

int svgir::shade_forward_impl(const svgir_shade_params* p, float* reduced, float* features, float* vfeatures, bool tables_ready, void* stream) {
    if (!p || p->P < 0 || p->Ns <= 0 || p->env_h <= 0 || p->env_w <= 0) return SVGIR_ERR_INVALID;
    if (p->P == 0) return 0;
    if (!p->base_color || !p->roughness || !p->normals || !p->viewdirs || !p->radiance || !p->visibility ||
        !p->env || !p->env_work || (vfeatures && !p->viewmatrix))
        return SVGIR_ERR_INVALID;
    if (!p->incident_dirs && !(p->lattice_normals && p->lattice_work)) return SVGIR_ERR_INVALID;
    if (((uintptr_t)p->env_work & 15) || (p->lattice_work && ((uintptr_t)p->lattice_work & 15))) return SVGIR_ERR_INVALID;   // read as float4
    if ((p->subset != nullptr) != (p->subset_count != nullptr)) return SVGIR_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const int ntex = p->env_h * p->env_w * 3;
    StageMarks tm = stage_begin(s);
    if (!tables_ready) launch_shade_prologue(p, nullptr, 0, s);
    stage_mark(tm, "shade_env_table");
    ShadeArgs a;
    a.p = *p; a.reduced = reduced; a.features = features; a.vfeatures = vfeatures;
    // (subset launches: the rows of the surfels outside the subset are zero-filled by the shading kernel's own waves)
#ifndef SHADE_FWD_QUAD
#define SHADE_FWD_QUAD 1
#endif
#ifndef SHADE_FWD_QUAD_MAX_NS
#define SHADE_FWD_QUAD_MAX_NS 128
#endif
    // The quad layout removes the per-surfel fixed costs (250 -> 202 us at P = 200 k, Ns = 64); with hundreds of samples per surfel
    // those are amortised anyway and the one-wave-per-surfel kernel streams the samples better (lane = sample: 768 contiguous bytes
    // per load; 771 us against 1 001 us at Ns = 384)
    if (SHADE_FWD_QUAD && p->Ns <= SHADE_FWD_QUAD_MAX_NS) {
        const size_t lds = (size_t)4 * (FQ_SURF * (FQ_ROW + FQ_IN) + FQ_SURF) * 4;
        if (p->radiance_ratio) hipLaunchKernelGGL(shade_fwd_quad_kernel<true>, dim3((p->P + 4 * FQ_SURF - 1) / (4 * FQ_SURF)), dim3(BLOCK), lds, s, a);
        else hipLaunchKernelGGL(shade_fwd_quad_kernel<false>, dim3((p->P + 4 * FQ_SURF - 1) / (4 * FQ_SURF)), dim3(BLOCK), lds, s, a);
    } else {
        const size_t lds = (size_t)4 * (64 * SREC + 80 + 32) * 4;
        if (p->radiance_ratio) hipLaunchKernelGGL(shade_fwd_kernel<true>, dim3((p->P + 3) / 4), dim3(BLOCK), lds, s, a);
        else hipLaunchKernelGGL(shade_fwd_kernel<false>, dim3((p->P + 3) / 4), dim3(BLOCK), lds, s, a);
    }
    stage_mark(tm, "shade_fwd");
    return hipGetLastError() == hipSuccess ? 0 : SVGIR_ERR_HIP;
}

extern "C" {

int svgir_shade_backward(const svgir_shade_params* p, const float* dL_dreduced, const float* dL_dfeatures,
                         const float* dL_dvfeatures, float* dL_dbase_color,
                         float* dL_droughness, float* dL_dnormals, float* dL_dradiance, float* dL_denv,
                         float* env_grad_work, float* dL_dradiance_ratio, void* stream) {
    return svgir::shade_backward_impl(p, dL_dreduced, dL_dfeatures, dL_dvfeatures, dL_dbase_color, dL_droughness, dL_dnormals, dL_dradiance,
                                      dL_denv, env_grad_work, dL_dradiance_ratio, false, false, stream);
}

}  // extern "C"

int svgir::shade_backward_impl(const svgir_shade_params* p, const float* dL_dreduced, const float* dL_dfeatures,
                               const float* dL_dvfeatures, float* dL_dbase_color,
                               float* dL_droughness, float* dL_dnormals, float* dL_dradiance, float* dL_denv,
                               float* env_grad_work, float* dL_dradiance_ratio, bool rows_precleared, bool tables_ready, void* stream) {
    if (!p || p->P < 0 || p->Ns <= 0 || p->env_h <= 0 || p->env_w <= 0) return SVGIR_ERR_INVALID;
    if (p->P == 0) return 0;
    if ((!dL_dreduced && !dL_dfeatures && !dL_dvfeatures) || (dL_dvfeatures && !p->viewmatrix) || !dL_dbase_color || !dL_droughness || !dL_dnormals || !dL_denv ||
        !env_grad_work || !p->env_work)
        return SVGIR_ERR_INVALID;
    // (with radiance_ratio the cache itself is detached in the reference: its gradient is optional; the scalar's gradient needs the scalar)
    if ((!dL_dradiance && !p->radiance_ratio) || (dL_dradiance_ratio && !p->radiance_ratio)) return SVGIR_ERR_INVALID;
    if (!p->incident_dirs && !(p->lattice_normals && p->lattice_work)) return SVGIR_ERR_INVALID;
    if (((uintptr_t)p->env_work & 15) || (p->lattice_work && ((uintptr_t)p->lattice_work & 15))) return SVGIR_ERR_INVALID;   // read as float4
    if ((p->subset != nullptr) != (p->subset_count != nullptr)) return SVGIR_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const int ntex = p->env_h * p->env_w * 3;
    StageMarks tm = stage_begin(s);
    if (!tables_ready) launch_shade_prologue(p, env_grad_work, ntex, s);

    ShadeBwdArgs a;
    a.p = *p; a.g_red = dL_dreduced; a.g_feat = dL_dfeatures; a.g_vfeat = dL_dvfeatures; a.d_base = dL_dbase_color; a.d_rough = dL_droughness; a.d_normals = dL_dnormals;
    a.d_radiance = dL_dradiance; a.d_envtab = env_grad_work;
    a.ratio_part = dL_dradiance_ratio ? env_grad_work + ntex : nullptr;   // (SVGIR_SHADE_RATIO_WORK floats behind the env-gradient table)
    a.zero_rest = (p->subset && !rows_precleared) ? 1 : 0;   // (every output is written completely: rows outside the subset are zero)
    const bool ratio = p->radiance_ratio != nullptr;
    const size_t per_wave = (size_t)(64 * BREC + 4 * KB_G * KREC) * 4;
    size_t lds = BWAVES * per_wave;
    int env_in_lds = 0;
    if (p->env_w > 16000 || p->env_h > 16000) return SVGIR_ERR_INVALID;   // (14-bit footprint origins in the sample records)
    static_assert((BWAVES * (64 * BREC + 4 * KB_G * KREC)) % 2 == 0, "the fp64 image behind the sample records is 8-byte aligned");
    if (lds + (size_t)ntex * 8 <= 160 * 1024) { env_in_lds = 1; lds += (size_t)ntex * 8; }   // one workgroup per CU
    {
        static bool attr_set[64] = {};   // per device (> 64 KB of dynamic LDS needs the opt-in; idempotent, so races are harmless)
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return SVGIR_ERR_HIP;
        if (dev < 0 || dev >= 64 || !attr_set[dev]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(shade_bwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(shade_bwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return SVGIR_ERR_HIP;
            if (dev >= 0 && dev < 64) attr_set[dev] = true;
        }
    }
    const int blocks = std::min((p->P + BWAVES - 1) / BWAVES, 256 * std::max(1, SHADE_BWPE * 4 / BWAVES));
    static_assert(256 * (SHADE_BWPE * 4 / BWAVES > 1 ? SHADE_BWPE * 4 / BWAVES : 1) <= SVGIR_SHADE_RATIO_WORK, "one partial sum per workgroup");
    stage_mark(tm, "shade_bwd_prologue");
    if (ratio) hipLaunchKernelGGL(shade_bwd_kernel<true>, dim3(blocks), dim3(BWAVES * 64), lds, s, a, env_in_lds);
    else hipLaunchKernelGGL(shade_bwd_kernel<false>, dim3(blocks), dim3(BWAVES * 64), lds, s, a, env_in_lds);
    stage_mark(tm, "shade_bwd");
    hipLaunchKernelGGL(env_grad_kernel, dim3((ntex + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, p->env, env_grad_work,
                       dL_denv, ntex, p->env_softplus, a.ratio_part, blocks, dL_dradiance_ratio);
    stage_mark(tm, "shade_env_grad");
    return hipGetLastError() == hipSuccess ? 0 : SVGIR_ERR_HIP;
}

extern "C" {

int svgir_incident_dirs(int32_t P, int32_t Ns, const float* normals, const float* offsets, float* lattice_work,
                        float* dirs, float* areas, void* stream) {
    if (P < 0 || Ns <= 0 || (P > 0 && (!normals || !lattice_work))) return SVGIR_ERR_INVALID;
    if (P == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(lattice_table_kernel, dim3((Ns + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, Ns, (float4*)lattice_work);
    const size_t n = (size_t)P * Ns;
    hipLaunchKernelGGL(incident_dirs_kernel, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, P, Ns, normals,
                       offsets, (const float4*)lattice_work, dirs, areas);
    return hipGetLastError() == hipSuccess ? 0 : SVGIR_ERR_HIP;
}

int svgir_resample_bilinear(const float* src, int32_t H, int32_t W, int32_t C, float* dst, int32_t out_h, int32_t out_w,
                            void* stream) {
    if (!src || !dst || H <= 0 || W <= 0 || C <= 0 || out_h <= 0 || out_w <= 0) return SVGIR_ERR_INVALID;
    const int n = out_h * out_w * C;
    hipLaunchKernelGGL(resample_kernel, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, (hipStream_t)stream, src, H, W, C, dst,
                       out_h, out_w);
    return hipGetLastError() == hipSuccess ? 0 : SVGIR_ERR_HIP;
}

}  // extern "C"
