// svg-ir_amd/csrc/preprocess.hip -- per-Gaussian forward stage.
//
// Replaces preprocessCUDA (svgss forward.cu:229-396, rgss forward.cu:176-318) with computeColorFromSH
// (:20-71), computeCov2D (:74-139), quaternion2rotmat (:165-180), computeCov3D (:186-226) and the helpers
// in_frustum / front_facing / local_homo / getRect / ndc2Pix (auxiliary.h), and checkFrustum
// (rasterizer_impl.cu:54-66).
//
// One lane per Gaussian.  Everything the composite kernels gather later is packed into ONE 96-byte record per
// Gaussian (six aligned float4 stores) instead of the reference's eleven separate arrays, so a (tile, splat)
// instance costs one contiguous gather.  The depth-sort key and the identity permutation for the depth sort
// are written here too.  Bandwidth-trivial (<= 236 B in, ~140 B out per Gaussian).
//
// Floating-point contraction is OFF in this file: culling decisions, radii and tile rectangles are integer
// outputs (num_rendered, radii, point_list) and are kept bit-identical to the un-fused arithmetic of the oracle.
#include "common.hpp"
#include "shade_tables.hpp"

#pragma clang fp contract(off)

namespace svgir {

namespace {

struct Mat3 {  // column-major, m[col][row]
    float m[3][3];
};
__device__ __forceinline__ Mat3 mmul(const Mat3& A, const Mat3& B) {
    Mat3 r;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int row = 0; row < 3; row++)
            r.m[c][row] = A.m[0][row] * B.m[c][0] + A.m[1][row] * B.m[c][1] + A.m[2][row] * B.m[c][2];
    return r;
}
__device__ __forceinline__ Mat3 mtr(const Mat3& A) {
    Mat3 r;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int row = 0; row < 3; row++) r.m[c][row] = A.m[row][c];
    return r;
}
__device__ __forceinline__ float ndc2pix(float v, int S) { return (float)((((double)v + 1.0) * S - 1.0) * 0.5); }

__device__ __forceinline__ float normalize3(float* v) {
    const float mod = fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), (float)0.00000001);
    v[0] /= mod; v[1] /= mod; v[2] /= mod;
    return mod;
}

__device__ __forceinline__ void tile_rect(float px, float py, int rad, int gx, int gy, int* rmin, int* rmax) {
    const float r = (float)rad;
    rmin[0] = min(gx, max(0, (int)((px - r) / TILE)));
    rmin[1] = min(gy, max(0, (int)((py - r) / TILE)));
    rmax[0] = min(gx, max(0, (int)((px + r + TILE - 1) / TILE)));
    rmax[1] = min(gy, max(0, (int)((py + r + TILE - 1) / TILE)));
}

__constant__ float kC2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                             0.5462742152960396f};
__constant__ float kC3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                             -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

// The SH coefficient row of Gaussian idx (zero above the active degree).
__device__ __forceinline__ void load_sh(const PreArgs& a, int idx, float* sh) {
    const float* shp = a.shs + (size_t)idx * a.M * 3;
    // M = 16: the 48-float coefficient row is contiguous and 16-byte aligned -> 12 float4 loads instead of 48 dword
    // loads (only the loads change, the arithmetic below is untouched)
    if (a.M == 16 && (((size_t)a.shs) & 15) == 0) {
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const float4 v = reinterpret_cast<const float4*>(shp)[i];
            sh[4 * i] = v.x; sh[4 * i + 1] = v.y; sh[4 * i + 2] = v.z; sh[4 * i + 3] = v.w;
        }
    } else {
        const int n = 3 * (a.D + 1) * (a.D + 1);
#pragma unroll
        for (int i = 0; i < 48; i++) sh[i] = i < n ? shp[i] : 0.f;
    }
}

// SH -> RGB, degrees 0..3; returns clamp mask in the low 3 bits.
__device__ __forceinline__ uint32_t sh_to_rgb(const PreArgs& a, const float* sh, const float* campos, const float* pos, float* rgb) {
    const float kC0 = 0.28209479177387814f, kC1 = 0.4886025119029199f;
    float dir[3] = {pos[0] - campos[0], pos[1] - campos[1], pos[2] - campos[2]};
    const float len = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    const float x = dir[0] / len, y = dir[1] / len, z = dir[2] / len;
    float res[3];
#pragma unroll
    for (int c = 0; c < 3; c++) res[c] = kC0 * sh[c];
    if (a.D > 0) {
#pragma unroll
        for (int c = 0; c < 3; c++) res[c] = res[c] - kC1 * y * sh[3 + c] + kC1 * z * sh[6 + c] - kC1 * x * sh[9 + c];
        if (a.D > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
#pragma unroll
            for (int c = 0; c < 3; c++)
                res[c] = res[c] + kC2[0] * xy * sh[12 + c] + kC2[1] * yz * sh[15 + c] +
                         kC2[2] * (2.0f * zz - xx - yy) * sh[18 + c] + kC2[3] * xz * sh[21 + c] +
                         kC2[4] * (xx - yy) * sh[24 + c];
            if (a.D > 2) {
#pragma unroll
                for (int c = 0; c < 3; c++)
                    res[c] = res[c] + kC3[0] * y * (3.0f * xx - yy) * sh[27 + c] + kC3[1] * xy * z * sh[30 + c] +
                             kC3[2] * y * (4.0f * zz - xx - yy) * sh[33 + c] +
                             kC3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[36 + c] +
                             kC3[4] * x * (4.0f * zz - xx - yy) * sh[39 + c] + kC3[5] * z * (xx - yy) * sh[42 + c] +
                             kC3[6] * x * (xx - 3.0f * yy) * sh[45 + c];
            }
        }
    }
    uint32_t mask = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        res[c] += 0.5f;
        if (res[c] < 0) mask |= 1u << c;
        rgb[c] = fmaxf(res[c], 0.0f);
    }
    return mask;
}

template <bool SVGSS>
__global__ void __launch_bounds__(BLOCK) preprocess_kernel(const PreArgs a) {
    const int idx = blockIdx.x * BLOCK + threadIdx.x;
    // Memory order (one round of waves: the kernel costs its chain of dependent memory latencies): the uniform inputs are read
    // before the first store (behind one, the compiler can no longer prove them unclobbered and they stop being scalar loads); a
    // Gaussian's own inputs are requested together right behind the frustum test, not where the arithmetic first needs them.
    float V[16], PM[16], campos[3] = {0.f, 0.f, 0.f}, pbb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; i++) { V[i] = a.view[i]; PM[i] = a.proj[i]; }
    if (!a.colors_precomp) { campos[0] = a.campos[0]; campos[1] = a.campos[1]; campos[2] = a.campos[2]; }
    if (SVGSS) { pbb[0] = a.patchbbox[0]; pbb[1] = a.patchbbox[1]; pbb[2] = a.patchbbox[2]; pbb[3] = a.patchbbox[3]; }
    for (int j = idx; j < a.n_zero_words; j += gridDim.x * BLOCK) a.zero_words[j] = 0u;
    if ((int)blockIdx.x >= a.pblocks) {   // fused shading: the workgroups behind the Gaussians' build the shading kernels' tables
        shade_table_entry(a.tabs, ((int)blockIdx.x - a.pblocks) * BLOCK + (int)threadIdx.x);
        return;
    }
    if (idx >= a.P) return;
    const bool surface = cfg_flag(a.cfg, 0), pix_depth = cfg_flag(a.cfg, 2);
    if (a.out_weights) a.out_weights[idx] = 0.f;
    if (a.needed) a.needed[idx] = 0;
    if (idx == 0 && a.span) *a.span = 0u;   // (accumulated with atomicMax by the offsets scan)
    // defaults for a culled Gaussian
    a.radii[idx] = 0;
    a.tiles[2 * idx] = 0;
    // (behind every visible key; under a speculated common top byte the depth sort skips that byte, so the culled keys carry it too.
    // Where a culled Gaussian lands in the depth order is immaterial: it emits nothing)
    a.key[idx] = a.spec_top >= 0 ? (((uint32_t)a.spec_top << 24) | 0x00FFFFFFu) : 0xFFFFFFFFu;
    a.idx[idx] = (uint32_t)idx;
    if ((threadIdx.x & 63) == 0) a.key_top[idx >> 6] = 0xff00u;   // this wave's summary: none visible so far

    const float po[3] = {a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]};
    float q[4] = {1.f, 0.f, 0.f, 0.f};
    if (a.rotations) {
        const float4 qq = reinterpret_cast<const float4*>(a.rotations)[idx];
        q[0] = qq.x; q[1] = qq.y; q[2] = qq.z; q[3] = qq.w;
    }
    const float hx = PM[0] * po[0] + PM[4] * po[1] + PM[8] * po[2] + PM[12];
    const float hy = PM[1] * po[0] + PM[5] * po[1] + PM[9] * po[2] + PM[13];
    const float hw = PM[3] * po[0] + PM[7] * po[1] + PM[11] * po[2] + PM[15];
    const float pw = 1.0f / (hw + 0.0000001f);
    const float ppx = hx * pw, ppy = hy * pw;
    const float pv[3] = {V[0] * po[0] + V[4] * po[1] + V[8] * po[2] + V[12],
                         V[1] * po[0] + V[5] * po[1] + V[9] * po[2] + V[13],
                         V[2] * po[0] + V[6] * po[1] + V[10] * po[2] + V[14]};
    const float pix[2] = {ndc2pix(ppx, a.W), ndc2pix(ppy, a.H)};
    if (SVGSS) {
        const float x0 = pbb[1], y0 = pbb[0], x1 = pbb[3], y1 = pbb[2];
        const float w = x1 - x0, h = y1 - y0, e = (float)0.2;
        if (pv[2] < 0 || pix[0] < x0 - w * e || pix[0] >= x1 + w * e || pix[1] < y0 - h * e || pix[1] >= y1 + h * e) {
            if (a.prefilter_violation) *a.prefilter_violation = 1u;   // the reference traps here (auxiliary.h:163-167)
            return;
        }
    } else {
        if (pv[2] <= 0.2f) {
            if (a.prefilter_violation) *a.prefilter_violation = 1u;
            return;
        }
    }

    const float r = q[0], x = q[1], y = q[2], z = q[3];
    Mat3 Rm;
    Rm.m[0][0] = 1.f - 2.f * (y * y + z * z); Rm.m[0][1] = 2.f * (x * y - r * z); Rm.m[0][2] = 2.f * (x * z + r * y);
    Rm.m[1][0] = 2.f * (x * y + r * z); Rm.m[1][1] = 1.f - 2.f * (x * x + z * z); Rm.m[1][2] = 2.f * (y * z - r * x);
    Rm.m[2][0] = 2.f * (x * z - r * y); Rm.m[2][1] = 2.f * (y * z + r * x); Rm.m[2][2] = 1.f - 2.f * (x * x + y * y);

    float nv[3] = {0, 0, 0}, J[10];
#pragma unroll
    for (int i = 0; i < 10; i++) J[i] = 0;
    if (surface) {
        const float nw[3] = {Rm.m[0][2], Rm.m[1][2], Rm.m[2][2]};
        const float a0w[3] = {Rm.m[0][0], Rm.m[1][0], Rm.m[2][0]};
        const float a1w[3] = {Rm.m[0][1], Rm.m[1][1], Rm.m[2][1]};
        float a0[3], a1[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            nv[i] = V[i] * nw[0] + V[4 + i] * nw[1] + V[8 + i] * nw[2];
            a0[i] = V[i] * a0w[0] + V[4 + i] * a0w[1] + V[8 + i] * a0w[2];
            a1[i] = V[i] * a1w[0] + V[4 + i] * a1w[1] + V[8 + i] * a1w[2];
        }
        const float dot = pv[0] * nv[0] + pv[1] * nv[1] + pv[2] * nv[2];
        if ((double)dot > -0.01) {  // back-facing
            if (a.prefilter_violation) *a.prefilter_violation = 1u;   // auxiliary.h:195-199
            return;
        }
        if (pix_depth) {
            // local homography between the screen and the tangent plane (auxiliary.h:291-388)
            const float qx = pv[0] / pv[2], qy = pv[1] / pv[2];
            const float S_fix = 1000, Svp = (a.focal_x + a.focal_y) / 2;
            float d0[3] = {qx + 1 / S_fix, qy, 1};
            const float m0 = normalize3(d0);
            float d1[3] = {qx, qy + 1 / S_fix, 1};
            const float m1 = normalize3(d1);
            const float c0 = d0[0] * nv[0] + d0[1] * nv[1] + d0[2] * nv[2];
            const float c1 = d1[0] * nv[0] + d1[1] * nv[1] + d1[2] * nv[2];
            if (fabsf(c0 / m0) < 0.01f || fabsf(c1 / m1) < 0.01f) return;  // grazing
            const float t = pv[0] * nv[0] + pv[1] * nv[1] + pv[2] * nv[2];
            const float t0 = t / c0, t1 = t / c1;
            float xu0[3], xu1[3];
#pragma unroll
            for (int i = 0; i < 3; i++) { xu0[i] = d0[i] * t0 - pv[i]; xu1[i] = d1[i] * t1 - pv[i]; }
            const float k = Svp / S_fix;
            J[0] = (xu0[0] * a0[0] + xu0[1] * a0[1] + xu0[2] * a0[2]) / k;
            J[1] = (xu1[0] * a0[0] + xu1[1] * a0[1] + xu1[2] * a0[2]) / k;
            J[2] = (xu0[0] * a1[0] + xu0[1] * a1[1] + xu0[2] * a1[2]) / k;
            J[3] = (xu1[0] * a1[0] + xu1[1] * a1[1] + xu1[2] * a1[2]) / k;
#pragma unroll
            for (int i = 0; i < 3; i++) { J[4 + i] = a0[i]; J[7 + i] = a1[i]; }
        }
    }

    // ---- every remaining input of this Gaussian (half of the surfels of a closed surface never get here: back-facing) ----
    float sc_in[3] = {0.f, 0.f, 0.f}, c3[6], rgb[3] = {0.f, 0.f, 0.f}, sh[48];
    if (a.scales) { sc_in[0] = a.scales[3 * idx]; sc_in[1] = a.scales[3 * idx + 1]; sc_in[2] = a.scales[3 * idx + 2]; }
    if (a.cov3D_precomp) {
#pragma unroll
        for (int i = 0; i < 6; i++) c3[i] = a.cov3D_precomp[6 * idx + i];
    }
    const float opacity = a.opacities[idx];
    if (a.colors_precomp) {
#pragma unroll
        for (int c = 0; c < 3; c++) rgb[c] = a.colors_precomp[3 * idx + c];
    } else {
        load_sh(a, idx, sh);
    }
    if (!a.cov3D_precomp) {
        // quirk Q1: `mod * surface ? 0 : scale.z`
        Mat3 Sm;
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int rr = 0; rr < 3; rr++) Sm.m[c][rr] = 0.f;
        Sm.m[0][0] = a.scale_modifier * sc_in[0];
        Sm.m[1][1] = a.scale_modifier * sc_in[1];
        Sm.m[2][2] = (a.scale_modifier * (surface ? 1.0f : 0.0f)) != 0.0f ? 0.0f : sc_in[2];
        const Mat3 Mm = mmul(Sm, Rm);
        const Mat3 Sg = mmul(mtr(Mm), Mm);
        c3[0] = Sg.m[0][0]; c3[1] = Sg.m[0][1]; c3[2] = Sg.m[0][2];
        c3[3] = Sg.m[1][1]; c3[4] = Sg.m[1][2]; c3[5] = Sg.m[2][2];
#pragma unroll
        for (int i = 0; i < 6; i++) a.cov3D[6 * idx + i] = c3[i];
    }

    // EWA 2D covariance
    float t[3] = {pv[0], pv[1], pv[2]};
    const float limx = 1.3f * a.tanx, limy = 1.3f * a.tany;
    const float txtz = t[0] / t[2], tytz = t[1] / t[2];
    t[0] = fminf(limx, fmaxf(-limx, txtz)) * t[2];
    t[1] = fminf(limy, fmaxf(-limy, tytz)) * t[2];
    Mat3 Jm, Wm, Vk;
    Jm.m[0][0] = a.focal_x / t[2]; Jm.m[0][1] = 0.f; Jm.m[0][2] = -(a.focal_x * t[0]) / (t[2] * t[2]);
    Jm.m[1][0] = 0.f; Jm.m[1][1] = a.focal_y / t[2]; Jm.m[1][2] = -(a.focal_y * t[1]) / (t[2] * t[2]);
    Jm.m[2][0] = 0.f; Jm.m[2][1] = 0.f; Jm.m[2][2] = 0.f;
    Wm.m[0][0] = V[0]; Wm.m[0][1] = V[4]; Wm.m[0][2] = V[8];
    Wm.m[1][0] = V[1]; Wm.m[1][1] = V[5]; Wm.m[1][2] = V[9];
    Wm.m[2][0] = V[2]; Wm.m[2][1] = V[6]; Wm.m[2][2] = V[10];
    Vk.m[0][0] = c3[0]; Vk.m[0][1] = c3[1]; Vk.m[0][2] = c3[2];
    Vk.m[1][0] = c3[1]; Vk.m[1][1] = c3[3]; Vk.m[1][2] = c3[4];
    Vk.m[2][0] = c3[2]; Vk.m[2][1] = c3[4]; Vk.m[2][2] = c3[5];
    const Mat3 Tm = mmul(Wm, Jm);
    const Mat3 cov = mmul(mmul(mtr(Tm), mtr(Vk)), Tm);
    const float ca = cov.m[0][0] + 0.3f, cb = cov.m[0][1], cc = cov.m[1][1] + 0.3f;
    const float det = ca * cc - cb * cb;
    if (det == 0.0f) return;
    const float det_inv = 1.f / det;
    const float conic[3] = {cc * det_inv, -cb * det_inv, ca * det_inv};
    const float mid = 0.5f * (ca + cc);
    const float l1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
    const float l2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
    const float my_radius = ceilf(3.f * sqrtf(fmaxf(l1, l2)));
    int rmin[2], rmax[2];
    tile_rect(pix[0], pix[1], (int)my_radius, a.gx, a.gy, rmin, rmax);
    const int area = (rmax[0] - rmin[0]) * (rmax[1] - rmin[1]);
    if (area == 0) return;

    float emb[5] = {0.f, 0.f, 0.f, 0.f, 0.f};   // no vfeatures: the feature row travels in the record (common.hpp rec_embeds_features)
    if (a.embed_S > 0) {
#pragma unroll
        for (int ch = 0; ch < 5; ch++) emb[ch] = ch < a.embed_S ? a.features[(size_t)idx * a.embed_S + ch] : 0.f;
    }
    uint32_t cmask = 0;
    if (!a.colors_precomp) cmask = sh_to_rgb(a, sh, campos, po, rgb);
    a.clamped[idx] = cmask;
    a.radii[idx] = (int)my_radius;
    reinterpret_cast<uint2*>(a.tiles)[idx] = make_uint2((uint32_t)area, (uint32_t)rmin[0] | ((uint32_t)rmin[1] << 10) | ((uint32_t)(rmax[0] - rmin[0]) << 20));
    const uint32_t depth_key = __float_as_uint(pv[2]);
    a.key[idx] = depth_key;
    {   // AND / OR of the top bytes of this wave's visible keys (the lanes still here), for the host's guess of the next view's common byte
        const unsigned long long act = __ballot(true);
        uint32_t av = 0, ov = 0;
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const unsigned long long m = __ballot((depth_key >> (24 + b)) & 1u);
            av |= (m == act) ? (1u << b) : 0u;
            ov |= (m != 0ull) ? (1u << b) : 0u;
        }
        if ((threadIdx.x & 63) == (unsigned)(__ffsll((long long)act) - 1)) a.key_top[idx >> 6] = (av << 8) | ov;
    }

    float iu = 0.f, iv = 0.f;
    if (SVGSS) {   // (no scales -- cov3D_precomp -- means lambda = 0, i.e. the footprint's 0.1 floor: svgss forward.cu:383, oracle idem)
        const float umx = (float)(0.5 * (double)sc_in[0] + 0.1);
        const float umy = (float)(0.5 * (double)sc_in[1] + 0.1);
        iu = 1.0f / umx; iv = 1.0f / umy;
    }
    float4* rec = reinterpret_cast<float4*>(a.rec + (size_t)idx * REC);
    rec[0] = make_float4(pix[0], pix[1], conic[0], conic[1]);
    // depth-differencing coefficients (common.hpp R_DA / R_DB)
    const float da = J[0] * J[6] + J[2] * J[9], db = J[1] * J[6] + J[3] * J[9];
    rec[1] = make_float4(conic[2], opacity, pv[2], da);
    const bool embed = a.embed_S > 0;
    rec[2] = embed ? make_float4(emb[0], emb[1], emb[2], emb[3]) : make_float4(J[0], J[1], J[2], J[3]);
    rec[3] = make_float4(db, rgb[0], rgb[1], rgb[2]);
    rec[4] = make_float4(nv[0], nv[1], nv[2], embed ? emb[4] : iu);
    rec[5] = make_float4(iv, 0.f, 0.f, 0.f);
}

__global__ void __launch_bounds__(BLOCK) mark_visible_kernel(int P, const float* means3D, const float* V,
                                                             uint8_t* present) {
    const int idx = blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= P) return;
    const float z = V[2] * means3D[3 * idx] + V[6] * means3D[3 * idx + 1] + V[10] * means3D[3 * idx + 2] + V[14];
    present[idx] = z <= 0.2f ? 0 : 1;
}

}  // namespace

void launch_preprocess(const PreArgs& a_, bool svgss, hipStream_t s) {
    PreArgs a = a_;
    a.pblocks = (a.P + BLOCK - 1) / BLOCK;
    const int grid = a.pblocks + (a.tabs.entries() + BLOCK - 1) / BLOCK;   // (+ the shading tables of the fused path)
    if (svgss) hipLaunchKernelGGL(preprocess_kernel<true>, dim3(grid), dim3(BLOCK), 0, s, a);
    else hipLaunchKernelGGL(preprocess_kernel<false>, dim3(grid), dim3(BLOCK), 0, s, a);
}

void launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t s) {
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, P, means3D, view, present);
}

}  // namespace svgir
