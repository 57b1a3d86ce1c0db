// svg-ir_amd/csrc/geom_bwd.hip -- per-Gaussian backward stage (one fused kernel).
//
// Replaces computeCov2DCUDA (svgss backward.cu:163-322, rgss :144-276), the backward preprocessCUDA
// (svgss :437-526, rgss :369-428), the SH backward computeColorFromSH (:20-158 / :20-142) and the covariance
// backward computeCov3D (:326-432 / :280-364), including the s.z != 0 half of quirk Q1, the x10 normal gradient
// injected into the third column of R (Q4), the un-normalised quaternion gradient (Q3) and -- svgss with
// config[3] > 0 only -- the camera gradients dL_dviewmat / dL_dprojmat / dL_dcampos.
//
// The reference launches two kernels with a round trip of dL_dcov3D / dL_dmean3D through HBM; here one lane does
// both halves for its Gaussian.  Contraction is off so results track the oracle's un-fused arithmetic.
//
// Memory order: the kernel is one round of waves (3 per SIMD at P = 200 k), i.e. it costs its chain of dependent memory
// latencies.  As far as the compiler knows every tensor may alias every other, so a load written below a store waits for that store:
// all inputs of a Gaussian are therefore requested at the top (one latency behind the radius test) and every gradient leaves at the
// bottom -- dL_dcov3D and the unpacked composite gradients are used from registers, not re-read (stage 40.5 -> 35.6 us at cfg2).
#include "common.hpp"

#pragma clang fp contract(off)

namespace svgir {

namespace {

struct Mat3 {
    float m[3][3];  // column-major m[col][row]
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

__constant__ float gC2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                             0.5462742152960396f};
__constant__ float gC3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                             -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

// The 48-float coefficient row of a Gaussian (and its gradient row) is 192 contiguous bytes, the lanes of a wave are 192 bytes -- with a
// list, anything -- apart: read or written per lane, each of the 12 float4 accesses of a wave touches one cache line per visible lane
// (~28 of 64 on the BASELINE scenes) and the kernel is bound by exactly that, the number of line accesses of its ~80 per-lane memory
// instructions.  The two rows therefore move ROW-wise through LDS: 12 lanes per row, five rows per instruction (visible Gaussians only),
// i.e. ~6 instructions of ~10 lines each instead of 12 of ~28.
// One wave per workgroup: 12.3 KB of LDS each, i.e. 13 waves per CU -- P = 200 000 is then still ONE round of waves (the kernel costs
// its chain of memory latencies once).
constexpr int GB_BLOCK = 64;
__global__ void __launch_bounds__(GB_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 8))) geom_bwd_kernel(const GeomBwdArgs a) {
    __shared__ float4 sRow[1][64][12];
    __shared__ int sIdx[1][64];
    const int w = blockIdx.x * GB_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = 0;
    // (with a list: only the Gaussians that received a blend weight have non-zero composite gradients; the outputs of the others stay
    // at the zeros svgir_backward cleared them to)
    const bool in_range = w < (a.list ? (int)min(*a.list_count, (uint32_t)a.P) : a.P);
    const int idx = in_range ? (a.list ? (int)a.list[w] : w) : 0;
    const bool visible = in_range && a.radii[idx] > 0;
    // The 48-float coefficient row of a Gaussian (M = 16) is contiguous and 16-byte aligned
    const bool vec = a.shs && a.M == 16 && ((((size_t)a.shs) | ((size_t)a.dL_dsh)) & 15) == 0;
    const bool coop = vec && a.D == 3;   // (wave-uniform) rows through LDS
    const unsigned long long vmask = __ballot(visible);
    const int nvis = __popcll(vmask);
    const int rank = __popcll(vmask & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
    const int rsub = lane / 12, rpart = lane - rsub * 12;   // cooperative row access: row slot (0..4; lanes 60..63 idle), float4 of the row
    if (nvis == 0) return;   // (wave-uniform)
    constexpr int CR = 6;   // row loads held in registers (5 rows each: 30 rows, the usual wave) while the per-lane inputs are requested
    float4 cr[CR];
    if (coop) {
        if (visible) sIdx[wave][rank] = idx;
        wave_lds_sync();
#pragma unroll
        for (int j = 0; j < CR; j++) {
            const int r = 5 * j + rsub;
            cr[j] = (rsub < 5 && r < nvis) ? reinterpret_cast<const float4*>(a.shs)[(size_t)sIdx[wave][r] * 12 + rpart]
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const bool surface = cfg_flag(a.cfg, 0);
    const bool lrn_cam = a.svgss && a.cfg.len >= 0 && cfg_flag(a.cfg, 3);
    float V[16], PR[16], campos[3] = {0.f, 0.f, 0.f};
    float in_color[3] = {0.f, 0.f, 0.f}, in_normal[3] = {0.f, 0.f, 0.f}, in_depth = 0.f, in_m2d[2] = {0.f, 0.f}, in_conic[3] = {0.f, 0.f, 0.f}, in_opacity = 0.f;
    float mean[3] = {0.f, 0.f, 0.f}, c3[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float4 qq = make_float4(0.f, 0.f, 0.f, 0.f);
    float sc_in[3] = {0.f, 0.f, 0.f};
    uint32_t cm = 0;
    if (visible) {
    // (the uniform inputs as well: a load the compiler cannot prove untouched by an earlier store is not a scalar load any more)
#pragma unroll
    for (int i = 0; i < 16; i++) { V[i] = a.view[i]; PR[i] = a.proj[i]; }
    if (a.shs) { campos[0] = a.campos[0]; campos[1] = a.campos[1]; campos[2] = a.campos[2]; }

    // ---------------- every input of this Gaussian ----------------
    if (a.packed) {
        // rgss: the backward composite accumulated this Gaussian's gradients in one packed row (common.hpp GradRowGeom:
        // colour3, normal3, depth, feature S | pad | mean2D.xy, conic.xyz, opacity); it is unpacked into the caller's tensors below
        const int P4 = (7 + a.S + 3) / 4 * 4, RS = (P4 + 6 + 3) / 4 * 4;
        const float* row = a.packed + (size_t)idx * RS;
        // (rows are 16-byte aligned multiples of four floats: the head {colour3, normal3, depth, .} and the tail {mean2D.xy, conic.xyz,
        // opacity, . .} are four 128-bit loads -- the lanes of a wave are a row apart, so it is the number of instructions that costs)
        const float4 h0 = reinterpret_cast<const float4*>(row)[0], h1 = reinterpret_cast<const float4*>(row)[1];
        const float4 t0 = reinterpret_cast<const float4*>(row + P4)[0], t1 = reinterpret_cast<const float4*>(row + P4)[1];
        in_color[0] = h0.x; in_color[1] = h0.y; in_color[2] = h0.z; in_normal[0] = h0.w; in_normal[1] = h1.x; in_normal[2] = h1.y;
        in_depth = h1.z;
        in_m2d[0] = t0.x; in_m2d[1] = t0.y; in_conic[0] = t0.z; in_conic[1] = t0.w; in_conic[2] = t1.x;
        in_opacity = t1.y;
    } else {
#pragma unroll
        for (int c = 0; c < 3; c++) { in_color[c] = a.dL_dcolor[3 * idx + c]; in_normal[c] = a.dL_dnormal[3 * idx + c]; }
        in_depth = a.dL_ddepth[idx];
        in_m2d[0] = a.dL_dmean2D[3 * idx]; in_m2d[1] = a.dL_dmean2D[3 * idx + 1];
        in_conic[0] = a.dL_dconic[4 * idx]; in_conic[1] = a.dL_dconic[4 * idx + 1]; in_conic[2] = a.dL_dconic[4 * idx + 3];
    }
    mean[0] = a.means3D[3 * idx]; mean[1] = a.means3D[3 * idx + 1]; mean[2] = a.means3D[3 * idx + 2];
#pragma unroll
    for (int i = 0; i < 6; i++) c3[i] = a.cov3D[6 * idx + i];
    if (a.shs) cm = a.clamped[idx];
    if (a.scales) {
        qq = reinterpret_cast<const float4*>(a.rotations)[idx];
#pragma unroll
        for (int i = 0; i < 3; i++) sc_in[i] = a.scales[3 * idx + i];
    }
    }   // visible: inputs requested
    if (coop) {   // the coefficient rows -> LDS (the loads were issued first: one memory latency for rows and per-lane inputs together)
#pragma unroll
        for (int j = 0; j < CR; j++) {
            const int r = 5 * j + rsub;
            if (rsub < 5 && r < nvis) sRow[wave][r][rpart] = cr[j];
        }
        for (int r0 = 5 * CR; r0 < nvis; r0 += 5) {   // (more than 30 visible Gaussians in the wave)
            const int r = r0 + rsub;
            if (rsub < 5 && r < nvis)
                sRow[wave][r][rpart] = reinterpret_cast<const float4*>(a.shs)[(size_t)sIdx[wave][r] * 12 + rpart];
        }
    }
    if (visible) {
    const int nk = (a.D + 1) * (a.D + 1);
    float sh[48];
    if (a.shs) {
        const float* shp = a.shs + (size_t)idx * a.M * 3;
        if (coop) {
            wave_lds_sync();   // (the rows are in LDS: the DS operations of a wave execute in order)
#pragma unroll
            for (int i = 0; i < 12; i++) {
                const float4 v = sRow[wave][rank][i];
                sh[4 * i] = v.x; sh[4 * i + 1] = v.y; sh[4 * i + 2] = v.z; sh[4 * i + 3] = v.w;
            }
        } else if (vec) {
#pragma unroll
            for (int i = 0; i < 12; i++) {
                const float4 v = reinterpret_cast<const float4*>(shp)[i];
                sh[4 * i] = v.x; sh[4 * i + 1] = v.y; sh[4 * i + 2] = v.z; sh[4 * i + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 48; i++) sh[i] = i < 3 * nk ? shp[i] : 0.f;
        }
    }
    if (a.packed) {   // (run-time feature width: row -> tensor, behind the loads above)
        const int P4 = (7 + a.S + 3) / 4 * 4, RS = (P4 + 6 + 3) / 4 * 4;
        const float* row = a.packed + (size_t)idx * RS;
        for (int c = 0; c < a.S; c++) a.dL_dfeature[(size_t)idx * a.S + c] = row[7 + c];
    }
    float dmean[3];
    float dcv[6];

    // ---------------- conic -> cov2D -> (cov3D, mean) ----------------
    {
        const float dcon[3] = {in_conic[0], in_conic[1], in_conic[2]};
        float t[3] = {V[0] * mean[0] + V[4] * mean[1] + V[8] * mean[2] + V[12],
                      V[1] * mean[0] + V[5] * mean[1] + V[9] * mean[2] + V[13],
                      V[2] * mean[0] + V[6] * mean[1] + V[10] * mean[2] + V[14]};
        const float limx = 1.3f * a.tanx, limy = 1.3f * a.tany;
        const float txtz = t[0] / t[2], tytz = t[1] / t[2];
        t[0] = fminf(limx, fmaxf(-limx, txtz)) * t[2];
        t[1] = fminf(limy, fmaxf(-limy, tytz)) * t[2];
        const float xgm = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
        const float ygm = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
        const float hx = a.focal_x, hy = a.focal_y;
        const float J0 = hx / t[2], J1 = -(hx * t[0]) / (t[2] * t[2]), J2 = hy / t[2], J3 = -(hy * t[1]) / (t[2] * t[2]);
        Mat3 Jm, Wm, Vk;
        Jm.m[0][0] = J0; Jm.m[0][1] = 0.f; Jm.m[0][2] = J1;
        Jm.m[1][0] = 0.f; Jm.m[1][1] = J2; Jm.m[1][2] = J3;
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
        const float denom = ca * cc - cb * cb;
        float da = 0.f, db = 0.f, dc = 0.f;
        const float d2i = 1.0f / ((denom * denom) + 0.0000001f);
#define TC(i, j) Tm.m[i][j]
#define VC_(i, j) Vk.m[i][j]
        if (d2i != 0) {
            da = d2i * (-cc * cc * dcon[0] + 2 * cb * cc * dcon[1] + (denom - ca * cc) * dcon[2]);
            dc = d2i * (-ca * ca * dcon[2] + 2 * ca * cb * dcon[1] + (denom - ca * cc) * dcon[0]);
            db = d2i * 2 * (cb * cc * dcon[0] - (denom + 2 * cb * cb) * dcon[1] + ca * cb * dcon[2]);
            dcv[0] = (TC(0, 0) * TC(0, 0) * da + TC(0, 0) * TC(1, 0) * db + TC(1, 0) * TC(1, 0) * dc);
            dcv[3] = (TC(0, 1) * TC(0, 1) * da + TC(0, 1) * TC(1, 1) * db + TC(1, 1) * TC(1, 1) * dc);
            dcv[5] = (TC(0, 2) * TC(0, 2) * da + TC(0, 2) * TC(1, 2) * db + TC(1, 2) * TC(1, 2) * dc);
            dcv[1] = 2 * TC(0, 0) * TC(0, 1) * da + (TC(0, 0) * TC(1, 1) + TC(0, 1) * TC(1, 0)) * db + 2 * TC(1, 0) * TC(1, 1) * dc;
            dcv[2] = 2 * TC(0, 0) * TC(0, 2) * da + (TC(0, 0) * TC(1, 2) + TC(0, 2) * TC(1, 0)) * db + 2 * TC(1, 0) * TC(1, 2) * dc;
            dcv[4] = 2 * TC(0, 2) * TC(0, 1) * da + (TC(0, 1) * TC(1, 2) + TC(0, 2) * TC(1, 1)) * db + 2 * TC(1, 1) * TC(1, 2) * dc;
        } else {
#pragma unroll
            for (int i = 0; i < 6; i++) dcv[i] = 0.f;
        }
        const float dT00 = 2 * (TC(0, 0) * VC_(0, 0) + TC(0, 1) * VC_(0, 1) + TC(0, 2) * VC_(0, 2)) * da +
                           (TC(1, 0) * VC_(0, 0) + TC(1, 1) * VC_(0, 1) + TC(1, 2) * VC_(0, 2)) * db;
        const float dT01 = 2 * (TC(0, 0) * VC_(1, 0) + TC(0, 1) * VC_(1, 1) + TC(0, 2) * VC_(1, 2)) * da +
                           (TC(1, 0) * VC_(1, 0) + TC(1, 1) * VC_(1, 1) + TC(1, 2) * VC_(1, 2)) * db;
        const float dT02 = 2 * (TC(0, 0) * VC_(2, 0) + TC(0, 1) * VC_(2, 1) + TC(0, 2) * VC_(2, 2)) * da +
                           (TC(1, 0) * VC_(2, 0) + TC(1, 1) * VC_(2, 1) + TC(1, 2) * VC_(2, 2)) * db;
        const float dT10 = 2 * (TC(1, 0) * VC_(0, 0) + TC(1, 1) * VC_(0, 1) + TC(1, 2) * VC_(0, 2)) * dc +
                           (TC(0, 0) * VC_(0, 0) + TC(0, 1) * VC_(0, 1) + TC(0, 2) * VC_(0, 2)) * db;
        const float dT11 = 2 * (TC(1, 0) * VC_(1, 0) + TC(1, 1) * VC_(1, 1) + TC(1, 2) * VC_(1, 2)) * dc +
                           (TC(0, 0) * VC_(1, 0) + TC(0, 1) * VC_(1, 1) + TC(0, 2) * VC_(1, 2)) * db;
        const float dT12 = 2 * (TC(1, 0) * VC_(2, 0) + TC(1, 1) * VC_(2, 1) + TC(1, 2) * VC_(2, 2)) * dc +
                           (TC(0, 0) * VC_(2, 0) + TC(0, 1) * VC_(2, 1) + TC(0, 2) * VC_(2, 2)) * db;
#undef TC
#undef VC_
        const float dJ00 = Wm.m[0][0] * dT00 + Wm.m[0][1] * dT01 + Wm.m[0][2] * dT02;
        const float dJ02 = Wm.m[2][0] * dT00 + Wm.m[2][1] * dT01 + Wm.m[2][2] * dT02;
        const float dJ11 = Wm.m[1][0] * dT10 + Wm.m[1][1] * dT11 + Wm.m[1][2] * dT12;
        const float dJ12 = Wm.m[2][0] * dT10 + Wm.m[2][1] * dT11 + Wm.m[2][2] * dT12;
        const float tz = 1.f / t[2], tz2 = tz * tz, tz3 = tz2 * tz;
        if (lrn_cam) {
            const float dW[16] = {dT00 * J0, dT10 * J2, dT00 * J1 + dT10 * J3, 0, dT01 * J0, dT11 * J2,
                                  dT01 * J1 + dT11 * J3, 0, dT02 * J0, dT12 * J2, dT02 * J1 + dT12 * J3, 0, 0, 0, 0, 0};
            for (int i = 0; i < 16; i++) if (dW[i] != 0.f) atomic_add_f32(&a.dL_dviewmat[i], dW[i]);
        }
        const float dtx = xgm * -hx * tz2 * dJ02;
        const float dty = ygm * -hy * tz2 * dJ12;
        const float dtz = -hx * tz2 * dJ00 - hy * tz2 * dJ11 + (2 * hx * t[0]) * tz3 * dJ02 + (2 * hy * t[1]) * tz3 * dJ12;
        dmean[0] = V[0] * dtx + V[1] * dty + V[2] * dtz;
        dmean[1] = V[4] * dtx + V[5] * dty + V[6] * dtz;
        dmean[2] = V[8] * dtx + V[9] * dty + V[10] * dtz;
    }

    // ---------------- mean2D, depth -> mean ----------------
    {
        const float mhw = PR[3] * mean[0] + PR[7] * mean[1] + PR[11] * mean[2] + PR[15];
        const float mw = 1.0f / (mhw + 0.0000001f);
        const float mul1 = (PR[0] * mean[0] + PR[4] * mean[1] + PR[8] * mean[2] + PR[12]) * mw * mw;
        const float mul2 = (PR[1] * mean[0] + PR[5] * mean[1] + PR[9] * mean[2] + PR[13]) * mw * mw;
        const float g2x = in_m2d[0], g2y = in_m2d[1];
        float dm[3];
        dm[0] = (PR[0] * mw - PR[3] * mul1) * g2x + (PR[1] * mw - PR[3] * mul2) * g2y;
        dm[1] = (PR[4] * mw - PR[7] * mul1) * g2x + (PR[5] * mw - PR[7] * mul2) * g2y;
        dm[2] = (PR[8] * mw - PR[11] * mul1) * g2x + (PR[9] * mw - PR[11] * mul2) * g2y;
        const float dd = in_depth;
        const float fd[3] = {dd * V[2], dd * V[6], dd * V[10]};
#pragma unroll
        for (int i = 0; i < 3; i++) dmean[i] += dm[i] + fd[i];
        if (lrn_cam) {
            const float pm[16] = {g2x * mean[0] * mw, g2y * mean[0] * mw, 0, g2x * -mul1 * mean[0] + g2y * -mul2 * mean[0],
                                  g2x * mean[1] * mw, g2y * mean[1] * mw, 0, g2x * -mul1 * mean[1] + g2y * -mul2 * mean[1],
                                  g2x * mean[2] * mw, g2y * mean[2] * mw, 0, g2x * -mul1 * mean[2] + g2y * -mul2 * mean[2],
                                  g2x * mw, g2y * mw, 0, g2x * -mul1 + g2y * -mul2};
            const float vd[16] = {0, 0, dd * mean[0], 0, 0, 0, dd * mean[1], 0, 0, 0, dd * mean[2], 0, 0, 0, dd, 0};
            for (int i = 0; i < 16; i++) {
                if (pm[i] != 0.f) atomic_add_f32(&a.dL_dprojmat[i], pm[i]);
                if (vd[i] != 0.f) atomic_add_f32(&a.dL_dviewmat[i], vd[i]);
            }
        }
    }

    // ---------------- colour -> SH, view direction -> mean ----------------
    if (a.shs) {
        const float kC0 = 0.28209479177387814f, kC1 = 0.4886025119029199f;
        const float dor[3] = {mean[0] - campos[0], mean[1] - campos[1], mean[2] - campos[2]};
        const float len = sqrtf(dor[0] * dor[0] + dor[1] * dor[1] + dor[2] * dor[2]);
        const float x = dor[0] / len, y = dor[1] / len, z = dor[2] / len;
        float g[3];
#pragma unroll
        for (int c = 0; c < 3; c++) g[c] = in_color[c] * (((cm >> c) & 1u) ? 0.f : 1.f);
        float ddx[3] = {0, 0, 0}, ddy[3] = {0, 0, 0}, ddz[3] = {0, 0, 0};
        float cf[16];
#pragma unroll
        for (int k = 0; k < 16; k++) cf[k] = 0.f;
#define SET(k, coef) cf[k] = (coef);
#define SH(k, c) sh[3 * (k) + (c)]
        SET(0, kC0)
        if (a.D > 0) {
            SET(1, -kC1 * y) SET(2, kC1 * z) SET(3, -kC1 * x)
#pragma unroll
            for (int c = 0; c < 3; c++) { ddx[c] = -kC1 * SH(3, c); ddy[c] = -kC1 * SH(1, c); ddz[c] = kC1 * SH(2, c); }
            if (a.D > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                SET(4, gC2[0] * xy) SET(5, gC2[1] * yz) SET(6, gC2[2] * (2.f * zz - xx - yy)) SET(7, gC2[3] * xz)
                SET(8, gC2[4] * (xx - yy))
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    ddx[c] += gC2[0] * y * SH(4, c) + gC2[2] * 2.f * -x * SH(6, c) + gC2[3] * z * SH(7, c) + gC2[4] * 2.f * x * SH(8, c);
                    ddy[c] += gC2[0] * x * SH(4, c) + gC2[1] * z * SH(5, c) + gC2[2] * 2.f * -y * SH(6, c) + gC2[4] * 2.f * -y * SH(8, c);
                    ddz[c] += gC2[1] * y * SH(5, c) + gC2[2] * 2.f * 2.f * z * SH(6, c) + gC2[3] * x * SH(7, c);
                }
                if (a.D > 2) {
                    SET(9, gC3[0] * y * (3.f * xx - yy)) SET(10, gC3[1] * xy * z) SET(11, gC3[2] * y * (4.f * zz - xx - yy))
                    SET(12, gC3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy)) SET(13, gC3[4] * x * (4.f * zz - xx - yy))
                    SET(14, gC3[5] * z * (xx - yy)) SET(15, gC3[6] * x * (xx - 3.f * yy))
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        ddx[c] += (gC3[0] * SH(9, c) * 3.f * 2.f * xy + gC3[1] * SH(10, c) * yz + gC3[2] * SH(11, c) * -2.f * xy +
                                   gC3[3] * SH(12, c) * -3.f * 2.f * xz + gC3[4] * SH(13, c) * (-3.f * xx + 4.f * zz - yy) +
                                   gC3[5] * SH(14, c) * 2.f * xz + gC3[6] * SH(15, c) * 3.f * (xx - yy));
                        ddy[c] += (gC3[0] * SH(9, c) * 3.f * (xx - yy) + gC3[1] * SH(10, c) * xz +
                                   gC3[2] * SH(11, c) * (-3.f * yy + 4.f * zz - xx) + gC3[3] * SH(12, c) * -3.f * 2.f * yz +
                                   gC3[4] * SH(13, c) * -2.f * xy + gC3[5] * SH(14, c) * -2.f * yz + gC3[6] * SH(15, c) * -3.f * 2.f * xy);
                        ddz[c] += (gC3[1] * SH(10, c) * xy + gC3[2] * SH(11, c) * 4.f * 2.f * yz +
                                   gC3[3] * SH(12, c) * 3.f * (2.f * zz - xx - yy) + gC3[4] * SH(13, c) * 4.f * 2.f * xz +
                                   gC3[5] * SH(14, c) * (xx - yy));
                    }
                }
            }
        }
#undef SET
#undef SH
        // dL_dsh[k][c] = cf[k] * g[c]; coefficients above the active degree are left untouched (the caller zero-fills)
        float* dsh = a.dL_dsh + (size_t)idx * a.M * 3;
        if (coop) {   // (the gradient row leaves row-wise, behind the per-Gaussian part: below)
            float o[48];
#pragma unroll
            for (int k = 0; k < 16; k++) { o[3 * k] = cf[k] * g[0]; o[3 * k + 1] = cf[k] * g[1]; o[3 * k + 2] = cf[k] * g[2]; }
#pragma unroll
            for (int i = 0; i < 12; i++) sRow[wave][rank][i] = make_float4(o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                if (k < nk) { dsh[3 * k] = cf[k] * g[0]; dsh[3 * k + 1] = cf[k] * g[1]; dsh[3 * k + 2] = cf[k] * g[2]; }
            }
        }
        const float ddir[3] = {ddx[0] * g[0] + ddx[1] * g[1] + ddx[2] * g[2], ddy[0] * g[0] + ddy[1] * g[1] + ddy[2] * g[2],
                               ddz[0] * g[0] + ddz[1] * g[1] + ddz[2] * g[2]};
        const float s2 = dor[0] * dor[0] + dor[1] * dor[1] + dor[2] * dor[2];
        const float i32 = 1.0f / sqrtf(s2 * s2 * s2);
        const float dm[3] = {((+s2 - dor[0] * dor[0]) * ddir[0] - dor[1] * dor[0] * ddir[1] - dor[2] * dor[0] * ddir[2]) * i32,
                             (-dor[0] * dor[1] * ddir[0] + (s2 - dor[1] * dor[1]) * ddir[1] - dor[2] * dor[1] * ddir[2]) * i32,
                             (-dor[0] * dor[2] * ddir[0] - dor[1] * dor[2] * ddir[1] + (s2 - dor[2] * dor[2]) * ddir[2]) * i32};
        if (lrn_cam)
            for (int i = 0; i < 3; i++) if (dm[i] != 0.f) atomic_add_f32(&a.dL_dcampos[i], -dm[i]);
#pragma unroll
        for (int i = 0; i < 3; i++) dmean[i] += dm[i];
    }

    // ---------------- cov3D -> scale, rotation (+ normal gradient into R) ----------------
    float dsc[3] = {0.f, 0.f, 0.f};
    float4 dq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.scales) {
        const float r = qq.x, x = qq.y, y = qq.z, z = qq.w;
        Mat3 Rm;
        Rm.m[0][0] = 1.f - 2.f * (y * y + z * z); Rm.m[0][1] = 2.f * (x * y - r * z); Rm.m[0][2] = 2.f * (x * z + r * y);
        Rm.m[1][0] = 2.f * (x * y + r * z); Rm.m[1][1] = 1.f - 2.f * (x * x + z * z); Rm.m[1][2] = 2.f * (y * z - r * x);
        Rm.m[2][0] = 2.f * (x * z - r * y); Rm.m[2][1] = 2.f * (y * z + r * x); Rm.m[2][2] = 1.f - 2.f * (x * x + y * y);
        const float s[3] = {a.scale_modifier * sc_in[0], a.scale_modifier * sc_in[1], a.scale_modifier * sc_in[2]};
        Mat3 Mm;  // S * R with S = diag(s): M[c][row] = s[row] * R[c][row]
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int row = 0; row < 3; row++) Mm.m[c][row] = s[row] * Rm.m[c][row];
        Mat3 dSig, twoM;
        dSig.m[0][0] = dcv[0]; dSig.m[0][1] = 0.5f * dcv[1]; dSig.m[0][2] = 0.5f * dcv[2];
        dSig.m[1][0] = 0.5f * dcv[1]; dSig.m[1][1] = dcv[3]; dSig.m[1][2] = 0.5f * dcv[4];
        dSig.m[2][0] = 0.5f * dcv[2]; dSig.m[2][1] = 0.5f * dcv[4]; dSig.m[2][2] = dcv[5];
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int row = 0; row < 3; row++) twoM.m[c][row] = 2.0f * Mm.m[c][row];
        const Mat3 dM = mmul(twoM, dSig);
        const Mat3 Rt = mtr(Rm), dMt = mtr(dM);
        dsc[0] = Rt.m[0][0] * dMt.m[0][0] + Rt.m[0][1] * dMt.m[0][1] + Rt.m[0][2] * dMt.m[0][2];
        dsc[1] = Rt.m[1][0] * dMt.m[1][0] + Rt.m[1][1] * dMt.m[1][1] + Rt.m[1][2] * dMt.m[1][2];
        dsc[2] = surface ? 0.f : (Rt.m[2][0] * dMt.m[2][0] + Rt.m[2][1] * dMt.m[2][1] + Rt.m[2][2] * dMt.m[2][2]);
        Mat3 dRt = dMt;
#pragma unroll
        for (int k = 0; k < 3; k++) { dRt.m[0][k] *= s[0]; dRt.m[1][k] *= s[1]; dRt.m[2][k] *= s[2]; }
        const float gn[3] = {in_normal[0], in_normal[1], in_normal[2]};
        dRt.m[2][0] += gn[0] * V[0] + gn[1] * V[1] + gn[2] * V[2];
        dRt.m[2][1] += gn[0] * V[4] + gn[1] * V[5] + gn[2] * V[6];
        dRt.m[2][2] += gn[0] * V[8] + gn[1] * V[9] + gn[2] * V[10];
        if (lrn_cam) {
            const float wN[3] = {Rm.m[0][2], Rm.m[1][2], Rm.m[2][2]};
            const float dv[16] = {gn[0] * wN[0], gn[1] * wN[0], gn[2] * wN[0], 0, gn[0] * wN[1], gn[1] * wN[1], gn[2] * wN[1], 0,
                                  gn[0] * wN[2], gn[1] * wN[2], gn[2] * wN[2], 0, 0, 0, 0, 0};
            for (int i = 0; i < 16; i++) if (dv[i] != 0.f) atomic_add_f32(&a.dL_dviewmat[i], dv[i]);
        }
#define DR(i, j) dRt.m[i][j]
        dq.x = 2 * z * (DR(0, 1) - DR(1, 0)) + 2 * y * (DR(2, 0) - DR(0, 2)) + 2 * x * (DR(1, 2) - DR(2, 1));
        dq.y = 2 * y * (DR(1, 0) + DR(0, 1)) + 2 * z * (DR(2, 0) + DR(0, 2)) + 2 * r * (DR(1, 2) - DR(2, 1)) - 4 * x * (DR(2, 2) + DR(1, 1));
        dq.z = 2 * x * (DR(1, 0) + DR(0, 1)) + 2 * r * (DR(2, 0) - DR(0, 2)) + 2 * z * (DR(1, 2) + DR(2, 1)) - 4 * y * (DR(2, 2) + DR(0, 0));
        dq.w = 2 * r * (DR(0, 1) - DR(1, 0)) + 2 * x * (DR(2, 0) + DR(0, 2)) + 2 * y * (DR(1, 2) + DR(2, 1)) - 4 * z * (DR(1, 1) + DR(0, 0));
#undef DR
    }

    // ---------------- every gradient of this Gaussian ----------------
    if (a.packed) {
#pragma unroll
        for (int c = 0; c < 3; c++) { a.dL_dcolor[3 * idx + c] = in_color[c]; a.dL_dnormal[3 * idx + c] = in_normal[c]; }
        a.dL_ddepth[idx] = in_depth;
        a.dL_dmean2D[3 * idx] = in_m2d[0]; a.dL_dmean2D[3 * idx + 1] = in_m2d[1];
        a.dL_dconic[4 * idx] = in_conic[0]; a.dL_dconic[4 * idx + 1] = in_conic[1]; a.dL_dconic[4 * idx + 3] = in_conic[2];
        a.dL_dopacity[idx] = in_opacity;
    }
#pragma unroll
    for (int i = 0; i < 6; i++) a.dL_dcov3D[6 * idx + i] = dcv[i];
#pragma unroll
    for (int i = 0; i < 3; i++) a.dL_dmean3D[3 * idx + i] = dmean[i];
    if (a.scales) {
#pragma unroll
        for (int i = 0; i < 3; i++) a.dL_dscale[3 * idx + i] = dsc[i];
        reinterpret_cast<float4*>(a.dL_drot)[idx] = dq;
    }
    }   // visible
    if (coop) {   // dL_dsh rows of the wave's visible Gaussians, 12 lanes per row
        wave_lds_sync();
        for (int r0 = 0; r0 < nvis; r0 += 5) {
            const int r = r0 + rsub;
            if (rsub < 5 && r < nvis)
                reinterpret_cast<float4*>(a.dL_dsh)[(size_t)sIdx[wave][r] * 12 + rpart] = sRow[wave][r][rpart];
        }
    }
}

}  // namespace

void launch_geom_bwd(const GeomBwdArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(geom_bwd_kernel, dim3((a.P + GB_BLOCK - 1) / GB_BLOCK), dim3(GB_BLOCK), 0, s, a);
}

}  // namespace svgir
