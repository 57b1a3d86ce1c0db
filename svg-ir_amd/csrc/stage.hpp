// svg-ir_amd/csrc/stage.hpp -- wave-level culling and LDS staging shared by the forward and backward composite
// kernels.
//
// Work decomposition: ONE wave64 per 8x8-pixel sub-tile (4 per 16x16 tile), no workgroup barriers.  A cull kernel
// (render_fwd.hip) tests every entry of a tile's depth-ordered splat list lane-parallel (one splat per lane) against the
// four 8x8 pixel rectangles of the tile and writes the survivors ("candidates") to one compact list per sub-tile.
// Candidates are then staged CH at a time: the 96-byte record written by preprocess.hip plus the S feature and VS
// vfeature floats are gathered into LDS with 16-byte loads (all loads of a batch in flight before the first LDS store,
// issued one batch ahead of their use) and every candidate is consumed with wave-uniform (broadcast) ds_read_b128.
//
// Staged layout of one candidate (floats):  [0,24) the record (common.hpp RecField), [24, 24+S4) the S feature
// floats padded to a float4 boundary, [24+S4, NFD) the VS vfeature floats; slot stride NF = an odd number of float4s.
#pragma once
#include "common.hpp"

namespace svgir {

template <int S, int VC>
struct StageGeom {
    static constexpr int S4 = (S + 3) / 4 * 4;
    static constexpr int NFD = REC + S4 + VC * 4;    // floats of data per staged candidate
    static constexpr int NF4 = (NFD / 4) | 1;        // slot stride in float4 (odd => conflict-free column access)
    static constexpr int NF = NF4 * 4;
    static constexpr int C4 = 6 + VC;                // 16-byte chunks gathered per candidate (record + vfeatures)
    static constexpr int CH = NF <= 48 ? 64 : 32;    // candidates staged per batch (<= ~13 KB of LDS per wave)
    static constexpr int QN = 2 * CH;                // {gid, slot} entries of the current and the next staging batch
#ifndef FWD_KB_V
#define FWD_KB_V 4
#endif
#ifndef FWD_KB_P
#define FWD_KB_P 4
#endif
    static constexpr int KB = VC > 0 ? FWD_KB_V : FWD_KB_P;   // candidates blended per branch-free group (divides CH)
#ifndef FWD_WPE_V
#define FWD_WPE_V 2
#endif
#ifndef FWD_WPE_P
#define FWD_WPE_P 3
#endif
    static constexpr int WPE = VC > 0 ? FWD_WPE_V : FWD_WPE_P;   // waves per SIMD the forward's register budget is held to
    static constexpr int F_OFF = REC;
    static constexpr int V_OFF = REC + S4;
    static constexpr size_t lds_bytes() { return (size_t)CH * NF * 4 + (size_t)QN * 8 + (size_t)2 * CH * 4; }   // + per-candidate weight sums (this / previous batch)
    static_assert(SEG % CH == 0, "segment boundaries must fall on staging-batch boundaries");
    static_assert(CH % KB == 0, "a staging batch is a whole number of blend batches");
};

#if defined(__HIPCC__)
// Exponent of the Gaussian falloff of a (pixel, splat) pair, rounded exactly like a plain fp32 evaluation of the
// reference's source (svgss forward.cu:534-535: -0.5 * ((a dx dx + c dy dy) + 2 b dx dy); rgss forward.cu:430:
// -0.5 * (a dx dx + c dy dy) - b dx dy) without FMA contraction of the products: forward and backward must take
// identical alpha >= 1/255 decisions, and that evaluation is what the parity oracle computes.  Both variants round
// the same real number once in their last step (scaling by 2 or 0.5 is exact), so one formula serves both:
// s = ((a dx) dx + (c dy) dy), m = (b dx) dy, power = round(-0.5 s - m).
__device__ __forceinline__ float pair_power(float a, float b, float c, float dx, float dy) {
    float s, m;
    {
#pragma clang fp contract(off)
        s = a * dx * dx + c * dy * dy;
        m = b * dx * dy;
    }
    return __builtin_fmaf(-0.5f, s, -m);
}

// exp(x) for x <= 0 to ~1 ulp in 6 instructions: the hardware exp2 is evaluated at the rounded product t = x log2(e)
// and corrected to first order for the rounding of t (residual r = x log2(e) - t, exact through an FMA and the low
// word of log2(e)): 2^(t + r) = 2^t (1 + r ln 2 + O(r^2)), |r| < 2^-21.  No range reduction is needed: v_exp_f32
// covers the whole range and flushes to 0 where exp underflows.
__device__ __forceinline__ float exp_nonpos(float x) {
    const float L2E_HI = 1.44269502162933349609375f;    // fp32(log2 e)
    const float L2E_LO = 1.92596299112661746e-8f;       // log2 e - fp32(log2 e)
    const float t = x * L2E_HI;
    float r = __builtin_fmaf(x, L2E_HI, -t);
    r = __builtin_fmaf(x, L2E_LO, r);
    const float e = __builtin_amdgcn_exp2f(t);
    return __builtin_fmaf(e, r * 0.693147182464599609375f, e);
}

// Conservative cull of a splat against pixel rectangles.  Does the splat with mean (mx,my), conic (a,b,c) and opacity
// `op` reach alpha >= 1/255 anywhere inside the pixel rectangle [X0,X1]x[Y0,Y1]?  The per-pixel test is
// q(d) = a dx^2 + 2 b dx dy + c dy^2 <= 2 ln(255 op); the minimum of the convex form over the rectangle is 0 when
// the mean is inside, else it lies on one of the four edges.  A slack absorbs fp32 rounding of the per-pixel
// evaluation (and of the hardware reciprocal / logarithm used here), so a splat is only ever dropped when no pixel
// of the rectangle can pass the exact test.  The per-splat part is computed once and tested against several rectangles.
struct SplatCull {
    float mx, my, a, b, c, tau2, ia, ic;
    bool degenerate;   // not a positive-definite conic: keep whenever the threshold is a number
};
__device__ __forceinline__ SplatCull cull_prepare(float mx, float my, float a, float b, float c, float op) {
    SplatCull s;
    s.mx = mx; s.my = my; s.a = a; s.b = b; s.c = c;
    s.tau2 = 2.f * 0.693147182f * __builtin_amdgcn_logf(255.f * op);   // NaN / -inf for op <= 0 => never passes
    s.degenerate = !(a > 0.f) || !(c > 0.f) || !(a * c - b * b > 0.f);
    s.ia = __builtin_amdgcn_rcpf(a); s.ic = __builtin_amdgcn_rcpf(c);
    return s;
}
__device__ __forceinline__ bool cull_test(const SplatCull& s, float X0, float Y0, float X1, float Y1) {
    const float u0 = s.mx - X1, u1 = s.mx - X0, v0 = s.my - Y1, v1 = s.my - Y0;  // ranges of dx, dy
    const float a = s.a, b = s.b, c = s.c;
    const bool in = (u0 <= 0.f) && (u1 >= 0.f) && (v0 <= 0.f) && (v1 >= 0.f);
    float v, u, q, qmin;
    v = fminf(v1, fmaxf(v0, -b * u0 * s.ic)); qmin = a * u0 * u0 + 2.f * b * u0 * v + c * v * v;
    v = fminf(v1, fmaxf(v0, -b * u1 * s.ic)); q = a * u1 * u1 + 2.f * b * u1 * v + c * v * v; qmin = fminf(qmin, q);
    u = fminf(u1, fmaxf(u0, -b * v0 * s.ia)); q = a * u * u + 2.f * b * u * v0 + c * v0 * v0; qmin = fminf(qmin, q);
    u = fminf(u1, fmaxf(u0, -b * v1 * s.ia)); q = a * u * u + 2.f * b * u * v1 + c * v1 * v1; qmin = fminf(qmin, q);
    qmin = in ? 0.f : qmin;
    const float um = fmaxf(fabsf(u0), fabsf(u1)), vm = fmaxf(fabsf(v0), fabsf(v1));
    const float slack = 0.02f + 4e-5f * (a * um * um + 2.f * fabsf(b) * um * vm + c * vm * vm);
    return s.degenerate ? (s.tau2 == s.tau2) : (qmin <= s.tau2 + slack);
}

// Wave-level gather of m (<= CHN) candidates into LDS slots [0, m), split in two halves so that the global loads of
// the NEXT batch can be in flight while the current one is consumed:
//   stage_load : all 16-byte record / vfeature loads and the feature loads of the batch into registers
//                (`gid_of(s)` returns the Gaussian id of the candidate that goes to slot s; loads are unconditional
//                with clamped indices, which keeps everything in registers);
//   stage_store: registers -> LDS (predicated).
// The caller separates the stores from the subsequent LDS reads with wave_lds_sync().
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int S, int VC, int CHN>
struct StageRegs {
    using SG = StageGeom<S, VC>;
    static constexpr int KV = (CHN * SG::C4 + 63) / 64;  // 16-byte loads per lane per batch
    static constexpr int KF = (CHN * S + 63) / 64;       // feature floats per lane per batch
    // native vector type: a HIP float4 is a struct, and struct copies global -> register array -> LDS are lowered to
    // memcpys that keep the whole array in scratch (every gather then waits for its own latency before it is spilled)
    f32x4 v[KV];
    float fv[KF > 0 ? KF : 1];
};

template <int S, int VC, int CHN, typename GidOf>
__device__ __forceinline__ void stage_load(StageRegs<S, VC, CHN>& r, int m, GidOf gid_of, int lane,
                                           const float* __restrict__ rec, const float* __restrict__ feat,
                                           const float* __restrict__ vfeat) {
    using SG = StageGeom<S, VC>;
    using SR = StageRegs<S, VC, CHN>;
    const f32x4* rec4 = reinterpret_cast<const f32x4*>(rec);
    const f32x4* vf4 = reinterpret_cast<const f32x4*>(vfeat);
    const int total = m * SG::C4;  // >= C4 (m >= 1)
#pragma unroll
    for (int u = 0; u < SR::KV; u++) {
        const int k = min(u * 64 + lane, total - 1);
        const int s = k / SG::C4, part = k - s * SG::C4;
        const size_t id = (size_t)gid_of(s);
        const f32x4* src = part < 6 ? rec4 + id * 6 + part : vf4 + id * VC + (part - 6);
        r.v[u] = *src;
    }
    if (S > 0) {
        const int totf = m * S;
#pragma unroll
        for (int u = 0; u < SR::KF; u++) {
            const int k = min(u * 64 + lane, totf - 1);
            const int s = k / S, c = k - s * S;
            r.fv[u] = feat[(size_t)gid_of(s) * S + c];
        }
    }
}

template <int S, int VC, int CHN>
__device__ __forceinline__ void stage_store(const StageRegs<S, VC, CHN>& r, float* __restrict__ sD, int m, int lane) {
    using SG = StageGeom<S, VC>;
    using SR = StageRegs<S, VC, CHN>;
    f32x4* sD4 = reinterpret_cast<f32x4*>(sD);
    const int total = m * SG::C4;
#pragma unroll
    for (int u = 0; u < SR::KV; u++) {
        const int k = u * 64 + lane;
        if (k < total) {
            const int s = k / SG::C4, part = k - s * SG::C4;
            sD4[s * SG::NF4 + (part < 6 ? part : SG::V_OFF / 4 + (part - 6))] = r.v[u];
        }
    }
    if (S > 0) {
        const int totf = m * S;
#pragma unroll
        for (int u = 0; u < SR::KF; u++) {
            const int k = u * 64 + lane;
            if (k < totf) {
                const int s = k / S, c = k - s * S;
                sD[s * SG::NF + SG::F_OFF + c] = r.fv[u];
            }
        }
    }
}

template <int S, int VC, int CHN, typename GidOf>
__device__ __forceinline__ void stage_candidates(float* __restrict__ sD, int m, GidOf gid_of, int lane,
                                                 const float* __restrict__ rec, const float* __restrict__ feat,
                                                 const float* __restrict__ vfeat) {
    StageRegs<S, VC, CHN> r;
    stage_load<S, VC, CHN>(r, m, gid_of, lane, rec, feat, vfeat);
    // Keep the compiler from sinking each load into its predicated LDS store (which serialises load -> wait -> store
    // per 16-byte chunk): every loaded value has to be live here, so all loads of the batch are in flight together.
#pragma unroll
    for (int u = 0; u < StageRegs<S, VC, CHN>::KV; u++)
        asm volatile("" : "+v"(r.v[u]));
    if (S > 0) {
#pragma unroll
        for (int u = 0; u < StageRegs<S, VC, CHN>::KF; u++) asm volatile("" : "+v"(r.fv[u]));
    }
    stage_store<S, VC, CHN>(r, sD, m, lane);
}

#endif

}  // namespace svgir
