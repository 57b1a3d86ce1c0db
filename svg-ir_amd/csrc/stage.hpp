// svg-ir_amd/csrc/stage.hpp -- LDS staging of a batch of splats, shared by the forward and backward composite
// kernels.
//
// Staged layout of one splat (floats):  [0,24) the record written by preprocess.hip (common.hpp RecField),
// [24, 24+S4) the S feature floats padded to a float4 boundary, [24+S4, NF) the VS vfeature floats.
// Every slot starts on a 16-byte boundary, so the walking waves read it with broadcast ds_read_b128.
#pragma once
#include "common.hpp"

namespace svgir {

template <int S, int VC>
struct StageGeom {
    static constexpr int S4 = (S + 3) / 4 * 4;      // feature slots padded to a float4 boundary
    static constexpr int NFD = REC + S4 + VC * 4;    // floats of data per staged splat
    // slot stride: an ODD number of float4s, so that 64 lanes reading the headers of 64 consecutive slots with
    // ds_read_b128 fall on distinct 16-byte LDS slots (bank-conflict free), while a wave-uniform read broadcasts
    static constexpr int NF4 = (NFD / 4) | 1;
    static constexpr int NF = NF4 * 4;
    static constexpr int C4 = 6 + VC;                // float4 chunks gathered with 16-byte loads (record + vfeatures)
    static constexpr int BATCH = NF <= 32 ? 256 : (NF <= 64 ? 192 : 128);  // <= 48 KB of LDS per workgroup
    static constexpr int F_OFF = REC;                // features
    static constexpr int V_OFF = REC + S4;           // vfeatures
    static constexpr size_t lds_bytes() { return (size_t)BATCH * NF * 4 + (size_t)BATCH * 4; }
};

#if defined(__HIPCC__)
// Conservative wave-level cull.  Lane-parallel over splats: does the splat with mean (mx,my), conic (a,b,c) and
// opacity `op` reach alpha >= 1/255 anywhere inside the pixel rectangle [X0,X1]x[Y0,Y1]?  The per-pixel test is
// q(d) = a dx^2 + 2 b dx dy + c dy^2 <= 2 ln(255 op); the minimum of the convex form over the rectangle is 0 when
// the mean is inside, else it lies on one of the four edges.  A slack absorbs fp32 rounding of the per-pixel
// evaluation, so a splat is only ever dropped when no pixel of the rectangle can pass the exact test.
__device__ __forceinline__ bool splat_may_touch(float mx, float my, float a, float b, float c, float op, float X0,
                                                float Y0, float X1, float Y1) {
    const float u0 = mx - X1, u1 = mx - X0, v0 = my - Y1, v1 = my - Y0;  // ranges of dx, dy
    const float tau2 = 2.f * __logf(255.f * op);                         // NaN / -inf for op <= 0 => never passes
    if (!(a > 0.f) || !(c > 0.f) || !(a * c - b * b > 0.f)) return tau2 == tau2;  // degenerate conic: keep
    float qmin = 0.f;
    const bool in = (u0 <= 0.f) && (u1 >= 0.f) && (v0 <= 0.f) && (v1 >= 0.f);
    if (!in) {
        const float ic = 1.f / c, ia = 1.f / a;
        float v, u, q;
        v = fminf(v1, fmaxf(v0, -b * u0 * ic)); qmin = a * u0 * u0 + 2.f * b * u0 * v + c * v * v;
        v = fminf(v1, fmaxf(v0, -b * u1 * ic)); q = a * u1 * u1 + 2.f * b * u1 * v + c * v * v; qmin = fminf(qmin, q);
        u = fminf(u1, fmaxf(u0, -b * v0 * ia)); q = a * u * u + 2.f * b * u * v0 + c * v0 * v0; qmin = fminf(qmin, q);
        u = fminf(u1, fmaxf(u0, -b * v1 * ia)); q = a * u * u + 2.f * b * u * v1 + c * v1 * v1; qmin = fminf(qmin, q);
    }
    const float um = fmaxf(fabsf(u0), fabsf(u1)), vm = fmaxf(fabsf(v0), fabsf(v1));
    const float slack = 0.02f + 4e-5f * (a * um * um + 2.f * fabsf(b) * um * vm + c * vm * vm);
    return qmin <= tau2 + slack;
}

// Gathers splats [0, n) of the current batch into LDS.  ids[s] must already hold the Gaussian id of slot s
// (written before a barrier).  All 256 threads take part; loads are 16 bytes wide except the S feature floats
// (rows of S floats are not 16-byte aligned in the caller's [P,S] tensor).
template <int S, int VC>
__device__ __forceinline__ void stage_batch(float* __restrict__ sD, const int* __restrict__ ids, int n,
                                            const float* __restrict__ rec, const float* __restrict__ feat,
                                            const float* __restrict__ vfeat) {
    using SG = StageGeom<S, VC>;
    float4* sD4 = reinterpret_cast<float4*>(sD);
    const float4* rec4 = reinterpret_cast<const float4*>(rec);
    const float4* vf4 = reinterpret_cast<const float4*>(vfeat);
    const int total = n * SG::C4;
    for (int k = threadIdx.x; k < total; k += BLOCK) {
        const int s = k / SG::C4, part = k - s * SG::C4;
        const size_t id = (size_t)ids[s];
        if (part < 6) sD4[s * SG::NF4 + part] = rec4[id * 6 + part];
        else sD4[s * SG::NF4 + SG::V_OFF / 4 + (part - 6)] = vf4[id * VC + (part - 6)];
    }
    if (S > 0) {
        const int totf = n * S;
        for (int k = threadIdx.x; k < totf; k += BLOCK) {
            const int s = k / S, c = k - s * S;
            sD[s * SG::NF + SG::F_OFF + c] = feat[(size_t)ids[s] * S + c];
        }
    }
}
#endif

}  // namespace svgir
