// svg-ir_amd/csrc/stage.hpp -- wave-level culling and LDS staging shared by the forward and backward composite
// kernels.
//
// Work decomposition: ONE wave64 per 8x8-pixel sub-tile (4 per 16x16 tile), no workgroup barriers.  A wave scans its
// tile's depth-ordered splat list 64 entries at a time, culls lane-parallel (one splat per lane) against its 8x8
// pixel rectangle, and queues the survivors ("candidates") in a small LDS ring.  Candidates are then staged CH at a
// time: the 96-byte record written by preprocess.hip plus the S feature and VS vfeature floats are gathered into LDS
// with 16-byte loads (all loads of a batch in flight before the first LDS store) and every candidate is then consumed
// with wave-uniform (broadcast) ds_read_b128.
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
    static constexpr int QN = 128;                   // candidate ring (>= CH + 64)
    static constexpr int F_OFF = REC;
    static constexpr int V_OFF = REC + S4;
    static constexpr size_t lds_bytes() { return (size_t)CH * NF * 4 + (size_t)QN * 8; }
    static_assert(SEG % CH == 0, "segment boundaries must fall on staging-batch boundaries");
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

// Wave-level gather of m (<= CHN) candidates into LDS slots [0, m), split in two halves so that the global loads of
// the NEXT batch can be in flight while the current one is consumed:
//   stage_load : all 16-byte record / vfeature loads and the feature loads of the batch into registers
//                (`gid_of(s)` returns the Gaussian id of the candidate that goes to slot s; loads are unconditional
//                with clamped indices, which keeps everything in registers);
//   stage_store: registers -> LDS (predicated).
// The caller separates the stores from the subsequent LDS reads with wave_lds_sync().
template <int S, int VC, int CHN>
struct StageRegs {
    using SG = StageGeom<S, VC>;
    static constexpr int KV = (CHN * SG::C4 + 63) / 64;  // 16-byte loads per lane per batch
    static constexpr int KF = (CHN * S + 63) / 64;       // feature floats per lane per batch
    float4 v[KV];
    float fv[KF > 0 ? KF : 1];
};

template <int S, int VC, int CHN, typename GidOf>
__device__ __forceinline__ void stage_load(StageRegs<S, VC, CHN>& r, int m, GidOf gid_of, int lane,
                                           const float* __restrict__ rec, const float* __restrict__ feat,
                                           const float* __restrict__ vfeat) {
    using SG = StageGeom<S, VC>;
    using SR = StageRegs<S, VC, CHN>;
    const float4* rec4 = reinterpret_cast<const float4*>(rec);
    const float4* vf4 = reinterpret_cast<const float4*>(vfeat);
    const int total = m * SG::C4;  // >= C4 (m >= 1)
#pragma unroll
    for (int u = 0; u < SR::KV; u++) {
        const int k = min(u * 64 + lane, total - 1);
        const int s = k / SG::C4, part = k - s * SG::C4;
        const size_t id = (size_t)gid_of(s);
        const float4* src = part < 6 ? rec4 + id * 6 + part : vf4 + id * VC + (part - 6);
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
    float4* sD4 = reinterpret_cast<float4*>(sD);
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
        asm volatile("" : "+v"(r.v[u].x), "+v"(r.v[u].y), "+v"(r.v[u].z), "+v"(r.v[u].w));
    if (S > 0) {
#pragma unroll
        for (int u = 0; u < StageRegs<S, VC, CHN>::KF; u++) asm volatile("" : "+v"(r.fv[u]));
    }
    stage_store<S, VC, CHN>(r, sD, m, lane);
}

// Block -> (tile, sub-tile) mapping: the four waves of a tile get block ids congruent mod 8, i.e. they run on the
// same XCD (blocks are dispatched round-robin over the 8 XCDs) and share that XCD's L2 for the records they gather.
// Tiles are taken in descending order of their list length (tile_order), so the longest sequential walks start
// first and the short ones fill the tail of the launch.
__device__ __forceinline__ void sub_tile_of_block(int b, int T, const uint32_t* __restrict__ tile_order, int& tile,
                                                  int& sub) {
    const int grp = b >> 5, r = b & 31;  // 32 blocks = 8 tiles x 4 sub-tiles
    const int k = grp * 8 + (r & 7);
    sub = r >> 3;
    tile = k < T ? (int)tile_order[k] : -1;
}
inline int sub_tile_grid(int T) { return ((T + 7) / 8) * 32; }
#endif

}  // namespace svgir
