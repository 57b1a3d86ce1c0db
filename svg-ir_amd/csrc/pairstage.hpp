// svg-ir_amd/csrc/pairstage.hpp -- pair-interleaved LDS staging of the forward composite (packed fp32 math, MFMA operands).
//
// gfx950 executes v_pk_{add,mul,fma}_f32 -- two fp32 lanes per aligned VGPR pair -- at the rate of the scalar forms, so
// everything the forward computes per candidate and does not sum over candidates (pixel offsets, the conic form, the exp
// argument reduction, alpha, depth differencing, the bilinear corner weights) runs on TWO consecutive candidates at once:
// element-wise the same operations in the same order, i.e. bit-identical results.  The per-pixel sums over the candidates
// (colour, normal, features, vfeatures) are a contraction [pixels x candidates] . [candidates x channels]: at svgss widths
// they run on the matrix pipe (render_fwd.hip), at rgss widths as packed FMAs on channel pairs (C0,C1) += (r,g) * (w,w).
// Packed operands are aligned register pairs and MFMA B operands are 16-float rows, so the staged data is laid out for
// its consumers: a staged PAIR of candidates (c0 = even slot, c1 = odd slot of the batch) is
//   [0, 2 GEOF)          geometry, interleaved: field g of candidate c at 2 g + c   (a ds_read_b128 yields two fields of both)
//   [2 GEOF + 16 c, ..)  channel block of candidate c: r g b nx ny nz F0..F(S-1), zero padded to 16 (one MFMA B row)
//   [.. + 64 c, ..)      vfeatures of candidate c, corner-major: plane j (corner j) holds channel ch at 16 j + ch
// Geometry fields: 0 x, 1 y, 2 conic.x, 3 conic.z, 4 conic.y, 5 opacity, 6 depth, 7 DA, 8 DB, 9 1/umax,
//                  [svgss:] 10 1/vmax, 11 first instance (bits), 12..15 J0..J3, 16 tile rect (bits), 17 pad.
// The gather itself is unchanged in spirit (stage.hpp): 16-byte loads of the 96-byte record and the vfeatures, one batch
// ahead of their use; here every lane always fetches the same piece of "its" candidate slot, so the four LDS destinations of
// its float4 are loop-invariant per-lane constants (the scatter costs four ds_write_b32 per load and no address math); at
// rgss widths only the four record pieces the forward reads are fetched (64 of the 96 bytes).
#pragma once
#include "common.hpp"
#include "stage.hpp"

namespace svgir {

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int S, int VC, int CHN = (VC > 0 ? 16 : 64)>
struct PairGeom {
    static constexpr bool SV = VC > 0;
    static constexpr int GEOF = SV ? 18 : 10;              // geometry fields per candidate
    static constexpr int NCH = 6 + S;                      // rgb, normal, features
    static constexpr int CHP = 16;                         // channel block per candidate (floats): one MFMA B row of 16 columns
    static constexpr int NPAIR = CHP / 2;
    static constexpr int VCP = 16;                         // channels per corner plane (one MFMA B row)
    static constexpr int VB = SV ? 4 * VCP : 0;            // vfeature block per candidate (floats)
    static constexpr int NVP = VCP / 2;                    // vfeature accumulator pairs
    static constexpr int PF = 2 * GEOF + 2 * CHP + 2 * VB; // floats per staged pair
    static constexpr int CH_OFF = 2 * GEOF, V_OFF = 2 * GEOF + 2 * CHP;
    static constexpr int CH = CHN;                         // candidates staged per batch (forward: 16 / 64)
    static constexpr int QN = 2 * CH;
    // gather geometry: record pieces (16 bytes) per candidate and the lane -> (candidate, piece) split
    static constexpr int RSLOT = SV ? 8 : 4;               // lanes per candidate in a record load (svgss: 6 of 8 used)
    static constexpr int RCPL = 64 / RSLOT;                // candidates per record load
    static constexpr int KR = CH / RCPL;                   // record loads per lane per batch
    static constexpr int VSLOT = 16;                       // lanes per candidate in a vfeature load (VC <= 16)
    static constexpr int VCPL = 4;
    static constexpr int KVF = SV ? CH / VCPL : 0;         // vfeature loads per lane per batch
    static constexpr bool EMB = rec_embeds_features(S, 4 * VC);   // features inside the record
    static constexpr int KF = EMB ? 0 : (CH * S + 63) / 64; // feature floats per lane per batch (separate feature rows)
    static constexpr int KE = EMB ? (CH + 63) / 64 : 0;    // embedded features: record piece 2, lane = candidate, per batch
    // blend-weight panels (the MFMA A operands of the forward accumulation): 4 rows (candidates of a group) + svgss 16 rows
    // (candidate x corner); at least 16 rows, the panel doubles as the 16 x 64 transposition tile of the accumulators
    static constexpr int PS = 80;                          // panel row stride (floats): rows of a 4-row A operand hit disjoint banks
    static constexpr int PROWS = SV ? 20 : 0;              // (packed-FMA accumulators live in registers: no panel)
    static constexpr int WROWS = 4;                        // blend-weight sums are parked per 16-lane row (render_fwd.hip, step 4)
    static constexpr size_t off_q = (size_t)(CH / 2) * PF * 4, off_w = off_q + (size_t)QN * 8, off_p = off_w + (size_t)2 * WROWS * CH * 4;
    static constexpr size_t lds_bytes() { return off_p + (size_t)PROWS * PS * 4; }
    static_assert(PF % 4 == 0 && (2 * GEOF) % 4 == 0, "float4-aligned blocks");
    static_assert(NCH <= 16 && VC <= 16, "one 16-column MFMA tile per channel group");
    static_assert(SEG % CH == 0, "segment boundaries must fall on staging-batch boundaries");
};

#if defined(__HIPCC__)
// element-wise twins of stage.hpp's pair_power / exp_nonpos (same operations, same order => same bits per element)
__device__ __forceinline__ f32x2 pair_power2(f32x2 a, f32x2 b, f32x2 c, f32x2 dx, f32x2 dy) {
    f32x2 s, m;
    {
#pragma clang fp contract(off)
        s = a * dx * dx + c * dy * dy;
        m = b * dx * dy;
    }
    const f32x2 mh = {-0.5f, -0.5f};
    return __builtin_elementwise_fma(mh, s, -m);
}
__device__ __forceinline__ f32x2 exp_nonpos2(f32x2 x) {
    const f32x2 HI = {1.44269502162933349609375f, 1.44269502162933349609375f};
    const f32x2 LO = {1.92596299112661746e-8f, 1.92596299112661746e-8f};
    const f32x2 LN2 = {0.693147182464599609375f, 0.693147182464599609375f};
    f32x2 t;
    {
#pragma clang fp contract(off)
        t = x * HI;
    }
    f32x2 r = __builtin_elementwise_fma(x, HI, -t);
    r = __builtin_elementwise_fma(x, LO, r);
    const f32x2 e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
    f32x2 rl;
    {
#pragma clang fp contract(off)
        rl = r * LN2;
    }
    return __builtin_elementwise_fma(e, rl, e);
}

// Where the four floats of record piece p go inside a staged pair (float offset for the EVEN candidate; `odd_step` is what
// the odd candidate adds): geometry fields are interleaved (2 g, step 1), channels sit in per-candidate blocks (step CHP).
// A negative offset = the value is not staged.
template <int S, int VC>
__device__ __forceinline__ void rec_piece_dest(int piece, int j, int& off, int& odd_step) {
    using PG = PairGeom<S, VC>;   // (offsets do not depend on the batch size)
    constexpr int G = 0, C = 1, X = 2;   // geometry field / channel / dropped
    // source float 4*piece + j of the record (common.hpp RecField) -> {kind, index}
    constexpr int kind[24] = {G, G, G, G,  G, G, G, G,  G, G, G, G,  G, C, C, C,  C, C, C, G,  G, G, G, X};
    constexpr int idx[24] = {0, 1, 2, 4,  3, 5, 6, 7,  12, 13, 14, 15,  8, 0, 1, 2,  3, 4, 5, 9,  10, 11, 16, 0};
    const int f = 4 * piece + j;
    int k = X, i = 0;
#pragma unroll
    for (int q = 0; q < 24; q++) if (q == f) { k = kind[q]; i = idx[q]; }
    if (PG::EMB) {   // floats 8..11 and 19 of the record are feature channels 0..3 and 4 (common.hpp rec_feature_slot)
        if (f >= 8 && f < 12) { k = (f - 8 < S) ? C : X; i = 6 + (f - 8); }
        if (f == 19) { k = S > 4 ? C : X; i = 10; }
    }
    if (k == G && i < PG::GEOF) { off = 2 * i; odd_step = 1; }
    else if (k == C) { off = PG::CH_OFF + i; odd_step = PG::CHP; }
    else { off = -1; odd_step = 0; }
}

template <int S, int VC, int CHN = (VC > 0 ? 16 : 64)>
struct PairRegs {
    using PG = PairGeom<S, VC, CHN>;
    f32x4 r[PG::KR];
    f32x4 v[PG::KVF > 0 ? PG::KVF : 1];
    float f[PG::KF > 0 ? PG::KF : 1];
    f32x4 e[PG::KE > 0 ? PG::KE : 1];   // embedded feature piece (record float4 #2) of candidate u * 64 + lane
};

// Per-lane constants of the scatter: byte addresses (relative to the staging buffer) of the four destinations of the
// lane's record piece for its candidate of load 0 (load u adds u * RCPL / 2 pairs), and of its vfeature channel.
template <int S, int VC, int CHN = (VC > 0 ? 16 : 64)>
struct PairMap {
    using PG = PairGeom<S, VC, CHN>;
    int rpiece;       // record piece this lane fetches (0..5), -1: idle lane
    int rdst[4];      // float offsets, -1: dropped
    int vch;          // vfeature channel this lane fetches, -1: idle
    int vdst;         // float offset of corner 0 of that channel (corner j adds j * VCP)
    __device__ __forceinline__ void init(int lane) {
        const int rs = lane % PG::RSLOT, rc = lane / PG::RSLOT;          // piece slot, candidate of load 0
        rpiece = PG::SV ? (rs < 6 ? rs : -1) : (rs < 2 ? rs : rs + 1);    // rgss: pieces 0, 1, 3, 4 (J and the tail are not needed)
        const int base = (rc >> 1) * PG::PF;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int off = -1, step = 0;
            if (rpiece >= 0) rec_piece_dest<S, VC>(rpiece, j, off, step);
            rdst[j] = off < 0 ? -1 : base + off + (rc & 1) * step;
        }
        const int vs = lane % PG::VSLOT, vc = lane / PG::VSLOT;
        vch = (PG::SV && vs < VC) ? vs : -1;
        vdst = (vc >> 1) * PG::PF + PG::V_OFF + (vc & 1) * PG::VB + vs;
    }
};

template <int S, int VC, int CHN, typename GidOf>
__device__ __forceinline__ void pair_stage_load(PairRegs<S, VC, CHN>& r, const PairMap<S, VC, CHN>& mp, int m, GidOf gid_of, int lane,
                                                const float* __restrict__ rec, const float* __restrict__ feat,
                                                const float* __restrict__ vfeat) {
    using PG = PairGeom<S, VC, CHN>;
    const f32x4* rec4 = reinterpret_cast<const f32x4*>(rec);
    const f32x4* vf4 = reinterpret_cast<const f32x4*>(vfeat);
    const int rc = lane / PG::RSLOT, rp = mp.rpiece < 0 ? 0 : mp.rpiece;
#pragma unroll
    for (int u = 0; u < PG::KR; u++) {
        const size_t id = (size_t)gid_of(min(u * PG::RCPL + rc, m - 1));   // (slots beyond the batch re-load its last candidate)
        r.r[u] = rec4[id * 6 + rp];
    }
    if (PG::SV) {
        const int vc = lane / PG::VSLOT, vch = mp.vch < 0 ? 0 : mp.vch;
#pragma unroll
        for (int u = 0; u < PG::KVF; u++) {
            const size_t id = (size_t)gid_of(min(u * PG::VCPL + vc, m - 1));
            r.v[u] = vf4[id * VC + vch];
        }
    }
    if (PG::EMB) {
#pragma unroll
        for (int u = 0; u < PG::KE; u++) {
            const size_t id = (size_t)gid_of(min(u * 64 + lane, m - 1));
            r.e[u] = rec4[id * 6 + 2];
        }
    } else if (S > 0) {
        const int totf = m * S;
#pragma unroll
        for (int u = 0; u < PG::KF; u++) {
            const int k = min(u * 64 + lane, totf - 1);
            const int s = k / S, c = k - s * S;
            r.f[u] = feat[(size_t)gid_of(s) * S + c];
        }
    }
}

template <int S, int VC, int CHN>
__device__ __forceinline__ void pair_stage_store(const PairRegs<S, VC, CHN>& r, const PairMap<S, VC, CHN>& mp, float* __restrict__ sD,
                                                 int m, int lane) {
    using PG = PairGeom<S, VC, CHN>;
    const int rc = lane / PG::RSLOT;
    if (m < CHN) {   // (uniform; the walk's last batch) the rest of the last group of four: opacity 0, i.e. alpha 0 -- the blend loop tests no bounds
        const int s = m + lane;
        if (s < ((m + 3) & ~3)) sD[(s >> 1) * PG::PF + 2 * 5 + (s & 1)] = 0.f;
    }
#pragma unroll
    for (int u = 0; u < PG::KR; u++) {
        if (u * PG::RCPL + rc < m && mp.rpiece >= 0) {
            float* d = sD + u * (PG::RCPL / 2) * PG::PF;
            if (mp.rdst[0] >= 0) d[mp.rdst[0]] = r.r[u].x;
            if (mp.rdst[1] >= 0) d[mp.rdst[1]] = r.r[u].y;
            if (mp.rdst[2] >= 0) d[mp.rdst[2]] = r.r[u].z;
            if (mp.rdst[3] >= 0) d[mp.rdst[3]] = r.r[u].w;
        }
    }
    if (PG::SV) {
        const int vc = lane / PG::VSLOT;
#pragma unroll
        for (int u = 0; u < PG::KVF; u++) {
            if (u * PG::VCPL + vc < m && mp.vch >= 0) {
                float* d = sD + u * (PG::VCPL / 2) * PG::PF + mp.vdst;
                d[0] = r.v[u].x; d[PG::VCP] = r.v[u].y; d[2 * PG::VCP] = r.v[u].z; d[3 * PG::VCP] = r.v[u].w;
            }
        }
    }
    if (PG::EMB) {
#pragma unroll
        for (int u = 0; u < PG::KE; u++) {
            const int s = u * 64 + lane;
            if (s < m) {
                float* d = sD + (s >> 1) * PG::PF + PG::CH_OFF + (s & 1) * PG::CHP + 6;
                d[0] = r.e[u].x;
                if (S > 1) d[1] = r.e[u].y;
                if (S > 2) d[2] = r.e[u].z;
                if (S > 3) d[3] = r.e[u].w;
            }
        }
    } else if (S > 0) {
        const int totf = m * S;
#pragma unroll
        for (int u = 0; u < PG::KF; u++) {
            const int k = u * 64 + lane;
            if (k < totf) {
                const int s = k / S, c = k - s * S;
                sD[(s >> 1) * PG::PF + PG::CH_OFF + (s & 1) * PG::CHP + 6 + c] = r.f[u];
            }
        }
    }
}
#endif

}  // namespace svgir
