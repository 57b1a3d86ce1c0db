// svg-ir_amd/csrc/render_bwd_plain.hip -- backward alpha compositing at the widths WITHOUT vfeatures (rgss, and svgss
// with VS = 0): replaces the backward renderCUDA of rgss backward.cu:431-757 (svgss backward.cu:529-934 with VS = 0).
//
// Same decomposition as render_bwd.hip (one wave64 per 64-candidate depth segment of an 8x8 sub-tile, started from the
// forward's dumped blend state), but built around how a gfx950 SIMD issues instructions (scripts/probes/valu_rate_probe.hip,
// measured): ONE wave issues one VALU instruction per ~8 cycles whether or not the instructions depend on each other, and a
// SIMD only reaches its 2 cycles per wave64 instruction with >= 4-6 resident waves.  So instruction-level parallelism inside
// a wave buys nothing, registers spent on it cost resident waves, and every instruction counts.  Hence:
//   * one candidate at a time (no lock-step groups), <= 80 VGPRs and 5.1 KB of LDS per wave -> 6 waves per SIMD (was 3);
//   * the per-candidate attributes are wave-uniform: they are fetched with SCALAR loads (constant address space, one
//     candidate ahead) into SGPRs and enter the per-pixel math as scalar operands -- no LDS staging buffer, no broadcast
//     ds_reads, no VGPRs for uniform data;
//   * all per-Gaussian sums over the 64 pixels run on the matrix pipe.  The blend weight w and ONE more per-(pixel,
//     candidate) scalar v = G dL_dalpha go to a 16-row LDS panel (8 candidates x {w, v}); rows w x G[pixel][colour3 normal3
//     depth feature S] give the channel gradients, rows v x Mom[pixel][1 px py px^2 px py py^2] give six pixel moments from
//     which the six geometric gradients (mean2D.xy, conic.xyz, opacity) follow per candidate:
//         sum_p v dx = X M0 - M1,  sum_p v dx^2 = X (X M0 - 2 M1) + M3,  ...   (X, Y, px, py relative to the sub-tile centre)
//     with dL_ddist = v (-0.5 opacity) -- v_mfma_f32_16x16x4_f32, exact fp32 products;
//   * the un-weighted depth-differencing term (quirk Q5) needs sum_p [pixel blends] (-gD): one DPP wave reduction per candidate;
//   * results: float atomics into ONE packed gradient row per Gaussian (common.hpp GradRowGeom), unpacked by geom_bwd.hip.
#include <algorithm>

#include "common.hpp"
#include "stage.hpp"
#include "dev_trace.hpp"

namespace svgir {

namespace {

typedef const __attribute__((address_space(4))) float cfloat;      // constant address space: uniform loads become s_load
typedef const __attribute__((address_space(4))) uint32_t cuint;

#ifndef BWDP_GRID_MULT
#define BWDP_GRID_MULT 4   // workgroups launched per resident wave slot
#endif
#ifndef BWDP_WPE
#define BWDP_WPE 4
#endif

template <int S>
struct PlainGeom {
    static constexpr int SB = 8;                 // candidates per panel (8 w rows + 8 v rows)
    static constexpr int NC0 = 7 + S;            // colour3, normal3, depth, feature S
    static constexpr int GROW = NC0 + 1;
    static constexpr int PS = 68;                // panel row stride (floats)
    static constexpr int PROWS = 24;             // panel rows: 8 blend weights w | 8 v | 8 u (Q5 term)
    static constexpr size_t off_m = (size_t)PROWS * PS * 4;            // moments + Q5 sums [SB][8]
    static constexpr size_t off_c = off_m + (size_t)SB * 8 * 4;        // per-candidate constants of the block [3][SB] float4 (LDS-DMA target)
    static constexpr size_t off_q = off_c + (size_t)3 * SB * 16;       // the segment's {gid, slot} entries, deepest first
    static constexpr size_t lds_bytes = off_q + (size_t)SEG * 8;
    static_assert(NC0 <= 16, "one 16-wide MFMA column tile");
    static_assert((size_t)64 * GROW * 4 <= off_m, "the G transposition tile aliases the panel");
};

template <int S, bool SVGSS>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(BWDP_WPE, BWDP_WPE)))
render_bwd_plain_kernel(const RenderBwdArgs a) {
    using PG = PlainGeom<S>;
    constexpr int SB = PG::SB, NC0 = PG::NC0, GROW = PG::GROW, PS = PG::PS;
    constexpr int SS = S > 0 ? S : 1;
    constexpr int P4 = (NC0 + 3) / 4 * 4, GEO = P4, RS = (GEO + 6 + 3) / 4 * 4;   // common.hpp GradRowGeom (VS = 0)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sP = reinterpret_cast<float*>(smem);                    // [24][PS] rows 0..7: blend weights, 8..15: v, 16..23: u
    float* sG = sP;                                                // [64][GROW] upstream gradients (transposition only)
    float* sM = reinterpret_cast<float*>(smem + PG::off_m);        // [SB][8] six moments, Q5 sum, pad
    float4* sC = reinterpret_cast<float4*>(smem + PG::off_c);      // [3][SB] record float4 #0, #1, #3 of the block's candidates
    uint2* sQ = reinterpret_cast<uint2*>(smem + PG::off_q);        // [SEG] {gid, slot}, deepest first

    const int lane = threadIdx.x;
    const bool surface = cfg_flag(a.cfg, 0), normalize_depth = cfg_flag(a.cfg, 1);
    const bool sp = surface && cfg_flag(a.cfg, 2);
    const bool bgeom = SVGSS ? true : (a.backward_geometry != 0);
    const size_t N_ = (size_t)a.W * a.H;
    const float ddelx_dx = 0.5f * a.W, ddely_dy = 0.5f * a.H;
    const int colB = lane & 15, grpB = lane >> 4;

    // B operand of the moment contraction, lane l: Mom[pixel 16 (l >> 4) + kk][column l & 15] = X(mx(kk & 7)) Y(my(kk >> 3)),
    // (mx, my) = pixel position inside the 8x8 sub-tile minus 3.5, columns 1, px, py, px^2, px py, py^2 (others 0):
    // X(m) = cx0 + m (cx1 + m cx2) with one-hot (cx0, cx1, cx2) by column -- evaluated between the MFMAs, where the wave
    // waits for the matrix pipe anyway -- and Y for the lane's two pixel rows in momy[2]
    float cx0, cx1, cx2, momy[2];
    {
        const int ex = colB == 1 || colB == 4 ? 1 : colB == 3 ? 2 : 0;
        const int ey = colB == 2 || colB == 4 ? 1 : colB == 5 ? 2 : 0;
        cx0 = (colB < 6 && ex == 0) ? 1.f : 0.f; cx1 = ex == 1 ? 1.f : 0.f; cx2 = ex == 2 ? 1.f : 0.f;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const float my = (float)(2 * grpB + h) - 3.5f;
            momy[h] = ey == 0 ? 1.f : ey == 1 ? my : my * my;
        }
    }

    typedef const __attribute__((address_space(4))) uint32_t cu32;
    // the segment count and this wave's first descriptor are independent loads (the list position of a work id does not
    // depend on the count: common.hpp seg_item_of)
    cu32* dsc0 = (cu32*)(uintptr_t)(a.seg_desc + min(seg_item_of(blockIdx.x), (uint32_t)a.seg_cap));
    uint32_t d_sm = dsc0[0], d_r0 = dsc0[1], d_len = dsc0[2], d_count = dsc0[3], d_ndump = dsc0[4], d_sb = dsc0[6];
    // The caller's gradient tensors start from zero and are first touched by the kernels BEHIND this one: every wave of the grid clears
    // its share here -- stores the wave never waits for, in a kernel that is bound by latency, not by bandwidth -- instead of a memset
    // on a side stream that has to be forked from and joined with the caller's stream.
    if (a.clear) {
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        for (size_t i = (size_t)blockIdx.x * 64 + threadIdx.x; i < a.clear_n16; i += (size_t)gridDim.x * 64) a.clear[i] = z;
    }
    const uint32_t nlive = min(a.seg_count[0], (uint32_t)a.seg_cap);
    const uint32_t nwork = seg_work_ids(nlive);
    if (blockIdx.x >= nwork) return;
    DEV_TRACE_DECL();
    [[maybe_unused]] unsigned dev_items = 0, dev_cands = 0;
    for (uint32_t w = blockIdx.x; w < nwork; w += gridDim.x) {
    // longest-first list, dealt to the XCDs in blocks of consecutive items
    const uint32_t item = seg_item_of(w);
    if (item >= nlive) continue;
    if (w != blockIdx.x) {
        cu32* dsc = (cu32*)(uintptr_t)(a.seg_desc + item);
        d_sm = dsc[0]; d_r0 = dsc[1]; d_len = dsc[2]; d_count = dsc[3]; d_ndump = dsc[4]; d_sb = dsc[6];
    }
    wave_lds_sync();   // the previous segment's LDS traffic is complete before its buffers are reused
    const uint32_t sm = d_sm, r0 = d_r0, tlen = d_len;
    const int count = (int)d_count, ndump = (int)d_ndump;
    const int sid = (int)(sm >> SEG_K_BITS), kseg = (int)(sm & ((1u << SEG_K_BITS) - 1u));
    const int tile = sid >> 2, sub = sid & 3;
    const int seg_lo = kseg * SEG, seg_hi = min(count, seg_lo + SEG);
    if (seg_hi <= seg_lo) continue;
    const int tx = tile % a.gx, ty = tile / a.gx;
    const int px = tx * TILE + (sub & 1) * 8 + (lane & 7);
    const int py = ty * TILE + (sub >> 1) * 8 + (lane >> 3);
    const bool inside = px < a.W && py < a.H;
    const float pxf = (float)px, pyf = (float)py;
    const float cxs = (float)(tx * TILE + (sub & 1) * 8) + 3.5f, cys = (float)(ty * TILE + (sub >> 1) * 8) + 3.5f;   // sub-tile centre
    const uint2* __restrict__ sub_in = a.sub_list + (size_t)4 * r0 + (size_t)sub * tlen;
    const size_t pid = inside ? (size_t)a.W * py + px : 0;
    const int nent = seg_hi - seg_lo;
    cuint* list_c = (cuint*)(uintptr_t)(sub_in + (seg_hi - 1));   // entry i of the walk = {list_c[-2 i], list_c[-2 i + 1]}

    // per-candidate attributes (wave-uniform -> SGPRs): list entries are fetched two candidates ahead, records one ahead
    struct Cand { float X, Y, cxx, cxy, cyy, op, dep, DA, DB, cr, cg, cb, nx, ny, nz, f[SS]; uint32_t slot; };
    struct Entry { uint32_t gid, slot; };
    typedef const __attribute__((address_space(4))) char cchar;
    cchar* rec_b = (cchar*)(uintptr_t)a.rec;
    cchar* feat_b = (cchar*)(uintptr_t)a.features;
    auto fetch_entry = [&](int i) -> Entry {
        const int ci = min(i, nent - 1);   // (uniform) clamped: slots beyond the list replay a real record with weight 0
        Entry e;
        e.gid = list_c[-2 * ci];
        const uint32_t sl = list_c[-2 * ci + 1];
        e.slot = i < nent ? sl : 0xffffffffu;   // (slot 2^32-1: never blends)
        return e;
    };
    auto fetch_rec = [&](const Entry& e) -> Cand {
        // 32-bit byte offsets: P * 96 B < 4 GiB (api.hip validate)
        cfloat* r = (cfloat*)(rec_b + (uint32_t)(e.gid * (uint32_t)(REC * 4)));
        Cand c;
        c.X = r[R_X]; c.Y = r[R_Y]; c.cxx = r[R_CX]; c.cxy = r[R_CY]; c.cyy = r[R_CZ]; c.op = r[R_OP]; c.dep = r[R_DEPTH]; c.DA = r[R_DA];
        c.DB = r[R_DB]; c.cr = r[R_R]; c.cg = r[R_G]; c.cb = r[R_B]; c.nx = r[R_NX]; c.ny = r[R_NY]; c.nz = r[R_NZ];
        if (rec_embeds_features(S, 0)) {   // the feature row rides in the record (preprocess; common.hpp rec_feature_slot)
#pragma unroll
            for (int ch = 0; ch < SS; ch++) c.f[ch] = ch < S ? r[rec_feature_slot(ch)] : 0.f;
        } else {
            cfloat* f = (cfloat*)(feat_b + (uint32_t)(e.gid * (uint32_t)(S * 4)));
            if (S >= 4) {   // (scalar loads only need dword alignment: one x4 + singles instead of S singles)
                typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
                const f32x4u f4 = *reinterpret_cast<const __attribute__((address_space(4))) f32x4u*>(f);
                c.f[0] = f4.x; c.f[1 % SS] = f4.y; c.f[2 % SS] = f4.z; c.f[3 % SS] = f4.w;
    #pragma unroll
                for (int ch = 4; ch < S; ch++) c.f[ch] = f[ch];
            } else {
    #pragma unroll
                for (int ch = 0; ch < SS; ch++) c.f[ch] = ch < S ? f[ch] : 0.f;
            }
        }
        c.slot = e.slot;
        return c;
    };
    // speculative: the replay usually starts at entry 0 -- its record is fetched while the per-pixel loads are in flight
    const Cand spec_cur = fetch_rec(fetch_entry(0));
    const Entry spec_en1 = fetch_entry(1);
    asm volatile("" :: "s"(spec_cur.X), "s"(spec_en1.gid));

    const float T_final = inside ? a.final_T[pid] : 0.f;
    const float D_final = (inside && normalize_depth) ? a.final_D[pid] : 0.f;
    const uint32_t last_contributor = inside ? (uint32_t)a.n_contrib[pid] : 0u;
    float gC[3], gN[3], gF[SS], gD = 0.f, gO = 0.f;
#pragma unroll
    for (int i = 0; i < 3; i++) { gC[i] = (inside && a.g_color) ? a.g_color[i * N_ + pid] : 0.f; gN[i] = (inside && a.g_normal) ? a.g_normal[i * N_ + pid] : 0.f; }
#pragma unroll
    for (int i = 0; i < SS; i++) gF[i] = (inside && i < S && a.g_feature) ? a.g_feature[i * N_ + pid] : 0.f;
    if (inside) { gD = a.g_depth ? a.g_depth[pid] : 0.f; gO = a.g_opacity ? a.g_opacity[pid] : 0.f; }
    const float bgdot = a.bg[0] * gC[0] + a.bg[1] * gC[1] + a.bg[2] * gC[2];
    const float omt = 1.f - T_final;
    const float gDn = normalize_depth ? gD / omt : gD;  // depth gradient seen by the blended depth
    // d(depth normalisation)/d alpha of the reference, gD*D_final/(1-Tf)^2 * -Tf/(1-alpha)/T_new, equals kdn / T_old
    const float kdn = normalize_depth ? -gD * D_final * T_final / (omt * omt) : 0.f;
    const float gO_kbg = gO - (bgdot + (normalize_depth ? 0.f : 10.f * gD));   // opacity minus background (+ un-normalised depth) term
    const float q5g = sp ? -gD : 0.f;   // Q5: un-weighted depth-differencing term

    // deepest contributor of the wave
    uint32_t wmax = last_contributor;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, d));
    if (wmax == 0) continue;

    // Replay state: the per-channel recurrences of the reference collapse into ONE scalar recurrence on
    // A = sum_ch accum_ch g_ch with s = sum_ch value_ch g_ch (render_bwd.hip):  A <- last_alpha s_last + (1 - last_alpha) A ;
    // dL_dalpha += s - A.
    float T = T_final;
    float last_alpha = 0.f;
    float A_acc = 0.f, s_last = 0.f;
    if (kseg < ndump) {
        // Not the deepest live segment: start from the forward state dumped at this segment's far end.  With
        // last_alpha = 0 the recurrence takes accum = blend of everything behind = (final - prefix) / T_end.
        constexpr int NST = 8 + S;
        const uint32_t sbase = d_sb;   // state slot of (sub-tile, 0)
        const float* e = a.seg_state + ((size_t)(sbase + kseg) * NST) * 64 + lane;
        const float* f = a.seg_state + ((size_t)(sbase + ndump) * NST) * 64 + lane;   // final state
        T = e[0];
        float dot = (f[7 * 64] - e[7 * 64]) * gDn;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            dot += (f[(1 + i) * 64] - e[(1 + i) * 64]) * gC[i];
            dot += (f[(4 + i) * 64] - e[(4 + i) * 64]) * gN[i];   // zero unless `surface` (the forward leaves N at 0)
        }
        if (bgeom) {
#pragma unroll
            for (int i = 0; i < S; i++) dot += (f[(8 + i) * 64] - e[(8 + i) * 64]) * gF[i];
        }
        A_acc = dot * __builtin_amdgcn_rcpf(T);
    }

    // G matrix of this sub-tile, row = pixel (lane), columns = [colour3 | normal3 x10 (Q4) | depth | feature S]; phase B needs
    // it as the MFMA B operand (lane l: G[pixel 16 (l >> 4) + kk][channel l & 15], kk = 0..15): transposed once through LDS
    {
        float* g = sG + lane * GROW;
        g[0] = gC[0]; g[1] = gC[1]; g[2] = gC[2];
        g[3] = surface ? gN[0] * 10.f : 0.f; g[4] = surface ? gN[1] * 10.f : 0.f; g[5] = surface ? gN[2] * 10.f : 0.f;
        g[6] = gDn;
#pragma unroll
        for (int i = 0; i < S; i++) g[7 + i] = gF[i];
    }
    float Bp[16];
    wave_lds_sync();
    {
        const float* gB = sG + (16 * grpB) * GROW + (colB < NC0 ? colB : 0);
#pragma unroll
        for (int kk = 0; kk < 16; kk++) Bp[kk] = colB < NC0 ? gB[kk * GROW] : 0.f;
    }
    wave_lds_sync();   // sG aliases the panel

    // The segment's list entries, deepest first (the replay walks back to front): LDS copy for the per-lane consumers
    // (atomics of phase B); the per-candidate math reads them with scalar loads.
    int nskip = 0;   // entries that lie behind every pixel of this wave (a prefix: slots descend)
    {
        uint2 e = make_uint2(0u, 0u);
        if (lane < nent) e = sub_in[seg_hi - 1 - lane];
        sQ[lane] = e;
        nskip = __popcll(__ballot(lane < nent && e.y >= wmax));
    }
    // flags folded into the per-pixel factors: the replay itself is branch-free
    const float gNe0 = surface ? gN[0] : 0.f, gNe1 = surface ? gN[1] : 0.f, gNe2 = surface ? gN[2] : 0.f;
    float gFe[SS];
#pragma unroll
    for (int i = 0; i < SS; i++) gFe[i] = bgeom ? gF[i] : 0.f;
    const float spf = sp ? 1.f : 0.f;
    const float gOT = gO_kbg * T_final;

    const int cstart = (nskip / SB) * SB;
    Cand cur = spec_cur;
    Entry en1 = spec_en1;
    if (cstart != 0) {   // (uniform) the deepest entries lie behind every pixel of the wave: start further in
        cur = fetch_rec(fetch_entry(cstart));
        en1 = fetch_entry(cstart + 1);
    }
    // Per-candidate constants of a block for its geometric epilogue (lane = (chunk, candidate)): copied global -> LDS by the
    // DMA path (no VGPRs, nothing waits) one block ahead -- issued here for the first block, then at the end of every block
    auto prefetch_consts = [&](int cb) {
        int lC = lane;
        asm volatile("" : "+v"(lC));
        if (lC < 3 * SB) {
            const int cq = lC & 7, chunk = lC >> 3;
            const uint32_t gq = sQ[min(cb + cq, SEG - 1)].x;
            const float* src = a.rec + (uint32_t)(gq * (uint32_t)REC + (uint32_t)(chunk == 2 ? 12 : 4 * chunk));
            __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(smem + PG::off_c), 16, 0, 0);
        }
    };
    wave_lds_sync();   // sQ visible
    prefetch_consts(cstart);
    DEV_TRACE_MARK(0);   // segment setup
    dev_items++; dev_cands += (unsigned)nent;
    for (int c0 = cstart; c0 < nent; c0 += SB) {
        uint32_t live = 0;  // bit k set: candidate c0+k has at least one blending pixel (wave-uniform)
        // ---------------- phase A: lane = pixel, one candidate at a time, attributes in SGPRs ----------------
        // (scalar loads and LDS stores share one completion counter: the panel stores of candidate k are issued right AFTER the
        // wait for candidate k+1's attributes, so that wait never includes a fresh LDS store)
        float hw = 0.f, hv = 0.f, hu = 0.f;
#pragma unroll
        for (int k = 0; k < SB; k++) {
            const Cand nxt = fetch_rec(en1);
            en1 = fetch_entry(c0 + k + 2);
            const float dx = cur.X - pxf, dy = cur.Y - pyf;
            if (k > 0) {
                sP[(k - 1) * PS + lane] = hw; sP[(SB + k - 1) * PS + lane] = hv; sP[(2 * SB + k - 1) * PS + lane] = hu;
            }
            const float pw = pair_power(cur.cxx, cur.cxy, cur.cyy, dx, dy);
            const float Gs = exp_nonpos(pw);
            const float al = fminf(0.99f, cur.op * Gs);
            const bool pre = cur.slot < last_contributor && pw <= 0.0f && al >= (1.0f / 255.0f);
            const float ioma = __builtin_amdgcn_rcpf(1.f - al);
            float sd = cur.cr * gC[0] + cur.cg * gC[1] + cur.cb * gC[2];
            sd += cur.nx * gNe0 + cur.ny * gNe1 + cur.nz * gNe2;
            const float d_cur = cur.dep - spf * (dx * cur.DA + dy * cur.DB);   // depth differencing (common.hpp R_DA / R_DB)
            sd += d_cur * gDn;
#pragma unroll
            for (int ch = 0; ch < S; ch++) sd += cur.f[ch] * gFe[ch];
            // the sequential part: T <- T / (1 - alpha) and the scalar replay recurrence (backward.cu:700-850)
            const float inv_Told = __builtin_amdgcn_rcpf(T);
            const float Tn = T * ioma;
            const float An = last_alpha * s_last + (1.f - last_alpha) * A_acc;
            float dL_dalpha = kdn * inv_Told + (sd - An);
            dL_dalpha *= Tn;
            dL_dalpha += gOT * ioma;
            T = pre ? Tn : T;
            A_acc = pre ? An : A_acc;
            s_last = pre ? sd : s_last;
            last_alpha = pre ? al : last_alpha;
            hw = pre ? al * Tn : 0.f;              // blend weight
            hv = pre ? Gs * dL_dalpha : 0.f;       // v
            hu = pre ? q5g : 0.f;                  // u: its pixel sum is the Q5 term (0 unless per-pixel depth is on)
            live |= (__builtin_amdgcn_ballot_w64(pre) != 0ull ? 1u : 0u) << k;
            cur = nxt;
        }
        sP[(SB - 1) * PS + lane] = hw; sP[(2 * SB - 1) * PS + lane] = hv; sP[(3 * SB - 1) * PS + lane] = hu;
        DEV_TRACE_MARK(2);   // phase A
        if (live != 0) {     // uniform
        wave_lds_sync();          // panel rows visible
        // ---------------- phase B: panel x [G | Mom] on the matrix pipe ----------------
        // D layout: lane l, register r -> row 4 (l >> 4) + r, column l & 15
        // (lane-derived addresses are re-derived here every block: hoisted out of the loop they would hold ~20 VGPRs across phase A)
        int lB = lane;
        asm volatile("" : "+v"(lB));
        const int colB = lB & 15, grpB = lB >> 4;
        f32x4 accM = {0.f, 0.f, 0.f, 0.f};
        {   // rows (v | u) x Mom: lanes 0..31, columns 0..5 = the six pixel moments of v; lanes 32..63, column 0 = sum of u
            const float4* ap = reinterpret_cast<const float4*>(sP + (SB + colB) * PS + 16 * grpB);
            const float4 a0 = ap[0], a1 = ap[1], a2 = ap[2], a3 = ap[3];
            const float av[16] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w,
                                  a2.x, a2.y, a2.z, a2.w, a3.x, a3.y, a3.z, a3.w};
            float c0l = cx0;
            asm volatile("" : "+v"(c0l));   // (opaque: the 16 operand values are not hoisted out of the block loop into 16 VGPRs)
#pragma unroll
            for (int kk = 0; kk < 16; kk++) {
                const float mx = (float)(kk & 7) - 3.5f;
                accM = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], (c0l + mx * (cx1 + mx * cx2)) * momy[kk >> 3], accM, 0, 0, 0);
            }
        }
        if (grpB < 2) {
            if (colB < 6) {
#pragma unroll
                for (int rr = 0; rr < 4; rr++) sM[(4 * grpB + rr) * 8 + colB] = accM[rr];
            }
        } else if (colB == 0) {
#pragma unroll
            for (int rr = 0; rr < 4; rr++) sM[(4 * (grpB - 2) + rr) * 8 + 6] = accM[rr];
        }
        f32x4 accP = {0.f, 0.f, 0.f, 0.f};
        {   // rows (w | v) x G: lanes 0..31 = the channel gradients of the 8 candidates
            const float4* ap = reinterpret_cast<const float4*>(sP + colB * PS + 16 * grpB);
            const float4 a0 = ap[0], a1 = ap[1], a2 = ap[2], a3 = ap[3];
            const float av[16] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w,
                                  a2.x, a2.y, a2.z, a2.w, a3.x, a3.y, a3.z, a3.w};
#pragma unroll
            for (int kk = 0; kk < 16; kk++) accP = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], Bp[kk], accP, 0, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): the block's constants have landed in LDS (issued a whole block ago)
        wave_lds_sync();
        DEV_TRACE_MARK(3);   // phase B (MFMA)
        // lane = (value j = lane >> 3, candidate c = lane & 7): the six geometric gradients from the moments
        {
            const int cq = lB & 7, jq = lB >> 3;
            if (jq < 6 && ((live >> cq) & 1u)) {
                const float4 m03 = *reinterpret_cast<const float4*>(sM + cq * 8);
                const float4 m47 = *reinterpret_cast<const float4*>(sM + cq * 8 + 4);
                const float4 Aq = sC[cq], Bq = sC[SB + cq];
                const float DBq = sC[2 * SB + cq].x;
                const float Xc = Aq.x - cxs, Yc = Aq.y - cys;       // mean relative to the sub-tile centre
                const float M0 = m03.x, M1 = m03.y, M2 = m03.z, M3 = m03.w, M4 = m47.x, M5 = m47.y, Q = m47.z;
                const float Sx = Xc * M0 - M1, Sy = Yc * M0 - M2;  // sum v dx, sum v dy
                const float hf = Bq.y * -0.5f;                      // dL_ddist = v * hf
                float ge;
                if (jq == 0) ge = hf * 2.f * (Aq.z * Sx + Aq.w * Sy) * ddelx_dx + Q * Bq.w;
                else if (jq == 1) ge = hf * 2.f * (Bq.x * Sy + Aq.w * Sx) * ddely_dy + Q * DBq;
                else if (jq == 2) ge = hf * (Xc * (Sx - M1) + M3);              // sum v dx^2
                else if (jq == 3) ge = hf * ((Xc * Sy - Yc * M1) + M4);          // sum v dx dy
                else if (jq == 4) ge = hf * (Yc * (Sy - M2) + M5);              // sum v dy^2
                else ge = M0;
                if (ge != 0.f) atomic_add_f32(a.grad_rows + (uint32_t)(sQ[c0 + cq].x * (uint32_t)RS + (uint32_t)(GEO + jq)), ge);
            }
        }
        if (grpB < 2) {
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int cB = 4 * grpB + rr;
                const bool mine = ((live >> cB) & 1u) && colB < NC0 && accP[rr] != 0.f;
                if (mine) atomic_add_f32(a.grad_rows + (uint32_t)(sQ[c0 + cB].x * (uint32_t)RS + (uint32_t)colB), accP[rr]);
            }
        }
        wave_lds_sync();   // moments / constants / panel consumed before they are overwritten
        }
        if (c0 + SB < nent) prefetch_consts(c0 + SB);
        DEV_TRACE_MARK(1);   // geometric epilogue
    }
    }   // loop over live segments
    DEV_TRACE_END(1, dev_items, dev_cands, blockIdx.x);
}

template <int S, bool SVGSS>
void launch(const RenderBwdArgs& a, hipStream_t s) {
    using PG = PlainGeom<S>;
    // one wave per live segment up to a few waves per resident slot; waves beyond the (device-side) count exit at once
    const int grid = std::max(8, std::min(a.seg_cap, BWDP_GRID_MULT * 256 * 4 * BWDP_WPE) & ~7);   // a multiple of 8: work id & 7 = XCD in every round
    hipLaunchKernelGGL((render_bwd_plain_kernel<S, SVGSS>), dim3(grid), dim3(64), PG::lds_bytes, s, a);
}

}  // namespace

int launch_render_bwd_plain(const RenderBwdArgs& a, bool svgss, hipStream_t s) {
    if (a.VS != 0) return -1;
#define CASE(SV, SG) if (a.S == SV && svgss == SG) { launch<SV, SG>(a, s); return 0; }
    CASE(0, true) CASE(5, true) CASE(0, false) CASE(5, false) CASE(3, false) CASE(1, false)
#undef CASE
    return -1;
}

#if defined(SVGIR_DEV)
extern "C" int svgir_dev_trace_read_bwd_plain(unsigned long long* out, int cap_records) {
    unsigned int n[2];
    if (hipMemcpyFromSymbol(n, HIP_SYMBOL(svgir::g_dev_trace_n), sizeof(n)) != hipSuccess) return -1;
    int cnt = (int)std::min<unsigned>(n[1], (unsigned)std::min(cap_records, svgir::DEV_TRACE_CAP));
    if (cnt > 0 && hipMemcpyFromSymbol(out, HIP_SYMBOL(svgir::g_dev_trace), (size_t)cnt * svgir::DEV_TRACE_WORDS * 8,
                                       (size_t)svgir::DEV_TRACE_CAP * svgir::DEV_TRACE_WORDS * 8) != hipSuccess) return -1;
    n[1] = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(svgir::g_dev_trace_n), n, sizeof(n));
    return cnt;
}
#endif

}  // namespace svgir
