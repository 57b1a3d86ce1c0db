// svg-ir_amd/csrc/render_bwd.hip -- backward alpha compositing.
//
// Replaces the backward renderCUDA (svgss backward.cu:529-934, rgss backward.cu:431-757): back-to-front replay
// of each pixel's blended splats (T <- T / (1 - alpha) starting from final_T), gradients of every blended quantity
// w.r.t. the per-Gaussian colour / feature / vfeature (x corner weight) / normal (x10, Q4) / depth, and through
// alpha to conic, mean2D (incl. the un-weighted depth-differencing term, Q5) and opacity.
//
// CDNA4 mapping
//   * one wave64 per depth segment of an 8x8-pixel sub-tile (one wave per workgroup, no barriers across waves): SEG
//     consecutive entries of the compact candidate list {Gaussian id, slot} that the cull kernel wrote for the
//     sub-tile (nothing is culled twice), walked in reverse.  The deepest live segment starts from final_T like the
//     reference; every other one starts from the forward state dumped at its far end (T there, and the blend of
//     everything behind = (final accumulators - prefix accumulators) / T), which turns the reference's strictly
//     sequential per-pixel replay into many short, evenly sized work items.  The waves stride over the compact list
//     of live segments the forward appended (seg_list / seg_count, read on the device), so no workgroup is
//     launched for a dead segment;
//   * candidates are staged CHB at a time into LDS like in the forward (stage.hpp), the gathers of the next batch in
//     flight while the current one is replayed;
//   * per-pixel replay state (running accumulators, last values, upstream gradients) lives in VGPRs thanks to
//     compile-time channel counts (the reference keeps ~330 floats per thread in scratch, backward.cu:617-635);
//   * the replay handles one candidate at a time (a wave issues one instruction per ~8 cycles whether or not instructions depend on each
//     other, so lock-step groups buy nothing and cost registers); only the (T, A) recurrence is inherently sequential;
//   * gradient accumulation.  The reference issues 13 + S + VS global float atomics per (pixel, splat) pair
//     (18 / 69 / 84).  Here every per-Gaussian gradient is first reduced over the 64 pixels of the wave:
//       - all gradients except the 6 geometric ones have the form  dL/dq[g][ch] = sum_pixels a_v[pixel] G[pixel][ch]
//         with a per-sub-tile-constant matrix G (upstream image gradients) and only 1 (+4 for the bilinear corners)
//         per-pair scalars a_v: phase A (lane = pixel) writes the scalars to an LDS panel [row][64 pixels], phase B
//         runs the contraction panel[rows][64] x G[64][channels] on the matrix pipe (v_mfma_f32_16x16x4_f32, exact
//         fp32; the one genuinely dense contraction of the path -- the blending itself stays scalar);
//       - the 6 geometric gradients (mean2D.xy, conic.xyz, opacity) are summed over the pixels in registers: a transposed DPP
//         reduction tree + two v_permlane*_swap steps (no LDS panel: the wave's LDS footprint decides its occupancy, BwdGeom);
//     the per-(wave, splat) results go to memory as plain stores of the complete gradient row of the (instance, sub-tile) pair,
//     summed per Gaussian by grad_reduce.hip (no atomics, bit-reproducible).  (Widths without vfeatures: render_bwd_plain.hip.)
#include <algorithm>

#include "common.hpp"
#include "stage.hpp"
#include "dev_trace.hpp"

namespace svgir {

namespace {

template <int S, int VC>
struct BwdGeom {
    using SG = StageGeom<S, VC>;
#ifndef BWD_KB_V
#define BWD_KB_V 1
#endif
    static_assert(VC > 0, "widths without vfeatures are handled by render_bwd_plain.hip");
    // One candidate per replay group: a wave issues one instruction per ~8 cycles whether or not instructions depend on each
    // other (scripts/probes/valu_rate_probe.hip), so lock-step groups buy no speed and cost registers and LDS.
    static constexpr int KB = BWD_KB_V;                  // candidates replayed per branch-free group
#ifndef BWD_CHB_V
#define BWD_CHB_V 4
#endif
    static constexpr int CHB = BWD_CHB_V;                // candidates staged per batch (4: 12.2 KB of LDS per wave = 12 waves per CU; 8: 11 -- 1 495 -> 1 406 us at cfg5, 344 -> 329 us at cfg4; 16: 1 655 us)
    static constexpr int SB = 4;                         // candidates per phase-B contraction (16 panel rows = 4 x 4 corners)
    static constexpr int NC0 = 7 + S;                    // colour3, normal3, depth, feature S  (<= 16)
    static constexpr int GPROW = NC0 + 1;                // row stride of the transposition tile of the NC0 "plain" columns
    static constexpr int GV = VC + 1;                    // row stride of the vfeature columns of G kept in LDS
    static constexpr int PS = 64;                        // panel row stride (floats); rows are XOR-swizzled instead of padded (panel_at)
#ifndef BWD_WPE_V
#define BWD_WPE_V 4
#endif
    // Waves per SIMD the register budget is held to.  The kernel is bound by latency at its occupancy (measured by padding the LDS
    // request, cfg5 / cfg3_train: 12 waves per CU 1 395 / 322 us, 11: 1 507 / 333, 10: 1 539 / 349, 9: 1 665 / 371, 8: 1 765 / 397, 5: 2 412 /
    // 547 -- T = a + b / waves with a = 655 / 170 us), so the footprint of a wave is what counts: 4 waves per SIMD need <= 128 VGPRs and
    // <= 10 240 B of LDS (160 KB / 16, allocated in 1 280-byte granules).  Hence: the segment's {gid, slot} entries live in registers
    // (lane = list position; read with v_readlane / ds_bpermute) instead of a 512-byte LDS copy; the weight panel's rows are XOR-swizzled
    // instead of padded; and the six geometric gradients of a candidate are summed over the pixels by one transposed DPP tree + two
    // v_permlane*_swap steps instead of a 6 x 64 LDS panel: 12.2 KB -> 9.8 KB per wave.
    static constexpr int WPE = BWD_WPE_V;
    static constexpr size_t off_p = (size_t)CHB * SG::NF * 4;
    static constexpr int PROWS = 16;                                  // weight panel rows: (candidate, corner)
    static constexpr size_t g_bytes = (size_t)64 * GV * 4;
    // G (the MFMA B operand): its NC0 plain columns are transposed once per segment through LDS (a tile that aliases the
    // weight panel) into 16 VGPRs; the VC vfeature columns stay in LDS for the whole segment (they are also the A operand of
    // the h contraction; row pitch GV = VC + 1: with a pitch of 16 the phase-B reads of the four 16-pixel groups would hit the same banks).
    static constexpr size_t off_g = off_p + (size_t)PROWS * PS * 4;
    static constexpr size_t lds_bytes = off_g + g_bytes;
    static_assert(NC0 <= 16 && VC <= 16, "one 16-wide MFMA column tile per channel group");
    static_assert(KB == 1 && SB % KB == 0 && CHB % SB == 0, "batch nesting");
    static_assert(lds_bytes <= 10240, "4 waves per SIMD = 16 per CU need <= 10 KB of LDS per wave");
    static_assert((size_t)64 * GPROW * 4 <= (size_t)PROWS * PS * 4, "the transposition tile aliases the weight panel");
};

// element (row r, pixel / column c) of the 16 x 64 weight panel: the row's 16-byte granules are permuted by the row number, so that the
// column-wise accesses of the MFMA operand loads (16 rows, same columns) hit 16 different bank groups without padding the rows
__device__ __forceinline__ int panel_at(int r, int c) { return r * 64 + (c ^ ((r & 15) << 2)); }

// lane l of every 16-lane row <- sum over the four rows of lane (l & 15): two v_permlane*_swap of the register with a copy of itself
// (gfx950; scripts/probes/permlane_probe.hip) -- inline asm: with the builtins' two-element result this compiler adds element 0 to itself
__device__ __forceinline__ float rows_total(float u) {
    float x = u, y = u;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
    const float s = x + y;
    float p = s, q = s;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(p), "+v"(q));
    return p + q;
}

template <int S, int VC, bool SVGSS>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(BwdGeom<S, VC>::WPE, BwdGeom<S, VC>::WPE)))
render_bwd_kernel(const RenderBwdArgs a) {
    constexpr int VS = VC * 4;
    using SG = StageGeom<S, VC>;
    using BG = BwdGeom<S, VC>;
    constexpr int SB = BG::SB, NC0 = BG::NC0, GV = BG::GV, GPROW = BG::GPROW, CHB = BG::CHB, PS = BG::PS;
    constexpr int SS = S > 0 ? S : 1, VV = VC > 0 ? VC : 1;
    constexpr int P4 = (NC0 + 3) / 4 * 4, GEO = P4 + VS, RS = (GEO + 6 + 3) / 4 * 4;   // common.hpp GradRowGeom
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sD = reinterpret_cast<float*>(smem);                    // [CHB][NF] staged candidates
    float* sP = reinterpret_cast<float*>(smem + BG::off_p);        // [16][64] blend-weight panel (MFMA A operand), swizzled: panel_at
    float* sG = reinterpret_cast<float*>(smem + BG::off_g);        // [64][GV] upstream vfeature gradients of the sub-tile (MFMA operands)

    const int lane = threadIdx.x;
    const bool surface = cfg_flag(a.cfg, 0), normalize_depth = cfg_flag(a.cfg, 1);
    const bool sp = surface && cfg_flag(a.cfg, 2);
    const bool bgeom = SVGSS ? true : (a.backward_geometry != 0);
    const size_t N_ = (size_t)a.W * a.H;
    const float ddelx_dx = 0.5f * a.W, ddely_dy = 0.5f * a.H;
    const int colB = lane & 15, grpB = lane >> 4;

    // The staging buffer and the weight panel start as zeros: slots beyond a batch's size then always hold finite values
    // (zeros or an older candidate), so the replay needs no per-candidate bounds branches -- such slots get weight 0.
    for (int i = lane; i < (int)(BG::off_p / 16); i += 64) reinterpret_cast<float4*>(sD)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = lane; i < BG::PROWS * PS / 4; i += 64) reinterpret_cast<float4*>(sP)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    // The waves stride over the list of live depth segments (common.hpp SEG; seg_build_kernel): one 32-byte descriptor per
    // segment, fetched with scalar loads.  The count and this wave's first descriptor are independent loads (the list
    // position of a work id does not depend on the count: common.hpp seg_item_of).
    typedef const __attribute__((address_space(4))) uint32_t cu32;
    cu32* dsc0 = (cu32*)(uintptr_t)(a.seg_desc + min(seg_item_of(blockIdx.x), (uint32_t)a.seg_cap));
    uint32_t d_sm = dsc0[0], d_r0 = dsc0[1], d_len = dsc0[2], d_count = dsc0[3], d_ndump = dsc0[4], d_pb = dsc0[5], d_sb = dsc0[6];
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
    for (uint32_t wi = blockIdx.x; wi < nwork; wi += gridDim.x) {
    const uint32_t item = seg_item_of(wi);   // longest-first list, dealt to the XCDs in blocks of consecutive items (common.hpp)
    if (item >= nlive) continue;
    if (wi != blockIdx.x) {
        cu32* dsc = (cu32*)(uintptr_t)(a.seg_desc + item);
        d_sm = dsc[0]; d_r0 = dsc[1]; d_len = dsc[2]; d_count = dsc[3]; d_ndump = dsc[4]; d_pb = dsc[5]; d_sb = dsc[6];
    }
    DEV_TRACE_MARK(3);
    wave_lds_sync();   // the previous segment's LDS traffic is complete before its buffers are reused
    const uint32_t sm = d_sm, r0 = d_r0, tlen = d_len;
    const int sid = (int)(sm >> SEG_K_BITS), kseg = (int)(sm & ((1u << SEG_K_BITS) - 1u));
    const int tile = sid >> 2, sub = sid & 3;
    const int count = (int)d_count;
    const int ndump = (int)d_ndump;
    const int seg_lo = kseg * SEG, seg_hi = min(count, seg_lo + SEG);
    if (seg_hi <= seg_lo) continue;
    const int tx = tile % a.gx, ty = tile / a.gx;
    const int px = tx * TILE + (sub & 1) * 8 + (lane & 7);
    const int py = ty * TILE + (sub >> 1) * 8 + (lane >> 3);
    const bool inside = px < a.W && py < a.H;
    const float pxf = (float)px, pyf = (float)py;
    const uint2* __restrict__ sub_in = a.sub_list + (size_t)4 * r0 + (size_t)sub * tlen;
    const size_t pid = inside ? (size_t)a.W * py + px : 0;
    const int nent = seg_hi - seg_lo;
    // gradient row of the candidate at segment position i (0 = deepest): compact, in list order (common.hpp)
    const uint32_t row_top = d_pb + (uint32_t)(seg_hi - 1);

    // ---- set-up loads that depend only on the descriptor, issued together: the segment's list entries (deepest first: the replay
    // walks back to front; lane = list position, they STAY in these two registers), the pixel's forward results and upstream gradients
    static_assert(SEG == 64, "one list entry per lane");
    uint2 ent = make_uint2(0u, 0u);
    if (lane < nent) ent = sub_in[seg_hi - 1 - lane];
    const float T_final = inside ? a.final_T[pid] : 0.f;
    const float D_final = (inside && normalize_depth) ? a.final_D[pid] : 0.f;
    const uint32_t last_contributor = inside ? (uint32_t)a.n_contrib[pid] : 0u;
    float gC[3], gN[3], gF[SS], gD = 0.f, gO = 0.f;
#pragma unroll
    for (int i = 0; i < 3; i++) { gC[i] = (inside && a.g_color) ? a.g_color[i * N_ + pid] : 0.f; gN[i] = (inside && a.g_normal) ? a.g_normal[i * N_ + pid] : 0.f; }
#pragma unroll
    for (int i = 0; i < SS; i++) gF[i] = (inside && i < S && a.g_feature) ? a.g_feature[i * N_ + pid] : 0.f;
    if (inside) { gD = a.g_depth ? a.g_depth[pid] : 0.f; gO = a.g_opacity ? a.g_opacity[pid] : 0.f; }
    constexpr int NST = 8 + S + VC;

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

    // entry of list position i (0 = deepest) from the lanes' registers: per-lane positions through the LDS crossbar (no LDS memory),
    // wave-uniform ones with v_readlane
    auto gid_at = [&](int i) -> uint32_t { return (uint32_t)__builtin_amdgcn_ds_bpermute(i << 2, (int)ent.x); };
    const int nskip = __popcll(__ballot(lane < nent && ent.y >= wmax));   // entries behind every pixel of this wave (a prefix: slots descend)
    StageRegs<S, VC, CHB> sr;
    int base = (nskip / CHB) * CHB;
    if (base < nent)   // the first batch's gathers start now and overlap the rest of the set-up
        stage_load<S, VC, CHB>(sr, min((int)CHB, nent - base), [&](int s) { return gid_at(base + s); }, lane, a.rec, a.features, a.vfeatures);

    // Replay state.  The reference keeps, per channel, the blend of everything behind the current splat (accum_rec)
    // and the last value, and adds (value - accum) * dL_dchannel to dL_dalpha for every channel.  The upstream
    // gradient of a pixel is the same for every splat, so the per-channel recurrences
    //     accum <- last_alpha * last + (1 - last_alpha) * accum
    // collapse into ONE scalar recurrence on A = sum_ch accum_ch * g_ch with s = sum_ch value_ch * g_ch:
    //     A <- last_alpha * s_last + (1 - last_alpha) * A ;   dL_dalpha += s - A.
    float T = T_final;
    float last_alpha = 0.f;
    float A_acc = 0.f, s_last = 0.f;

    // G matrix of this sub-tile: row = pixel (lane), columns = [colour3 | normal3 x10 (Q4) | depth | feature S | vfeature VC].
    // Phase B needs it as the MFMA B operand (lane l: G[pixel = 16*(l>>4) + kk][channel = l&15], kk = 0..15): the NC0 plain
    // columns are transposed once through LDS (a tile that aliases the weight panel) into 16 registers, the vfeature columns
    // stay in LDS (sG, row stride GV).  The vfeature gradients also enter the start value of A (below) and are dead afterwards:
    // they are loaded, used and dropped in two halves so that the set-up never holds more than the loop does.
    float Bp[16];
    {
        const bool from_state = kseg < ndump;
        // Not the deepest live segment: start from the forward state dumped at this segment's far end.  With last_alpha = 0 the
        // recurrence takes accum = blend of everything behind = (final - prefix) / T_end.
        // (state layout: T | colour 3 | normal 3 | depth | feature S | vfeature VC)
        const float* e = a.seg_state + ((size_t)(d_sb + (uint32_t)kseg) * NST) * 64 + lane;
        const float* f = a.seg_state + ((size_t)(d_sb + (uint32_t)ndump) * NST) * 64 + lane;   // final state
        float dot = 0.f;
        if (from_state) {
            T = e[0];
            float dd[7 + SS];
#pragma unroll
            for (int i = 0; i < 7 + S; i++) dd[i] = f[(1 + i) * 64] - e[(1 + i) * 64];
            dot = dd[6] * gDn;
#pragma unroll
            for (int i = 0; i < 3; i++) {
                dot += dd[i] * gC[i];
                dot += dd[3 + i] * gN[i];   // zero unless `surface` (the forward leaves N at 0)
            }
            if (bgeom) {
#pragma unroll
                for (int i = 0; i < S; i++) dot += dd[7 + i] * gF[i];
            }
        }
        float* g = sP + lane * GPROW;
        g[0] = gC[0]; g[1] = gC[1]; g[2] = gC[2];
        g[3] = surface ? gN[0] * 10.f : 0.f; g[4] = surface ? gN[1] * 10.f : 0.f; g[5] = surface ? gN[2] * 10.f : 0.f;
        g[6] = gDn;
#pragma unroll
        for (int i = 0; i < S; i++) g[7 + i] = gF[i];
        float* gv = sG + lane * GV;
        constexpr int VH = (VC + 1) / 2;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            float gq[VH], dq[VH];
#pragma unroll
            for (int i = 0; i < VH; i++) {
                const int ch = h * VH + i;
                gq[i] = (ch < VC && inside && a.g_vfeature) ? a.g_vfeature[(size_t)ch * N_ + pid] : 0.f;
                dq[i] = (ch < VC && from_state) ? f[(8 + S + ch) * 64] - e[(8 + S + ch) * 64] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < VH; i++) {
                const int ch = h * VH + i;
                if (ch < VC) { gv[ch] = gq[i]; dot += dq[i] * gq[i]; }
            }
        }
        if (from_state) A_acc = dot * __builtin_amdgcn_rcpf(T);
        wave_lds_sync();
        const float* gB = sP + (16 * grpB) * GPROW + (colB < NC0 ? colB : 0);
#pragma unroll
        for (int kk = 0; kk < 16; kk++) Bp[kk] = colB < NC0 ? gB[kk * GPROW] : 0.f;
        wave_lds_sync();
        // the weight panel starts as zeros again (slots beyond a batch's size must hold finite values)
        for (int i = lane; i < BG::PROWS * PS / 4; i += 64) reinterpret_cast<float4*>(sP)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    DEV_TRACE_MARK(0);   // segment setup
    dev_items++; dev_cands += (unsigned)nent;
    // Batches of CHB candidates; the gathers of batch b+1 are in flight (registers) while batch b is replayed, so
    // neither their latency nor the completion of this batch's gradient stores is waited for.
    for (; base < nent; base += CHB) {
        const int m = min((int)CHB, nent - base);
        wave_lds_sync();  // previous batch fully consumed
        stage_store<S, VC, CHB>(sr, sD, m, lane);
        {
            const int nb = base + CHB;
            if (nb < nent)
                stage_load<S, VC, CHB>(sr, min((int)CHB, nent - nb), [&](int s) { return gid_at(nb + s); }, lane, a.rec, a.features, a.vfeatures);
        }
        wave_lds_sync();
        DEV_TRACE_MARK(1);   // staging

        for (int c0 = 0; c0 < m; c0 += SB) {
            uint32_t live = 0;  // bit cs set: candidate c0+cs has at least one blending pixel (wave-uniform)
            if (sp) {
                // h[pixel][(candidate, corner)] = sum_ch vfeature[candidate][ch][corner] * gVF[pixel][ch] for the SB = 4
                // candidates of this block on the matrix pipe: M = pixels (4 tiles of 16), K = channels (4 steps of 4),
                // N = (candidate, corner).  A = the gVF columns of sG, B = the staged vfeature floats (lane: channel
                // 4 ks + (l >> 4), candidate (l & 15) >> 2, corner l & 3).  The D tiles (lane l, register r: pixel
                // 16 mt + 4 (l >> 4) + r, column l & 15) go to the weight panel's rows (candidate, corner) -- the rows phase A
                // later overwrites with the blend weights of the same (candidate, corner) -- and are read back lane = pixel.
                f32x4 hacc[4];
#pragma unroll
                for (int mt = 0; mt < 4; mt++) hacc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const float* vb = sD + (c0 + (colB >> 2)) * SG::NF + SG::V_OFF + (colB & 3);
#pragma unroll
                for (int ks = 0; ks < (VC + 3) / 4; ks++) {
                    const int ch = 4 * ks + grpB;
                    const float bv = ch < VC ? vb[4 * ch] : 0.f;
#pragma unroll
                    for (int mt = 0; mt < 4; mt++) {
                        const float av = ch < VC ? sG[(16 * mt + colB) * GV + ch] : 0.f;
                        hacc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, hacc[mt], 0, 0, 0);
                    }
                }
                wave_lds_sync();   // (the previous block's phase B has read its panel)
#pragma unroll
                for (int mt = 0; mt < 4; mt++) *reinterpret_cast<f32x4*>(sP + panel_at(colB, 16 * mt + 4 * grpB)) = hacc[mt];
                wave_lds_sync();
            }
            // ---------------- phase A: lane = pixel, one candidate at a time ----------------
#pragma unroll 1
            for (int cs = 0; cs < SB && c0 + cs < m; cs++) {
                const int cb = c0 + cs;   // the candidate's slot in the staging buffer
                // (1) alpha, corner weights, s = sum_ch value_ch * g_ch
                const float4* q = reinterpret_cast<const float4*>(sD + cb * SG::NF);
                const float4 A = q[0];    // x, y, conic.x, conic.y
                const float4 B = q[1];    // conic.z, opacity, depth, DA
                const float4 E = q[3];    // DB, r, g, b
                const float4 Nn = q[4];   // nx, ny, nz, 1/umax
                const uint32_t slot = (uint32_t)__builtin_amdgcn_readlane((int)ent.y, base + cb);
                const float dx = A.x - pxf, dy = A.y - pyf;
                const float pw = pair_power(A.z, A.w, B.x, dx, dy);
                const float Gs = exp_nonpos(pw);
                const float al = fminf(0.99f, B.y * Gs);
                const bool pre = slot < last_contributor && pw <= 0.0f && al >= (1.0f / 255.0f);
                const float ioma = __builtin_amdgcn_rcpf(1.f - al);
                float cw0 = 0.f, cw1 = 0.f, cw2 = 0.f, cw3 = 0.f;
                if (SVGSS && sp) {
                    const float4 Jv = q[2];  // J0..J3
                    const float ivm = q[5].x;
                    const float du = dx * Jv.x + dy * Jv.y, dv = dx * Jv.z + dy * Jv.w;
                    float u = du * Nn.w * 0.5f + 0.5f, v = dv * ivm * 0.5f + 0.5f;
                    u = fminf(0.999f, fmaxf(0.001f, u));
                    v = fminf(0.999f, fmaxf(0.001f, v));
                    cw0 = (1.f - u) * (1.f - v); cw1 = u * (1.f - v); cw2 = (1.f - u) * v; cw3 = u * v;
                }
                float sd = E.y * gC[0] + E.z * gC[1] + E.w * gC[2];
                if (surface) sd += Nn.x * gN[0] + Nn.y * gN[1] + Nn.z * gN[2];
                float d_cur = B.z;
                if (sp) d_cur -= dx * B.w + dy * E.x;   // depth differencing (common.hpp R_DA / R_DB)
                sd += d_cur * gDn;
                if (S > 0 && bgeom) {
                    const float* fp = sD + cb * SG::NF + SG::F_OFF;
#pragma unroll
                    for (int ch = 0; ch < S; ch++) sd += fp[ch] * gF[ch];
                }
                // sum_ch (c4[ch] . cw) gVF[ch] = cw . (sum_ch c4[ch] gVF[ch]) = cw . h (h: the block's MFMA above)
                if (sp) {
                    const float h0 = sP[panel_at(cs * 4, lane)], h1 = sP[panel_at(cs * 4 + 1, lane)];
                    const float h2 = sP[panel_at(cs * 4 + 2, lane)], h3 = sP[panel_at(cs * 4 + 3, lane)];
                    sd += (h0 * cw0 + h1 * cw1) + (h2 * cw2 + h3 * cw3);
                }
                // (2) the sequential part: T <- T / (1 - alpha) and the scalar replay recurrence (backward.cu:700-850)
                const float inv_Told = __builtin_amdgcn_rcpf(T);
                const float Tn = T * ioma;
                const float An = last_alpha * s_last + (1.f - last_alpha) * A_acc;
                float dL_dalpha = kdn * inv_Told + (sd - An);
                dL_dalpha *= Tn;
                dL_dalpha += gO_kbg * (T_final * ioma);
                T = pre ? Tn : T;
                A_acc = pre ? An : A_acc;
                s_last = pre ? sd : s_last;
                last_alpha = pre ? al : last_alpha;
                const float dLa = pre ? dL_dalpha : 0.f;
                const float vw = pre ? al * Tn : 0.f;
                // (3) weight panel rows (candidate, corner); the four corner weights sum to the blend weight
                sP[panel_at(cs * 4, lane)] = sp ? cw0 * vw : vw;
                sP[panel_at(cs * 4 + 1, lane)] = cw1 * vw;
                sP[panel_at(cs * 4 + 2, lane)] = cw2 * vw;
                sP[panel_at(cs * 4 + 3, lane)] = cw3 * vw;
                const bool clive = __ballot(pre) != 0ull;   // (uniform) some pixel blends this candidate
                live |= (clive ? 1u : 0u) << cs;
                if (!clive) continue;
                // (4) the six geometric gradients, summed over the 64 pixels without LDS: a transposed reduction tree (a lane keeps the value
                // its low bits name and hands the other one over) leaves the quad sums of g0..g3 in u (lane l: value l & 3) and of g4, g5 in
                // w (value 4 + (l & 1)); two row shifts put every 16-lane row's sums into its lanes 12..15; w moves to lanes 8..11; the four
                // rows meet in rows_total.  Lanes 12..15 then hold the totals of g0..g3, lanes 8, 9 those of g4, g5.
                {
                    const float dL_ddist = dLa * (B.y * -0.5f) * Gs;
                    float g0 = dL_ddist * 2.f * (A.z * dx + A.w * dy) * ddelx_dx;
                    float g1 = dL_ddist * 2.f * (B.x * dy + A.w * dx) * ddely_dy;
                    if (sp) { g0 += q5g * B.w; g1 += q5g * E.x; }   // + Q5, d(depth offset)/d(mean2D) = (DA, DB)
                    g0 = pre ? g0 : 0.f; g1 = pre ? g1 : 0.f;
                    const float g2 = dL_ddist * (dx * dx), g3 = dL_ddist * (dx * dy), g4 = dL_ddist * (dy * dy);   // (dL_ddist = 0 where !pre)
                    const float g5 = Gs * dLa;
                    const bool o1 = (lane & 1) != 0, o2 = (lane & 2) != 0;
                    const float t01 = (o1 ? g1 : g0) + dpp_f32<0xB1>(o1 ? g0 : g1);   // quad_perm [1,0,3,2]
                    const float t23 = (o1 ? g3 : g2) + dpp_f32<0xB1>(o1 ? g2 : g3);
                    float u = (o2 ? t23 : t01) + dpp_f32<0x4E>(o2 ? t01 : t23);         // quad_perm [2,3,0,1]
                    float w = (o1 ? g5 : g4) + dpp_f32<0xB1>(o1 ? g4 : g5);
                    w += dpp_f32<0x4E>(w);
                    u += dpp_f32<0x114>(u); w += dpp_f32<0x114>(w);   // row_shr:4
                    u += dpp_f32<0x118>(u); w += dpp_f32<0x118>(w);   // row_shr:8
                    const float w8 = dpp_f32<0x104>(w);               // row_shl:4: lanes 8..11 <- lanes 12..15
                    const float tot = rows_total((lane & 12) == 8 ? w8 : u);
                    const int jq = (lane & 4) ? (lane & 3) : 4 + (lane & 1);
                    if (lane >= 8 && lane < 16 && (lane & 14) != 10) {   // lanes 8, 9, 12, 13, 14, 15
                        // gradient row of the (instance, sub-tile) pair (see phase B)
                        const float* rr = sD + cb * SG::NF;
                        const uint32_t ib = __builtin_bit_cast(uint32_t, rr[R_IBASE]);
                        const uint32_t rc = __builtin_bit_cast(uint32_t, rr[R_RECT]);
                        const uint32_t x0 = rc & 1023u, y0 = (rc >> 10) & 1023u, wr = rc >> 20;
                        const size_t sl = (size_t)4 * (ib + ((uint32_t)ty - y0) * wr + ((uint32_t)tx - x0)) + (uint32_t)sub;
                        const uint32_t row = row_top - (uint32_t)(base + cb);
                        if (row < a.rows_cap) {
                            a.grad_rows[(size_t)row * RS + GEO + jq] = tot;
                            if (jq == 0) a.row_of[sl] = row + 1u;   // where grad_reduce finds this pair's row
                        }
                    }
                }
            }
            DEV_TRACE_MARK(2);   // phase A
            if (live == 0) continue;  // uniform
            wave_lds_sync();          // the weight panel is complete

            // ---------------- phase B: panel x G on the matrix pipe ----------------
            {
                float av[16];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float4 t4 = *reinterpret_cast<const float4*>(sP + panel_at(colB, 16 * grpB + 4 * j));
                    av[4 * j] = t4.x; av[4 * j + 1] = t4.y; av[4 * j + 2] = t4.z; av[4 * j + 3] = t4.w;
                }
                f32x4 accP = {0.f, 0.f, 0.f, 0.f}, accV = {0.f, 0.f, 0.f, 0.f};
                const float* gB = sG + (16 * grpB) * GV;
                const int colV = colB < VC ? colB : 0;
#pragma unroll
                for (int kk = 0; kk < 16; kk++) {
                    accP = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], Bp[kk], accP, 0, 0, 0);
                    const float bv = gB[kk * GV + colV];
                    accV = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], colB < VC ? bv : 0.f, accV, 0, 0, 0);
                }
                // D layout: lane l, register r -> row 4*(l>>4) + r, column l&15
                // ---- gradient rows (common.hpp GradRowGeom): plain stores, summed per Gaussian afterwards ----
                // rows of the panel = (candidate grpB, corner r); the candidate's gradient row: compact, in list order
                const bool mine = c0 + grpB < m && ((live >> grpB) & 1u);
                if (mine) {
                    const uint32_t rowi = row_top - (uint32_t)(base + c0 + grpB);
                    float* row = a.grad_rows + (size_t)rowi * RS;
                    if (rowi < a.rows_cap) {
                        if (colB < NC0) row[colB] = (accP[0] + accP[1]) + (accP[2] + accP[3]);
                        if (colB < VC)
                            reinterpret_cast<float4*>(row + P4)[colB] =
                                sp ? make_float4(accV[0], accV[1], accV[2], accV[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
            }
            wave_lds_sync();   // panel consumed before the next block overwrites it
            DEV_TRACE_MARK(3);   // phase B
        }
    }
    }   // loop over live segments
    DEV_TRACE_END(1, dev_items, dev_cands, blockIdx.x);
}

#ifndef BWD_LDS_MIN
#define BWD_LDS_MIN 0
#endif
template <int S, int VC, bool SVGSS>
void launch(const RenderBwdArgs& a, hipStream_t s) {
    using BG = BwdGeom<S, VC>;
    // The live-segment count only exists on the device: launch a few waves per resident slot (256 CUs x 4 SIMDs x WPE),
    // capped by the list's upper bound, and let them stride over the list.  Small workloads get one segment per wave
    // (the dispatcher balances), large ones several; waves beyond the count exit after one scalar load.
    const int grid = std::max(8, std::min(a.seg_cap, 4 * 256 * 4 * BG::WPE) & ~7);   // a multiple of 8: work id & 7 = XCD in every round
    // (the LDS request doubles as a residency control for occupancy sweeps: scripts/build_variant.sh -DBWD_LDS_MIN=...)
    hipLaunchKernelGGL((render_bwd_kernel<S, VC, SVGSS>), dim3(grid), dim3(64), std::max(BG::lds_bytes, (size_t)BWD_LDS_MIN), s, a);
}

}  // namespace

int launch_render_bwd(const RenderBwdArgs& a, bool svgss, hipStream_t s) {
    const int VC = a.VS / 4;
    if (VC == 0) return launch_render_bwd_plain(a, svgss, s);   // widths without vfeatures: render_bwd_plain.hip
#define CASE(SV, VCV, SG) if (a.S == SV && VC == VCV && svgss == SG) { launch<SV, VCV, SG>(a, s); return 0; }
    CASE(4, 13, true) CASE(7, 16, true) CASE(3, 2, true) CASE(1, 1, true)
#undef CASE
    return -1;
}

#if defined(SVGIR_DEV)
extern "C" int svgir_dev_trace_read_bwd(unsigned long long* out, int cap_records) {
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
