// svg-ir_amd/csrc/render_fwd.hip -- forward alpha compositing.
//
// Replaces renderCUDA (svgss forward.cu:401-750, rgss forward.cu:323-535): front-to-back blending of the
// depth-ordered splat list of each 16x16 tile; per-pixel depth by depth differencing; svgss additionally blends
// VS/4 "vfeature" channels, each the bilinear interpolation of 4 corner values in the surfel's tangent plane.
//
// CDNA4 mapping (see stage.hpp for the staging details): two kernels.
//   cull_kernel: one 256-thread workgroup per tile walks the tile's depth-ordered list 256 entries per round, one
//     splat per lane: coalesced id load, 32-byte header gather, exact minimum of the conic form over each of the
//     tile's four 8x8 pixel rectangles against the 1/255 alpha threshold (the per-splat part of the test is shared by
//     the four rectangles, the header is gathered once instead of once per sub-tile).  Survivors are compacted in
//     order (ballot + popcount in the wave, per-wave counts through LDS across the waves) into one list per sub-tile
//     {Gaussian id, slot in the tile list}; forward and backward composite both consume these lists.
//   render_fwd_kernel: one wave64 per 8x8 sub-tile (one wave per workgroup, no barriers), sub-tiles dispatched by
//     descending candidate count.  Candidates are staged CH at a time into LDS in the pair-interleaved layout of
//     pairstage.hpp (16-byte gathers, all in flight together and issued one batch AHEAD) and consumed four at a time:
//     the alphas of a group are packed fp32 instructions on two candidates each, only the transmittance chain
//     (T <- T (1 - alpha), the 1e-4 cut-off) is scalar and sequential, and the colour / normal / feature / vfeature sums
//     are MFMAs (svgss widths: blend weights -> LDS panel -> A operands, staged channel rows -> B operands) or packed
//     FMAs on channel pairs (rgss widths);
//   * channel counts are template parameters so every accumulator lives in a VGPR (the reference keeps >640 floats
//     per thread in scratch, forward.cu:483-493);
//   * the per-(pixel,splat) out_weights atomic of the reference (forward.cu:653) becomes one DPP wave reduction per
//     (wave, splat) parked in LDS and ONE atomic instruction per staging batch (lane = candidate);
//   * after every SEG-th candidate (and once at the end) the wave dumps its blend state -- T and every accumulator,
//     [state][channel][64 pixels] -- and appends the live backward segments of its sub-tile to a compact list
//     (common.hpp SEG): the backward is parallel over depth segments and starts each one from these states.
//
// Numerics: alpha is evaluated in exactly the operation order of the reference's source (no FMA contraction in the
// quadratic form) with a ~1 ulp exp, so the alpha >= 1/255 and T < 1e-4 decisions agree with a plain fp32
// evaluation of the reference's formulas except where exp itself differs in the last bit.
#include <algorithm>

#include "common.hpp"
#include "shade_tables.hpp"
#include "stage.hpp"
#include "pairstage.hpp"
#include "dev_trace.hpp"

namespace svgir {

namespace {

// ---- cull: tile list -> four compact sub-tile lists -----------------------------------------------------------
#ifndef CULL_THREADS
#define CULL_THREADS 256
#endif
constexpr int CT = CULL_THREADS, CNW = CT / 64;   // threads / waves of a cull workgroup (a tile's list is walked CT entries per round)

// the result of a pixel nothing is blended into (forward.cu:665-700 with an empty list)
__device__ __forceinline__ void write_background(const RenderArgs& a, size_t pid, size_t N_) {
    const float T = (float)(1 - 0.000001);   // forward.cu:671
    a.final_T[pid] = T; a.final_D[pid] = 0.f; a.n_contrib[pid] = 0;
    a.out_color[pid] = T * a.bg[0]; a.out_color[N_ + pid] = T * a.bg[1]; a.out_color[2 * N_ + pid] = T * a.bg[2];
    for (int ch = 0; ch < a.S; ch++) a.out_feature[ch * N_ + pid] = 0.f;
    for (int ch = 0; ch < a.VS / 4; ch++) a.out_vfeature[ch * N_ + pid] = 0.f;
    a.out_normal[pid] = 0.f; a.out_normal[N_ + pid] = 0.f; a.out_normal[2 * N_ + pid] = 0.f;
    a.out_depth[pid] = cfg_flag(a.cfg, 1) ? 0.f / (1.f - T) : 0.f + T * 10.f;
    a.out_opacity[pid] = 1.f - T;
}
__device__ __forceinline__ void write_zero_planes(const RenderArgs& a, size_t pid, size_t N_) {   // planes this call leaves at zero
    if (a.zero_a) { a.zero_a[pid] = 0.f; a.zero_a[N_ + pid] = 0.f; a.zero_a[2 * N_ + pid] = 0.f; }
    if (a.zero_b) { a.zero_b[pid] = 0.f; a.zero_b[N_ + pid] = 0.f; a.zero_b[2 * N_ + pid] = 0.f; }
}

__global__ void __launch_bounds__(CT) cull_kernel(const RenderArgs a) {
    __shared__ uint32_t wcnt[CNW][4];   // [wave][sub-tile] survivors of the current round
    // (the tiles are dealt to the XCDs in blocks of 4 x 4: a splat's tiles are culled on one XCD -- common.hpp xcd_tile_of_work)
    const uint32_t tile_u = xcd_tile_of_work(blockIdx.x, a.gx, a.gy);
    if (tile_u == ORDER_NONE) return;
    const int tile = (int)tile_u;
    const uint32_t r0 = a.ranges[2 * tile], r1 = a.ranges[2 * tile + 1];
    const int len = (int)(r1 - r0);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int tx = tile % a.gx, ty = tile / a.gx;
    // (bg_in_render: the specialised composite kernel writes the empty tiles' pixels and the all-zero planes itself -- its empty
    // sub-tiles' waves are dispatched last and run in its idle tail: 6.5 us less here at cfg2, nothing more there)
    if (!a.bg_in_render && t < 256 && (a.zero_a || a.zero_b)) {   // planes this call leaves at zero: one thread per pixel of the tile
        const int px = tx * TILE + (t & 15), py = ty * TILE + (t >> 4);
        if (px < a.W && py < a.H) write_zero_planes(a, (size_t)a.W * py + px, (size_t)a.W * a.H);
    }
    if (len == 0) {
        // Empty tile: nothing will ever be blended here.  Run-time-width composite: the 256 threads write the background result of the
        // whole 16x16 tile (64-byte rows) and the four composite waves of the tile exit at once.
        if (t < 4) { a.sub_total[4 * tile + t] = 0u; a.sub_count[4 * tile + t] = 0u; a.sub_ndump[4 * tile + t] = 0u; }
        const int px = tx * TILE + (t & 15), py = ty * TILE + (t >> 4);
        if (!a.bg_in_render && t < 256 && px < a.W && py < a.H) write_background(a, (size_t)a.W * py + px, (size_t)a.W * a.H);
        return;
    }
    const float X0 = (float)(tx * TILE), Y0 = (float)(ty * TILE);
    const float4* __restrict__ rec4 = reinterpret_cast<const float4*>(a.rec);
    uint2* __restrict__ out = a.sub_list + (size_t)4 * r0;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    uint32_t run[4] = {0u, 0u, 0u, 0u};   // survivors so far per sub-tile (uniform)
    // software pipeline over the rounds: ids two rounds ahead, record headers one round ahead (a round was three dependent
    // memory round trips; the longest lists set the kernel's duration)
    uint32_t gid_n = t < len ? a.point_list[r0 + t] : 0u;           // round 0
    uint32_t gid_nn = CT + t < len ? a.point_list[r0 + CT + t] : 0u;   // round 1
    float4 A_n = rec4[(size_t)gid_n * 6], B_n = rec4[(size_t)gid_n * 6 + 1];
    for (int base = 0; base < len; base += CT) {
        const int i = base + t;
        bool m[4] = {false, false, false, false};
        const uint32_t gid = gid_n;
        const float4 A = A_n, B = B_n;
        gid_n = gid_nn;
        if (base + CT < len) { A_n = rec4[(size_t)gid_n * 6]; B_n = rec4[(size_t)gid_n * 6 + 1]; }
        gid_nn = base + 2 * CT + t < len ? a.point_list[r0 + base + 2 * CT + t] : 0u;
        if (i < len) {
            const SplatCull sc = cull_prepare(A.x, A.y, A.z, A.w, B.x, B.y);
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const float x0 = X0 + (float)((w & 1) * 8), y0 = Y0 + (float)((w >> 1) * 8);
                m[w] = cull_test(sc, x0, y0, x0 + 7.f, y0 + 7.f);
            }
        }
        unsigned long long mask[4];
#pragma unroll
        for (int w = 0; w < 4; w++) {
            mask[w] = __ballot(m[w]);
            if (lane == 0) wcnt[wave][w] = (uint32_t)__popcll(mask[w]);
        }
        __syncthreads();
        const uint2 e = make_uint2(gid, (uint32_t)i);
#pragma unroll
        for (int w = 0; w < 4; w++) {
            uint32_t before = 0, all = 0;
#pragma unroll
            for (int v = 0; v < CNW; v++) { const uint32_t c = wcnt[v][w]; all += c; before += v < wave ? c : 0u; }
            if (m[w]) out[(size_t)w * len + run[w] + before + (uint32_t)__popcll(mask[w] & lt_mask)] = e;
            run[w] += all;
        }
        __syncthreads();   // counts consumed before the next round overwrites them
    }
    if (t < 4) a.sub_total[4 * tile + t] = t == 0 ? run[0] : t == 1 ? run[1] : t == 2 ? run[2] : run[3];
}

// ---- blend --------------------------------------------------------------------------------------------------
// WPE = the most waves per SIMD the kernel may run at (the register allocator is told "2 to WPE": it never squeezes the kernel below the
// 155-157 VGPRs it wants -- a "3 to 3" build spills 9-15 dwords -- and with WPE = 2 the allocation is padded so that no third wave fits).
// The svgss widths exist in two variants: WPE = 2 (8 waves per CU) for launches that do not fill the machine anyway, where a CU with
// three long lists on one SIMD is what the kernel ends on, and FWD_WPE_HI = 3 (11 waves per CU, the LDS limit) for launches of many
// rounds of waves, where resident waves are what hides the gathers' latency (cfg5: 666 -> 593 us, cfg5_dense 910 -> 834 us; cfg4, one
// round of waves: 213 -> 222 us) -- chosen per launch from the workload's fill (RenderArgs::hi_fill).
template <int S, int VC, bool SVGSS, int WPE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE < 2 ? WPE : 2, WPE)))
render_fwd_kernel(const RenderArgs a) {
    using PG = PairGeom<S, VC>;
    constexpr int CH = PG::CH, PF = PG::PF;
    constexpr int KB = StageGeom<S, VC>::KB, NP = KB / 2;   // candidates / staged pairs per branch-free group
    static_assert(KB == 4 && CH % KB == 0, "a group = the 4 K-rows of one MFMA");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sD = reinterpret_cast<float*>(smem);                  // [CH / 2][PF] staged pairs (pairstage.hpp)
    uint2* sQ = reinterpret_cast<uint2*>(smem + PG::off_q);      // [2][CH] {gid, slot}: this batch / next batch
    float* sW = reinterpret_cast<float*>(smem + PG::off_w);      // [2][4][CH] blend-weight sums of the four 16-lane rows
    float* sP = reinterpret_cast<float*>(smem + PG::off_p);      // [PROWS][PS] blend-weight panel (MFMA A operand) / transposition tile
    constexpr int PS = PG::PS;

    const uint32_t sid = a.sub_order[blockIdx.x];   // (grid = RenderArgs::order_n)
    if (sid == ORDER_NONE) return;                  // padding of the per-XCD order (common.hpp)
    const int tile = (int)(sid >> 2), sub = (int)(sid & 3u);
    const int tx = tile % a.gx, ty = tile / a.gx;
    const int lane = threadIdx.x;
    const int bx = tx * TILE + (sub & 1) * 8, by = ty * TILE + (sub >> 1) * 8;
    const int px = bx + (lane & 7), py = by + (lane >> 3);
    const bool inside = px < a.W && py < a.H;
    const float pxf = (float)px, pyf = (float)py;
    const f32x2 pxx = {pxf, pxf}, pyy = {pyf, pyf};
    const uint32_t r0 = a.ranges[2 * tile], r1 = a.ranges[2 * tile + 1];
    const int len = (int)(r1 - r0);
    if (len == 0) {   // empty tile: its pixels are the background (written here, or -- run-time widths -- by the cull kernel)
        if (a.bg_in_render && !a.dump_only && inside) {
            write_background(a, (size_t)a.W * py + px, (size_t)a.W * a.H);
            write_zero_planes(a, (size_t)a.W * py + px, (size_t)a.W * a.H);
        }
        return;
    }
    DEV_TRACE_DECL();
    const int total = (int)a.sub_total[sid];
    // The kernel ends when its longest candidate list has been walked (the walk is sequential per pixel), and waves are
    // dispatched longest-first: blockIdx.x is the wave's rank.  The SIMD's instruction arbiter serves the longer list first.
#ifndef FWD_PRIO
#define FWD_PRIO 1
#endif
    if (FWD_PRIO) {
        if (blockIdx.x < 1024u) __builtin_amdgcn_s_setprio(3);
        else if (blockIdx.x < 2048u) __builtin_amdgcn_s_setprio(2);
        else if (blockIdx.x < 3072u) __builtin_amdgcn_s_setprio(1);
    }
    const bool surface = cfg_flag(a.cfg, 0), normalize_depth = cfg_flag(a.cfg, 1);
    const bool sp = surface && cfg_flag(a.cfg, 2);
    const uint2* __restrict__ list = a.sub_list + (size_t)4 * r0 + (size_t)sub * len;

    bool done = !inside;
    float T = 1.0f, D = 0.f;
    // Channel accumulators (r g b nx ny nz F0..F(S-1)) and vfeature accumulators.
    //  * with vfeatures (svgss widths) they live on the matrix pipe: acc[mt] / vacc[mt] are the D tiles of
    //    v_mfma_f32_16x16x4_f32 for the pixels 16 mt .. 16 mt + 15 (lane l, register r: pixel 16 mt + 4 (l >> 4) + r,
    //    channel l & 15): the 4 + 52..64 products per (pixel, candidate) would otherwise be 13-16 broadcast ds_read_b128 and
    //    as many VALU instructions per candidate -- the forward at those widths was bound by exactly that;
    //  * without (rgss: 11 channels) packed FMAs on channel pairs (C0,C1) (C2,N0) ... are cheaper than the panel round trip.
    constexpr bool MF = VC > 0;
    constexpr int NPAIR = (PG::NCH + 1) / 2;
    f32x2 accp[NPAIR];
#pragma unroll
    for (int i = 0; i < NPAIR; i++) accp[i] = (f32x2){0.f, 0.f};
    f32x4 acc[4], vacc[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; vacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const int kq = lane >> 4, nq = lane & 15;
    // lane = pixel view of the accumulators (through the LDS tile): chv[c], vfv[c]
    float chv[PG::NCH], vfv[VC > 0 ? VC : 1];
    auto gather_acc = [&]() {
        if (!MF) {
#pragma unroll
            for (int c = 0; c < PG::NCH; c++) chv[c] = (c & 1) ? accp[c >> 1].y : accp[c >> 1].x;
            return;
        }
        wave_lds_sync();
#pragma unroll
        for (int mt = 0; mt < 4; mt++) {
            float* d = sP + nq * PS + 16 * mt + 4 * kq;
            d[0] = acc[mt].x; d[1] = acc[mt].y; d[2] = acc[mt].z; d[3] = acc[mt].w;
        }
        wave_lds_sync();
#pragma unroll
        for (int c = 0; c < PG::NCH; c++) chv[c] = sP[c * PS + lane];
        if (VC > 0) {
            wave_lds_sync();
#pragma unroll
            for (int mt = 0; mt < 4; mt++) {
                float* d = sP + nq * PS + 16 * mt + 4 * kq;
                d[0] = vacc[mt].x; d[1] = vacc[mt].y; d[2] = vacc[mt].z; d[3] = vacc[mt].w;
            }
            wave_lds_sync();
#pragma unroll
            for (int c = 0; c < VC; c++) vfv[c] = sP[c * PS + lane];
        }
        wave_lds_sync();
    };
    auto chan = [&](int c) -> float { return chv[c]; };
    auto vchan = [&](int c) -> float { return vfv[c]; };
    uint32_t last_walk = 0;   // 1 + position (in the sub-tile's candidate list) of the pixel's last contributor, 0: none

    // segment-boundary state dumps (see common.hpp SEG)
    constexpr int NST = 8 + S + VC;
    const uint32_t dump_base = a.sub_slot_base[sid];   // first state slot of this sub-tile (compact: common.hpp SEG)
    uint32_t ndump = 0;
    auto dump_state = [&](uint32_t j, bool last) {
        // no slot: a speculative launch whose capacity guess was too small (the view's backward dumps the states again), or a
        // forward_only view (slot capacity 0) -- then the accumulators are only gathered for the epilogue (`last`)
        const bool slot = dump_base + j < a.slot_cap;
        if (!slot && !last) return;
        gather_acc();
        if (!slot) return;
        float* d = a.seg_state + ((size_t)(dump_base + j) * NST) * 64 + lane;
        d[0] = T; d[64] = chan(0); d[128] = chan(1); d[192] = chan(2);
        // (the normal channels are blended unconditionally; without `surface` they do not exist for the consumers)
        d[256] = surface ? chan(3) : 0.f; d[320] = surface ? chan(4) : 0.f; d[384] = surface ? chan(5) : 0.f; d[448] = D;
#pragma unroll
        for (int ch = 0; ch < S; ch++) d[(8 + ch) * 64] = chan(6 + ch);
#pragma unroll
        for (int ch = 0; ch < VC; ch++) d[(8 + S + ch) * 64] = vchan(ch);
    };

    uint32_t head = 0;   // candidates consumed so far (wave-uniform)
    if (total > 0) {
        // The staging buffer starts as zeros: slots beyond a batch's size then always hold finite values (zeros or an
        // older candidate), so the blend loop needs no per-candidate bounds branches -- such slots get weight 0.  The
        // padding channels of the blocks are never written and stay zero.
        for (int i = lane; i < (CH / 2) * PF / 4; i += 64) reinterpret_cast<float4*>(sD)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        // software pipeline: {gid, slot} entries are fetched two batches ahead, records one batch ahead
        auto load_entries = [&](int b) -> uint2 {
            const int i = b * CH + lane;
            return (lane < CH && i < total) ? list[i] : make_uint2(0u, 0u);
        };
        const int nb = (total + CH - 1) / CH;
        if (lane < CH) sQ[lane] = load_entries(0);
        uint2 e_next = load_entries(1);
        wave_lds_sync();
        PairMap<S, VC> pmap;
        pmap.init(lane);
        PairRegs<S, VC> sr;
        pair_stage_load<S, VC, CH>(sr, pmap, min(CH, total), [&](int s) { return sQ[s].x; }, lane, a.rec, a.features, a.vfeatures);
        bool wave_done = __all(done);
        int nflush = 0;   // candidates of the previous batch whose out_weights sums are still parked in LDS
        // out_weights of a batch: one atomic instruction, lane = candidate.  It is issued one batch late, BEFORE the next
        // batch's gathers: a wait for those gathers then never waits for an atomic that was issued after them.
        auto flush_weights = [&](int bprev) {
            if (lane < nflush) {
                const float* wp = sW + (bprev & 1) * (4 * CH) + lane;
                const float wsum = (wp[0] + wp[CH]) + (wp[2 * CH] + wp[3 * CH]);
                if (wsum != 0.f && !a.dump_only) atomic_add_f32(&a.out_weights[sQ[(bprev & 1) * CH + lane].x], wsum);
            }
        };
        int b = 0;
        for (; b < nb && !wave_done; b++) {
            const int m = min(CH, total - b * CH);
            const uint2* q_cur = sQ + (b & 1) * CH;
            float* w_cur = sW + (b & 1) * (4 * CH);
            wave_lds_sync();   // previous batch fully consumed
            pair_stage_store<S, VC, CH>(sr, pmap, sD, m, lane);
            flush_weights(b - 1);
            wave_lds_sync();   // ... before its {gid, slot} entries are overwritten
            if (lane < CH) sQ[((b + 1) & 1) * CH + lane] = e_next;
            e_next = load_entries(b + 2);
            wave_lds_sync();
            if (b + 1 < nb) {
                const uint2* q_nxt = sQ + ((b + 1) & 1) * CH;
                pair_stage_load<S, VC, CH>(sr, pmap, min(CH, total - (b + 1) * CH), [&](int s) { return q_nxt[s].x; }, lane, a.rec,
                                       a.features, a.vfeatures);
            }
            DEV_TRACE_MARK(1);   // staging
            int nproc = m;   // candidates of this batch whose weight sums are valid
            for (int c0 = 0; c0 < m; c0 += KB) {
                // ---- (1) the alphas of KB candidates: NP pairs, packed (element-wise the reference's operations) ----
                f32x2 dx[NP], dy[NP];
                f32x4 Gd[NP], Ge[NP];   // (depth, DA) pairs; (DB, 1/umax) pairs
                float pw[KB], al[KB];
#pragma unroll
                for (int p = 0; p < NP; p++) {
                    const f32x4* P = reinterpret_cast<const f32x4*>(sD + ((c0 >> 1) + p) * PF);
                    const f32x4 G0 = P[0], G1 = P[1], G2 = P[2];   // (x, y), (conic.x, conic.z), (conic.y, opacity) of both
                    Gd[p] = P[3]; Ge[p] = P[4];
                    dx[p] = G0.xy - pxx; dy[p] = G0.zw - pyy;
                    const f32x2 pw2 = pair_power2(G1.xy, G2.xy, G1.zw, dx[p], dy[p]);
                    f32x2 a2;
                    {
#pragma clang fp contract(off)
                        a2 = G2.zw * exp_nonpos2(pw2);
                    }
                    pw[2 * p] = pw2.x; pw[2 * p + 1] = pw2.y;
                    al[2 * p] = fminf(0.99f, a2.x); al[2 * p + 1] = fminf(0.99f, a2.y);
                }
                // (slots beyond the batch hold opacity 0 -- pair_stage_store -- i.e. alpha 0: no bounds test per candidate)
                bool pre[KB];
#pragma unroll
                for (int k = 0; k < KB; k++) pre[k] = pw[k] <= 0.0f && al[k] >= (1.0f / 255.0f);
                // ---- (2) the sequential part: transmittance chain and cut-off (forward.cu:541-560) ----
                float w[KB];
#pragma unroll
                for (int k = 0; k < KB; k++) {
                    const bool live = pre[k] && !done;
                    const float test_T = T * (1.f - al[k]);
                    const bool stop = test_T < 0.0001f;   // (one comparison: `pass` and `term` are mask operations on it)
                    const bool pass = live && !stop;
                    done = done || (live && stop);
                    w[k] = pass ? al[k] * T : 0.f;
                    T = pass ? test_T : T;
                    // the last contributor as its position in the walk (wave-uniform operand); its slot in the tile list is looked up once, at the end
                    last_walk = pass ? head + (uint32_t)(c0 + k) + 1u : last_walk;
                }
                // ---- (3) accumulations (weight 0 for everything that did not pass) ----
                f32x2 wq[NP][4];   // svgss: bilinear corner weights x blend weight, per pair
#pragma unroll
                for (int p = 0; p < NP; p++) {
                    f32x2 dep = Gd[p].xy;
                    const f32x2 w2 = {w[2 * p], w[2 * p + 1]};
                    if (sp) {
                        dep -= dx[p] * Gd[p].zw + dy[p] * Ge[p].xy;   // depth differencing (common.hpp R_DA / R_DB)
                        if (SVGSS && VC > 0) {
                            const f32x4* P = reinterpret_cast<const f32x4*>(sD + ((c0 >> 1) + p) * PF);
                            const f32x4 G5 = P[5], G6 = P[6], G7 = P[7];   // (1/vmax, .), (J0, J1), (J2, J3)
                            const f32x2 du = dx[p] * G6.xy + dy[p] * G6.zw;
                            const f32x2 dv = dx[p] * G7.xy + dy[p] * G7.zw;
                            const f32x2 half = {0.5f, 0.5f}, one = {1.f, 1.f};
                            f32x2 u = du * Ge[p].zw * half + half, v = dv * G5.xy * half + half;
                            u.x = __builtin_amdgcn_fmed3f(u.x, 0.001f, 0.999f); u.y = __builtin_amdgcn_fmed3f(u.y, 0.001f, 0.999f);
                            v.x = __builtin_amdgcn_fmed3f(v.x, 0.001f, 0.999f); v.y = __builtin_amdgcn_fmed3f(v.y, 0.001f, 0.999f);
                            // pre-multiplied by the blend weight
                            wq[p][0] = (one - u) * (one - v) * w2; wq[p][1] = u * (one - v) * w2;
                            wq[p][2] = (one - u) * v * w2; wq[p][3] = u * v * w2;
                        }
                    }
                    D += dep.x * w[2 * p];
                    D += dep.y * w[2 * p + 1];
                }
                if (!MF) {
#pragma unroll
                    for (int k = 0; k < KB; k++) {
                        const f32x4* Cb = reinterpret_cast<const f32x4*>(sD + ((c0 + k) >> 1) * PF + PG::CH_OFF + (k & 1) * PG::CHP);
                        const f32x2 ww = {w[k], w[k]};
#pragma unroll
                        for (int q = 0; q < (PG::NCH + 3) / 4; q++) {
                            const f32x4 c4 = Cb[q];
                            accp[2 * q] = __builtin_elementwise_fma(c4.xy, ww, accp[2 * q]);
                            if (2 * q + 1 < NPAIR) accp[2 * q + 1] = __builtin_elementwise_fma(c4.zw, ww, accp[2 * q + 1]);
                        }
                    }
                } else {
                    // colour / normal / feature / vfeature sums on the matrix pipe: panel rows (lane = pixel) -> A operands
    #pragma unroll
                    for (int k = 0; k < KB; k++) sP[k * PS + lane] = w[k];
                    if (VC > 0) {
    #pragma unroll
                        for (int k = 0; k < KB; k++)
    #pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const float wj = (k & 1) ? wq[k >> 1][j].y : wq[k >> 1][j].x;
                                sP[(4 + k * 4 + j) * PS + lane] = sp ? wj : 0.f;
                            }
                    }
                    wave_lds_sync();
                    {   // channels: K = the 4 candidates of the group; B[k][n] = channel n of candidate k
                        const float bch = sD[((c0 >> 1) + (kq >> 1)) * PF + PG::CH_OFF + (kq & 1) * PG::CHP + nq];
    #pragma unroll
                        for (int mt = 0; mt < 4; mt++)
                            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(sP[kq * PS + 16 * mt + nq], bch, acc[mt], 0, 0, 0);
                    }
                    if (VC > 0) {   // vfeatures: one MFMA set per candidate, K = its 4 corners; B[k][n] = corner k of channel n
    #pragma unroll
                        for (int k = 0; k < KB; k++) {
                            const float bv = sD[((c0 + k) >> 1) * PF + PG::V_OFF + (k & 1) * PG::VB + kq * PG::VCP + nq];
    #pragma unroll
                            for (int mt = 0; mt < 4; mt++)
                                vacc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(sP[(4 + k * 4 + kq) * PS + 16 * mt + nq], bv, vacc[mt], 0, 0, 0);
                        }
                    }
                    wave_lds_sync();   // panel consumed before the next group overwrites it
                }
                // ---- (4) out_weights: the pixel sums of the KB = 4 weights, parked in LDS ----
                // ONE reduction tree for the four candidates: the first two levels (lane pairs, quads) exchange instead of
                // add-and-discard -- a lane keeps the candidate its low bits name and hands the other one over -- so that after them
                // lane l holds the quad's sum for candidate l & 3; two row shifts then leave each 16-lane row's sums in its lanes
                // 12..15, and the four rows meet when the batch's sums are flushed (13 instead of 28 instructions per group).
                {
                    const bool o1 = (lane & 1) != 0, o2 = (lane & 2) != 0;
                    const float t01 = (o1 ? w[1] : w[0]) + dpp_f32<0xB1>(o1 ? w[0] : w[1]);   // quad_perm [1,0,3,2]
                    const float t23 = (o1 ? w[3] : w[2]) + dpp_f32<0xB1>(o1 ? w[2] : w[3]);
                    float u = (o2 ? t23 : t01) + dpp_f32<0x4E>(o2 ? t01 : t23);                // quad_perm [2,3,0,1]
                    u += dpp_f32<0x114>(u);   // row_shr:4
                    u += dpp_f32<0x118>(u);   // row_shr:8
                    if ((lane & 12) == 12) w_cur[(lane >> 4) * CH + c0 + (lane & 3)] = u;
                }
                if (__all(done)) { wave_done = true; nproc = min(m, c0 + KB); break; }
            }
            nflush = nproc;
            head += (uint32_t)m;
            DEV_TRACE_MARK(2);   // blending
            // batches are CH-aligned and CH divides SEG: segment boundaries are batch ends
            if (!wave_done && (head & (uint32_t)(SEG - 1)) == 0u) dump_state(ndump++, false);
        }
        wave_lds_sync();
        flush_weights(b - 1);
    }
    // the live segments -- those that hold at least one consumed candidate -- are listed in tile order by seg_build_kernel
    // from these counts; the per-block totals it needs are summed here (fire-and-forget atomics, 4 T / 1024 counters)
    if (lane == 0 && !a.dump_only) {
        a.sub_count[sid] = head; a.sub_ndump[sid] = ndump;
        const uint32_t nseg = head != 0 ? min((head + (uint32_t)SEG - 1u) / (uint32_t)SEG, ndump + 1u) : 0u;
        if (nseg != 0) {   // nseg - 1 full segments + the last one, by its length class
            uint32_t* cnt = a.seg_block + (tile >> 8) * SEG_BLOCK_STRIDE;
            const int lc = seg_class(head - (nseg - 1u) * (uint32_t)SEG);
            if (nseg > 1u || lc == 0) atomicAdd(cnt, nseg - (lc == 0 ? 0u : 1u));
            if (lc != 0) atomicAdd(cnt + lc, 1u);
        }
    }
    if (head != 0 && ndump != 0) dump_state(ndump, true);   // final state (only needed by segments that do not start from the end)
    else gather_acc();

    if (inside && !a.dump_only) {
        const size_t N_ = (size_t)a.W * a.H;
        const size_t pid = (size_t)a.W * py + px;
        if (a.bg_in_render) write_zero_planes(a, pid, N_);
        T = fminf((float)(1 - 0.000001), T);
        a.final_T[pid] = T;
        a.n_contrib[pid] = last_walk ? (int32_t)(list[last_walk - 1u].y + 1u) : 0;   // (forward.cu:553: index in the TILE's list + 1)
        a.out_color[pid] = chan(0) + T * a.bg[0];
        a.out_color[N_ + pid] = chan(1) + T * a.bg[1];
        a.out_color[2 * N_ + pid] = chan(2) + T * a.bg[2];
#pragma unroll
        for (int ch = 0; ch < S; ch++) a.out_feature[ch * N_ + pid] = chan(6 + ch);
#pragma unroll
        for (int ch = 0; ch < VC; ch++) a.out_vfeature[ch * N_ + pid] = vchan(ch);
        a.out_normal[pid] = surface ? chan(3) : 0.f;
        a.out_normal[N_ + pid] = surface ? chan(4) : 0.f;
        a.out_normal[2 * N_ + pid] = surface ? chan(5) : 0.f;
        a.out_depth[pid] = normalize_depth ? D / (1.f - T) : D + T * 10.f;
        a.out_opacity[pid] = 1.f - T;
        a.final_D[pid] = D;
    }
    DEV_TRACE_MARK(3);   // dumps + epilogue
    DEV_TRACE_END(0, (unsigned)total, head, blockIdx.x);
}

// ---- contribution pre-pass of the fused shading -----------------------------------------------------------------------------
// Which surfels receive a blend weight at all?  The transmittance walk of render_fwd_kernel -- the same alpha (pair_power / exp_nonpos
// are the element-wise twins of the packed forms, stage.hpp / pairstage.hpp), the same alpha >= 1/255 and T < 1e-4 decisions in the same
// order, hence the same set bit for bit -- without anything else: no channels, no depth, no weights sum, no state; a candidate costs its
// 32-byte record header, ~35 instructions and one ballot.  One wave per 8x8 sub-tile in the composite's dispatch order; 64 candidates are
// staged per batch (lane = candidate: entry -> two 16-byte gathers, one batch ahead) and walked with wave-uniform LDS reads.
__global__ void __launch_bounds__(64) contrib_prepass_kernel(const RenderArgs a) {
    // pair-interleaved staging (as pairstage.hpp, geometry only): pair p of a batch = {x0 x1 y0 y1 | a0 a1 c0 c1 | b0 b1 o0 o1}, so that
    // one ds_read_b128 yields the aligned register pairs of TWO candidates and their alphas are packed fp32 instructions
    __shared__ __attribute__((aligned(16))) float sQ[2][32][12];
    __shared__ uint32_t sG[2][64];
    const uint32_t sid = a.sub_order[blockIdx.x];
    if (sid == ORDER_NONE) return;
    const int tile = (int)(sid >> 2), sub = (int)(sid & 3u);
    const uint32_t r0 = a.ranges[2 * tile], r1 = a.ranges[2 * tile + 1];
    const int len = (int)(r1 - r0);
    if (len == 0) return;
    const int total = (int)a.sub_total[sid];
    if (total == 0) return;
    const int lane = threadIdx.x;
    const int tx = tile % a.gx, ty = tile / a.gx;
    const int px = tx * TILE + (sub & 1) * 8 + (lane & 7), py = ty * TILE + (sub >> 1) * 8 + (lane >> 3);
    const float pxf = (float)px, pyf = (float)py;
    const f32x2 pxx = {pxf, pxf}, pyy = {pyf, pyf};
    const uint2* __restrict__ list = a.sub_list + (size_t)4 * r0 + (size_t)sub * len;
    const float4* __restrict__ rec4 = reinterpret_cast<const float4*>(a.rec);
    bool done = !(px < a.W && py < a.H);
    float T = 1.0f;
    const int nb = (total + 63) / 64;
    uint32_t gid = lane < total ? list[lane].x : 0u;
    float4 A = rec4[(size_t)gid * 6], B = rec4[(size_t)gid * 6 + 1];
    bool valid = lane < total;
    for (int b = 0; b < nb; b++) {
        const int buf = b & 1;
        {
            float* q = &sQ[buf][lane >> 1][lane & 1];
            q[0] = A.x; q[2] = A.y; q[4] = A.z; q[6] = B.x; q[8] = A.w; q[10] = valid ? B.y : 0.f;   // (opacity 0: a padding slot never blends)
            sG[buf][lane] = gid;
        }
        if (b + 1 < nb) {   // the next batch's gathers fly while this one is walked
            const int i = (b + 1) * 64 + lane;
            valid = i < total;
            gid = valid ? list[i].x : 0u;
            A = rec4[(size_t)gid * 6]; B = rec4[(size_t)gid * 6 + 1];
        }
        wave_lds_sync();
        const int m = min(64, total - b * 64);
        for (int c = 0; c < m; c += 2) {
            const f32x4* P = reinterpret_cast<const f32x4*>(&sQ[buf][c >> 1][0]);
            const f32x4 G0 = P[0], G1 = P[1], G2 = P[2];
            const f32x2 dx = G0.xy - pxx, dy = G0.zw - pyy;
            const f32x2 pw = pair_power2(G1.xy, G2.xy, G1.zw, dx, dy);
            f32x2 a2;
            {
#pragma clang fp contract(off)
                a2 = G2.zw * exp_nonpos2(pw);
            }
            const float al[2] = {fminf(0.99f, a2.x), fminf(0.99f, a2.y)};
            const float pwk[2] = {pw.x, pw.y};
            bool pass[2];
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const bool live = pwk[k] <= 0.0f && al[k] >= (1.0f / 255.0f) && !done;
                const float test_T = T * (1.f - al[k]);
                const bool term = live && test_T < 0.0001f;
                pass[k] = live && !term;
                done = done || term;
                T = pass[k] ? test_T : T;
            }
            // (uniform) some pixel blends the candidate: the composite will read its packed rows
            const bool any0 = __builtin_amdgcn_ballot_w64(pass[0]) != 0ull, any1 = __builtin_amdgcn_ballot_w64(pass[1]) != 0ull;
            if (any0 || any1) {
                if (lane < 2 && (lane == 0 ? any0 : any1)) a.needed[sG[buf][c + lane]] = 1;
            }
            if (__builtin_amdgcn_ballot_w64(!done) == 0ull) return;   // (uniform) every pixel is saturated
        }
        wave_lds_sync();   // (the buffer written two batches from now is this one)
    }
}

// ---- live backward segments, longest first (common.hpp SEG) ------------------------------------------------------
// One thread per tile (its four sub-tiles), 256 tiles per workgroup.  A sub-tile that consumed `count` candidates and dumped
// `ndump` states has min(ceil(count / SEG), ndump + 1) live segments: all full but possibly the last.  The forward has
// already summed them per workgroup and length class (seg_block), so every workgroup derives its own bases, scans its 256
// tiles and writes ids + descriptors: class-major, tile order inside a class.  Deterministic.
__global__ void __launch_bounds__(256) seg_build_kernel(const RenderArgs a, int T, int nblk, uint4* __restrict__ clear, size_t clear_n16,
                                                        const ShadeTables tabs, const float* __restrict__ weights, int P,
                                                        uint32_t* __restrict__ part_sums) {
    // piggy-backed: the backward's scratch clear (validity bytes / packed gradient rows), grid-stride over all workgroups -- one
    // launch in front of the composite backward instead of a memset + this kernel
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < clear_n16; i += (size_t)gridDim.x * 256) clear[i] = make_uint4(0u, 0u, 0u, 0u);
    // ... and, fused shading, the tables + the zeroed env-gradient accumulator the shading backward behind the composite needs
    if (tabs.env) {
        const int ne = tabs.entries();
        for (int i = (int)blockIdx.x * 256 + (int)threadIdx.x; i < ne; i += (int)gridDim.x * 256) shade_table_entry(tabs, i);
        for (int i = (int)blockIdx.x * 256 + (int)threadIdx.x; i < tabs.nzero; i += (int)gridDim.x * 256) tabs.zero[i] = 0.f;
    }
    // ... and the first half of the partition "surfels that received a blend weight" (subset.hip): per chunk of PART_ELEMS surfels the
    // number with weights > 0 (the scatter half is one small launch behind this one instead of two)
    if (weights) {
        __shared__ uint32_t pw[4];
        const int nch = (P + PART_ELEMS - 1) / PART_ELEMS;
        for (int ch = (int)blockIdx.x; ch < nch; ch += (int)gridDim.x) {
            const int base = ch * PART_ELEMS + (int)threadIdx.x * 8;
            uint32_t c = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) c += (base + i < P && weights[base + i] > 0.f) ? 1u : 0u;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) c += (uint32_t)__shfl_xor((int)c, d);
            __syncthreads();   // (pw of the previous chunk has been read)
            if ((threadIdx.x & 63) == 0) pw[threadIdx.x >> 6] = c;
            __syncthreads();
            if (threadIdx.x == 0) part_sums[ch] = pw[0] + pw[1] + pw[2] + pw[3];
        }
    }
    if ((int)blockIdx.x >= nblk) return;   // (workgroups beyond the tile blocks only clear)
    __shared__ uint32_t red_b[SEG_CLASSES][4], red_t[SEG_CLASSES][4], wfull[4];
    __shared__ unsigned long long wpart[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int tile = blockIdx.x * 256 + t;
    // per class: segments in the workgroups before this one, and in all of them
    uint32_t bef[SEG_CLASSES], tot[SEG_CLASSES];
#pragma unroll
    for (int c = 0; c < SEG_CLASSES; c++) bef[c] = tot[c] = 0u;
    for (int j = t; j < nblk; j += 256) {
#pragma unroll
        for (int c = 0; c < SEG_CLASSES; c++) {
            const uint32_t v = a.seg_block[j * SEG_BLOCK_STRIDE + c];
            tot[c] += v;
            bef[c] += j < (int)blockIdx.x ? v : 0u;
        }
    }
#pragma unroll
    for (int c = 0; c < SEG_CLASSES; c++) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { bef[c] += (uint32_t)__shfl_xor((int)bef[c], d); tot[c] += (uint32_t)__shfl_xor((int)tot[c], d); }
        if (lane == 0) { red_b[c][wave] = bef[c]; red_t[c][wave] = tot[c]; }
    }
    uint32_t nseg[4] = {0u, 0u, 0u, 0u}, head[4] = {0u, 0u, 0u, 0u}, nd[4] = {0u, 0u, 0u, 0u}, pb[4] = {0u, 0u, 0u, 0u}, sb[4] = {0u, 0u, 0u, 0u}, r0 = 0, r1 = 0;
    int lcls[4] = {0, 0, 0, 0};
    if (tile < T) {
        const uint4 c4 = reinterpret_cast<const uint4*>(a.sub_count)[tile], d4 = reinterpret_cast<const uint4*>(a.sub_ndump)[tile];
        const uint4 b4 = reinterpret_cast<const uint4*>(a.sub_pair_base)[tile], s4 = reinterpret_cast<const uint4*>(a.sub_slot_base)[tile];
        pb[0] = b4.x; pb[1] = b4.y; pb[2] = b4.z; pb[3] = b4.w;
        sb[0] = s4.x; sb[1] = s4.y; sb[2] = s4.z; sb[3] = s4.w;
        const uint2 rr = reinterpret_cast<const uint2*>(a.ranges)[tile];
        r0 = rr.x; r1 = rr.y;
        head[0] = c4.x; head[1] = c4.y; head[2] = c4.z; head[3] = c4.w;
        nd[0] = d4.x; nd[1] = d4.y; nd[2] = d4.z; nd[3] = d4.w;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            nseg[w] = head[w] != 0 ? min((head[w] + (uint32_t)SEG - 1u) / (uint32_t)SEG, nd[w] + 1u) : 0u;
            lcls[w] = nseg[w] != 0 ? seg_class(head[w] - (nseg[w] - 1u) * (uint32_t)SEG) : 0;
        }
    }
    // this thread's segments: full ones (class 0) and, packed 16 bits per class, its partial last segments (classes 1..4)
    uint32_t nfull = 0;
    unsigned long long npart = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        if (nseg[w] == 0) continue;
        nfull += nseg[w] - (lcls[w] == 0 ? 0u : 1u);
        if (lcls[w] != 0) npart += 1ull << (16 * (lcls[w] - 1));
    }
    uint32_t ifull = nfull;
    unsigned long long ipart = npart;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t of = (uint32_t)__shfl_up((int)ifull, d);
        const unsigned long long op = (unsigned long long)__shfl_up((long long)ipart, d);
        if (lane >= d) { ifull += of; ipart += op; }
    }
    if (lane == 63) { wfull[wave] = ifull; wpart[wave] = ipart; }
    __syncthreads();
    uint32_t base[SEG_CLASSES], n_all = 0;
#pragma unroll
    for (int c = 0; c < SEG_CLASSES; c++) {
        base[c] = n_all + (red_b[c][0] + red_b[c][1]) + (red_b[c][2] + red_b[c][3]);   // classes before + workgroups before
        n_all += (red_t[c][0] + red_t[c][1]) + (red_t[c][2] + red_t[c][3]);
    }
    uint32_t xfull = ifull - nfull;
    unsigned long long xpart = ipart - npart;
#pragma unroll
    for (int v = 0; v < 4; v++) { xfull += v < wave ? wfull[v] : 0u; xpart += v < wave ? wpart[v] : 0ull; }
    if (blockIdx.x == 0 && t == 0) a.seg_count[0] = n_all;
    if (nfull == 0 && npart == 0ull) return;
    uint32_t at[SEG_CLASSES];
    at[0] = base[0] + xfull;
#pragma unroll
    for (int c = 1; c < SEG_CLASSES; c++) at[c] = base[c] + (uint32_t)((xpart >> (16 * (c - 1))) & 0xffffull);
#pragma unroll
    for (int w = 0; w < 4; w++) {
        for (uint32_t k = 0; k < nseg[w]; k++) {
            const int c = (k + 1u == nseg[w]) ? lcls[w] : 0;
            uint32_t pos;
            if (c == 0) pos = at[0]++; else if (c == 1) pos = at[1]++; else if (c == 2) pos = at[2]++; else if (c == 3) pos = at[3]++; else pos = at[4]++;
            const uint32_t sm = ((uint32_t)(4 * tile + w) << SEG_K_BITS) | k;
            a.seg_list[pos] = sm;
            uint4* d = reinterpret_cast<uint4*>(a.seg_desc + pos);
            d[0] = make_uint4(sm, r0, r1 - r0, head[w]);
            d[1] = make_uint4(nd[w], pb[w], sb[w], 0u);
        }
    }
}

// The LDS request doubles as a residency control for experiments (build_variant.sh -DFWD_LDS_MIN=...).
#ifndef FWD_LDS_MIN
#define FWD_LDS_MIN 0
#endif
#ifndef FWD_WPE_HI
#define FWD_WPE_HI 3
#endif
template <int S, int VC, bool SVGSS>
void launch(const RenderArgs& a, hipStream_t s) {
    using PG = PairGeom<S, VC>;
    constexpr int W0 = StageGeom<S, VC>::WPE;
    constexpr int W1 = (VC >= 13 && FWD_WPE_HI > W0) ? FWD_WPE_HI : W0;   // (the callers' two svgss widths; everything else has one variant)
    const size_t lds = std::max(PG::lds_bytes(), (size_t)FWD_LDS_MIN);
    if (W1 != W0 && a.hi_fill) hipLaunchKernelGGL((render_fwd_kernel<S, VC, SVGSS, W1>), dim3(a.order_n), dim3(64), lds, s, a);
    else hipLaunchKernelGGL((render_fwd_kernel<S, VC, SVGSS, W0>), dim3(a.order_n), dim3(64), lds, s, a);
}

}  // namespace

void launch_seg_build(const RenderArgs& a, void* clear, size_t clear_bytes, const ShadeTables& tabs, const float* weights, int P, uint32_t* part_sums,
                      hipStream_t s) {
    const int T = a.gx * a.gy, nblk = (T + 255) / 256;
    const size_t n16 = clear ? (clear_bytes + 15) / 16 : 0;   // (the scratch regions are 256-byte aligned and padded: common.hpp align_up)
    int grid = (int)std::max<size_t>((size_t)nblk, std::min<size_t>((n16 + 255) / 256, 2048));
    if (weights) grid = std::max(grid, std::min((P + PART_ELEMS - 1) / PART_ELEMS, 2048));
    hipLaunchKernelGGL(seg_build_kernel, dim3(grid), dim3(256), 0, s, a, T, nblk, (uint4*)clear, n16, tabs, weights, P, part_sums);
}

void launch_contrib_prepass(const RenderArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(contrib_prepass_kernel, dim3(a.order_n), dim3(64), 0, s, a);
}

void launch_cull(const RenderArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(cull_kernel, dim3((unsigned)(order_entries(a.gx, a.gy) / 4)), dim3(CT), 0, s, a);
}

// Channel-count specialisations: the widths the reference's callers use (render.py:91 S=5; svgss.py:148-166
// train S=4,VS=52 / eval S=7,VS=64), the no-feature case, and small generic widths for tests.
int launch_render_fwd(const RenderArgs& a, bool svgss, hipStream_t s) {
    const int VC = a.VS / 4;
#define CASE(SV, VCV, SG) if (a.S == SV && VC == VCV && svgss == SG) { launch<SV, VCV, SG>(a, s); return 0; }
    CASE(0, 0, true) CASE(4, 13, true) CASE(7, 16, true) CASE(3, 2, true) CASE(1, 1, true) CASE(5, 0, true)
    CASE(0, 0, false) CASE(5, 0, false) CASE(3, 0, false) CASE(1, 0, false)
#undef CASE
    return -1;
}

bool render_specialised(int S, int VS, bool svgss) {
    const int VC = VS / 4;
#define CASE(SV, VCV, SG) if (S == SV && VC == VCV && svgss == SG) return true;
    CASE(0, 0, true) CASE(4, 13, true) CASE(7, 16, true) CASE(3, 2, true) CASE(1, 1, true) CASE(5, 0, true)
    CASE(0, 0, false) CASE(5, 0, false) CASE(3, 0, false) CASE(1, 0, false)
#undef CASE
    return false;
}

#if defined(SVGIR_DEV)
// development builds only: copies the per-wave records of kernel slot 0 (forward) / 1 (backward) and resets the slot
extern "C" int svgir_dev_trace_read(int slot, unsigned long long* out, int cap_records) {
    unsigned int n[2];
    if (hipMemcpyFromSymbol(n, HIP_SYMBOL(svgir::g_dev_trace_n), sizeof(n)) != hipSuccess) return -1;
    int cnt = (int)std::min<unsigned>(n[slot], (unsigned)std::min(cap_records, svgir::DEV_TRACE_CAP));
    if (cnt > 0 && hipMemcpyFromSymbol(out, HIP_SYMBOL(svgir::g_dev_trace), (size_t)cnt * svgir::DEV_TRACE_WORDS * 8,
                                       (size_t)slot * svgir::DEV_TRACE_CAP * svgir::DEV_TRACE_WORDS * 8) != hipSuccess) return -1;
    n[slot] = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(svgir::g_dev_trace_n), n, sizeof(n));
    return cnt;
}
#endif

}  // namespace svgir
