// svg-ir_amd/csrc/experiments/render_fwd_wide.hip -- NOT BUILT, NOT SHIPPED.  A measured-and-dropped experiment kept for the
// record ("candidates spread over lanes"): correct (parity suite green) but 136 us at cfg2 against 88 us for
// render_fwd.hip -- the per-group code came out at ~128 instructions + 19 per candidate of accumulation, i.e. ~170 issue slots
// per (sub-tile, candidate) against ~90, at 138 VGPRs (3 waves/SIMD): the shorter critical path does not pay for twice the work.
// To try it again: add the file to SRCS, declare launch_render_fwd_wide in common.hpp and call it first in launch_render_fwd
// for VC == 0.
//
// render_fwd.hip gives an 8x8 sub-tile one wave (lane = pixel) and walks its candidate list one candidate after the other:
// ~90 instruction issues per candidate, and the kernel lasts as long as its longest list at the ~8 cycles per instruction a
// lone wave gets.  Only the transmittance chain is sequential, though.  Here a sub-tile is one 256-thread
// workgroup: wave w owns pixel rows 2w, 2w+1 (16 pixels) and its lane = (pixel p, slot q) works on candidate 4g + q of group
// g: alpha, depth term, the colour / normal / feature products and the weight sums of FOUR candidates cost what one did --
//   * alpha of (pixel, candidate 4g + q): the reference's operations in the reference's order (stage.hpp pair_power,
//     exp_nonpos), one candidate per lane;
//   * the chain T' = T (1 - alpha), the 1e-4 cut-off and the last contributor stay strictly sequential and in the reference's
//     order: four steps per group, each fed by a quad broadcast (DPP) of slot j's alpha; every lane of a quad carries the
//     pixel's T and `done`;
//   * every lane accumulates ITS candidates' contributions; the four partial sums of a pixel are added (slot order 0..3)
//     at the state dumps and at the end.  Sums are therefore associated differently from a one-candidate-at-a-time walk
//     (rounding only; no threshold depends on them);
//   * out_weights: per wave a strided DPP sum over its 16 pixels, parked in LDS; the four waves' parts are added in order
//     and one atomic instruction per batch is issued (lane = candidate).
// The four waves meet at ONE barrier per batch of 16 candidates (staging, done flags and weight sums are double-buffered);
// the next batch's records are gathered by all 256 threads (one 16-byte piece each) while the current one is blended.
// State dumps, sub_count / sub_ndump and the per-block segment counters: exactly as render_fwd.hip (the backward cannot tell
// the two kernels apart).
#include <algorithm>

#include "common.hpp"
#include "stage.hpp"

namespace svgir {

namespace {

constexpr int WB = 16;    // candidates per batch
constexpr int WCF = 24;   // floats per staged candidate: x y cx cy | cz op depth DA | DB r g b | nx ny nz - | f0 f1 f2 f3 | f4 - - -
static_assert(SEG % WB == 0, "segment boundaries fall on batch boundaries");

__device__ __forceinline__ float quad_bcast(float v, int j) {   // value of slot j of the lane's quad
    const int x = __builtin_bit_cast(int, v);
    int r;
    switch (j) {
        case 0: r = __builtin_amdgcn_update_dpp(0, x, 0x00, 0xf, 0xf, false); break;
        case 1: r = __builtin_amdgcn_update_dpp(0, x, 0x55, 0xf, 0xf, false); break;
        case 2: r = __builtin_amdgcn_update_dpp(0, x, 0xAA, 0xf, 0xf, false); break;
        default: r = __builtin_amdgcn_update_dpp(0, x, 0xFF, 0xf, 0xf, false); break;
    }
    return __builtin_bit_cast(float, r);
}
__device__ __forceinline__ float quad_sum_ordered(float v) {   // ((s0 + s1) + s2) + s3, the same in all four lanes
    const float s0 = quad_bcast(v, 0), s1 = quad_bcast(v, 1), s2 = quad_bcast(v, 2), s3 = quad_bcast(v, 3);
    return ((s0 + s1) + s2) + s3;
}
__device__ __forceinline__ uint32_t quad_max_u32(uint32_t v) {
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    return v;
}
// sum over the 16 lanes of the wave that share (lane & 3); valid in lanes 0..3 (and all others)
__device__ __forceinline__ float slot_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));  // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

template <int S>
__global__ void __launch_bounds__(256) render_fwd_wide_kernel(const RenderArgs a) {
    constexpr int NCH = 6 + S;
    constexpr int NST = 8 + S;
    __shared__ __attribute__((aligned(16))) float sD[2][WB][WCF];
    __shared__ uint2 sQ[2][WB];
    __shared__ float sWs[2][4][WB];
    __shared__ int sDone[2][4];

    if ((int)blockIdx.x >= 4 * a.gx * a.gy) return;
    const uint32_t sid = a.sub_order[blockIdx.x];
    const int tile = (int)(sid >> 2), sub = (int)(sid & 3u);
    const int tx = tile % a.gx, ty = tile / a.gx;
    const uint32_t r0 = a.ranges[2 * tile], r1 = a.ranges[2 * tile + 1];
    const int len = (int)(r1 - r0);
    if (len == 0) return;   // empty tile: the cull kernel has written its background pixels
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int p = lane >> 2, q = lane & 3;
    const int px = tx * TILE + (sub & 1) * 8 + (p & 7), py = ty * TILE + (sub >> 1) * 8 + 2 * wave + (p >> 3);
    const int pi = 16 * wave + p;   // pixel index inside the sub-tile: the lane number of render_fwd.hip / the backward
    const bool inside = px < a.W && py < a.H;
    const float pxf = (float)px, pyf = (float)py;
    const int total = (int)a.sub_total[sid];
    const bool surface = cfg_flag(a.cfg, 0), normalize_depth = cfg_flag(a.cfg, 1);
    const bool sp = surface && cfg_flag(a.cfg, 2);
    const uint2* __restrict__ list = a.sub_list + (size_t)4 * r0 + (size_t)sub * len;
    const float4* __restrict__ rec4 = reinterpret_cast<const float4*>(a.rec);

    bool done = !inside;
    float T = 1.0f, D = 0.f;
    float acc[NCH];
#pragma unroll
    for (int i = 0; i < NCH; i++) acc[i] = 0.f;
    uint32_t last_contributor = 0;

    const uint32_t dump_base = seg_state_base(r0, (uint32_t)len, tile, sub);
    auto dump_state = [&](uint32_t j) {   // [state][channel][64 pixels]; lane q == 0 of every quad writes its pixel
        float ch[NCH];
#pragma unroll
        for (int i = 0; i < NCH; i++) ch[i] = quad_sum_ordered(acc[i]);
        const float Ds = quad_sum_ordered(D);
        if (q == 0) {
            float* d = a.seg_state + ((size_t)(dump_base + j) * NST) * 64 + pi;
            d[0] = T; d[64] = ch[0]; d[128] = ch[1]; d[192] = ch[2];
            // (the normal channels are blended unconditionally; without `surface` they do not exist for the consumers)
            d[256] = surface ? ch[3] : 0.f; d[320] = surface ? ch[4] : 0.f; d[384] = surface ? ch[5] : 0.f; d[448] = Ds;
#pragma unroll
            for (int c = 0; c < S; c++) d[(8 + c) * 64] = ch[6 + c];
        }
    };

    uint32_t head = 0, ndump = 0;   // candidates consumed so far; state dumps written (workgroup-uniform)
    if (total > 0) {
        const int nb = (total + WB - 1) / WB;
        // ---- staging: thread -> (candidate slot, piece).  Pieces 0..3 = the record's float4 #0, #1, #3, #4; threads 64..79 the
        // features; threads 80..95 the list entries.  Entries are fetched two batches ahead, records one batch ahead.
        const int sc = t < 64 ? (t >> 2) : (t & 15), piece = t < 64 ? (t & 3) : (t < 80 ? 4 : 5);
        const bool stager = t < 96;
        auto load_entry = [&](int b) -> uint2 {
            const int i = b * WB + sc;
            return (stager && i < total) ? list[i] : make_uint2(0u, 0u);
        };
        float4 st4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float stf[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        auto load_pieces = [&](int b, uint2 e) {   // the pieces of candidate b * WB + sc (zeros beyond the list: weight 0, finite values)
            const bool ok = stager && b * WB + sc < total;
            st4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 5; i++) stf[i] = 0.f;
            if (ok && piece < 4) st4 = rec4[(size_t)e.x * 6 + (piece == 0 ? 0 : piece == 1 ? 1 : piece == 2 ? 3 : 4)];
            if (ok && piece == 4) {
#pragma unroll
                for (int i = 0; i < S; i++) stf[i] = a.features[(size_t)e.x * S + i];
            }
        };
        auto store_pieces = [&](int buf, uint2 e) {
            if (!stager) return;
            if (piece < 4) *reinterpret_cast<float4*>(&sD[buf][sc][4 * piece]) = st4;
            else if (piece == 4) {
                *reinterpret_cast<float4*>(&sD[buf][sc][16]) = make_float4(stf[0], stf[1], stf[2], stf[3]);
                sD[buf][sc][20] = stf[4];
            } else sQ[buf][sc] = e;
        };
        uint2 e_cur = load_entry(0);
        load_pieces(0, e_cur);
        uint2 e_nxt = load_entry(1);
        store_pieces(0, e_cur);
        __syncthreads();

        for (int b = 0; b < nb; b++) {
            const int buf = b & 1;
            const int m = min(WB, total - b * WB);
            // the next batch's gathers are in flight while this one is blended
            const uint2 e_stage = e_nxt;
            if (b + 1 < nb) load_pieces(b + 1, e_stage);
            e_nxt = load_entry(b + 2);
            if (!__all(done)) {
#pragma unroll
                for (int g = 0; g < WB / 4; g++) {
                    const int c = 4 * g + q;
                    const float4* cd = reinterpret_cast<const float4*>(&sD[buf][c][0]);
                    const float4 A = cd[0], B = cd[1], C = cd[2], N4 = cd[3], F = cd[4];
                    const float f4 = sD[buf][c][20];
                    const uint32_t slot = sQ[buf][c].y;
                    const float dx = A.x - pxf, dy = A.y - pyf;
                    const float pw = pair_power(A.z, A.w, B.x, dx, dy);
                    float al;
                    {
#pragma clang fp contract(off)
                        al = B.y * exp_nonpos(pw);
                    }
                    al = fminf(0.99f, al);
                    const bool pre = c < m && pw <= 0.0f && al >= (1.0f / 255.0f);
                    const float al_s = pre ? al : -1.0f;
                    // ---- the sequential part, slot by slot (forward.cu:541-560) ----
                    float w = 0.f;
                    bool pass_mine = false;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const float alj = quad_bcast(al_s, j);
                        const bool live = alj >= 0.f && !done;
                        const float test_T = T * (1.f - alj);
                        const bool term = live && test_T < 0.0001f;
                        const bool pass = live && !term;
                        done = done || term;
                        const float wj = pass ? alj * T : 0.f;
                        T = pass ? test_T : T;
                        if (q == j) { w = wj; pass_mine = pass; }
                    }
                    last_contributor = max(last_contributor, quad_max_u32(pass_mine ? slot + 1u : 0u));   // (slots ascend along the list)
                    // ---- this lane's candidate ----
                    float dep = B.z;
                    if (sp) dep -= dx * B.w + dy * C.x;   // depth differencing (common.hpp R_DA / R_DB)
                    D += dep * w;
                    acc[0] += C.y * w; acc[1] += C.z * w; acc[2] += C.w * w;
                    acc[3] += N4.x * w; acc[4] += N4.y * w; acc[5] += N4.z * w;
                    const float fv[5] = {F.x, F.y, F.z, F.w, f4};
#pragma unroll
                    for (int i = 0; i < S; i++) acc[6 + i] += fv[i] * w;
                    const float ws = slot_sum(w);
                    if (lane < 4) sWs[buf][wave][4 * g + lane] = ws;
                }
            } else if (lane < WB) {
                sWs[buf][wave][lane] = 0.f;
            }
            if (b + 1 < nb) store_pieces(buf ^ 1, e_stage);
            const bool wave_all_done = __all(done);   // (evaluated by the whole wave, not inside the lane-0 branch)
            if (lane == 0) sDone[buf][wave] = wave_all_done ? 1 : 0;
            __syncthreads();
            const bool wg_done = (sDone[buf][0] & sDone[buf][1] & sDone[buf][2] & sDone[buf][3]) != 0;
            if (t >= 80 && t < 80 + m) {   // out_weights of this batch: one atomic instruction, lane = candidate.  (The threads that
                // will overwrite sQ[buf] two batches on are these very threads: program order keeps the read first.)
                const int c = t - 80;
                const float wsum = ((sWs[buf][0][c] + sWs[buf][1][c]) + sWs[buf][2][c]) + sWs[buf][3][c];
                if (wsum != 0.f) atomic_add_f32(&a.out_weights[sQ[buf][c].x], wsum);
            }
            head += (uint32_t)m;
            if (!wg_done && (head & (uint32_t)(SEG - 1)) == 0u) dump_state(ndump++);
            if (wg_done) break;
        }
    }
    // the live segments -- those that hold at least one consumed candidate -- are listed in tile order by seg_build_kernel
    // from these counts; the per-block totals it needs are summed here (render_fwd.hip)
    if (t == 0) {
        a.sub_count[sid] = head; a.sub_ndump[sid] = ndump;
        const uint32_t nseg = head != 0 ? min((head + (uint32_t)SEG - 1u) / (uint32_t)SEG, ndump + 1u) : 0u;
        if (nseg != 0) {
            uint32_t* cnt = a.seg_block + (tile >> 8) * SEG_BLOCK_STRIDE;
            const int lc = seg_class(head - (nseg - 1u) * (uint32_t)SEG);
            if (nseg > 1u || lc == 0) atomicAdd(cnt, nseg - (lc == 0 ? 0u : 1u));
            if (lc != 0) atomicAdd(cnt + lc, 1u);
        }
    }
    if (head != 0 && ndump != 0) dump_state(ndump);   // final state (only needed by segments that do not start from the end)
    float ch[NCH];
#pragma unroll
    for (int i = 0; i < NCH; i++) ch[i] = quad_sum_ordered(acc[i]);
    const float Ds = quad_sum_ordered(D);
    if (inside && q == 0) {
        const size_t N_ = (size_t)a.W * a.H;
        const size_t pid = (size_t)a.W * py + px;
        T = fminf((float)(1 - 0.000001), T);
        a.final_T[pid] = T;
        a.n_contrib[pid] = (int32_t)last_contributor;
        a.out_color[pid] = ch[0] + T * a.bg[0];
        a.out_color[N_ + pid] = ch[1] + T * a.bg[1];
        a.out_color[2 * N_ + pid] = ch[2] + T * a.bg[2];
#pragma unroll
        for (int c = 0; c < S; c++) a.out_feature[c * N_ + pid] = ch[6 + c];
        a.out_normal[pid] = surface ? ch[3] : 0.f;
        a.out_normal[N_ + pid] = surface ? ch[4] : 0.f;
        a.out_normal[2 * N_ + pid] = surface ? ch[5] : 0.f;
        a.out_depth[pid] = normalize_depth ? Ds / (1.f - T) : Ds + T * 10.f;
        a.out_opacity[pid] = 1.f - T;
        a.final_D[pid] = Ds;
    }
}

}  // namespace

// VS = 0 widths with a specialised kernel; < 0 otherwise
int launch_render_fwd_wide(const RenderArgs& a, hipStream_t s) {
#define CASE(SV) if (a.S == SV) { hipLaunchKernelGGL((render_fwd_wide_kernel<SV>), dim3(4 * a.gx * a.gy), dim3(256), 0, s, a); return 0; }
    CASE(0) CASE(1) CASE(3) CASE(5)
#undef CASE
    return -1;
}

}  // namespace svgir
