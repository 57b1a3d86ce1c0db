// svg-ir_amd/csrc/experiments/render_fwd_pc.hip -- NOT BUILT, NOT SHIPPED.  A measured-and-dropped experiment kept for
// the record ("producer/consumer forward"): on cfg2 it is slower than render_fwd.hip in all three forms tried
// (scalar-operand consumer 147 us; + entry windows via readlane 135 us; LDS-only consumer as below 103 us; render_fwd.hip
// 88 us), because the forward is bound by the SIMDs' total issue slots, not by its longest list, and two waves per
// sub-tile add instructions.  The last form also faults on one of the small parity cases; it was not debugged further.
// To try it again: add the file to SRCS, declare launch_render_fwd_plain in common.hpp and call it first in
// launch_render_fwd for VC == 0.
//
// The forward composite of a sub-tile is a strictly sequential walk of its candidate list (the T < 1e-4 cut-off makes
// everything behind it depend on the exact transmittance in front), the kernel lasts as long as its longest list, and ONE
// wave issues one instruction per ~8 cycles however independent the instructions are (scripts/probes/valu_rate_probe.hip).
// So the critical path is the instruction count per candidate of one wave -- and that is what this kernel cuts, by giving
// every 8x8 sub-tile TWO waves (one 128-thread workgroup) that split the per-candidate work by dependence, not by pixels:
//   * the PRODUCER wave evaluates everything that does not depend on the transmittance: pixel offsets, the conic form
//     (the reference's operation order, no contraction), the ~1 ulp exp, alpha with its three tests, and the depth
//     differencing term -- per candidate two floats per pixel, written to an LDS ring slot (lane = pixel);
//   * the CONSUMER wave runs only the transmittance chain (T' = T (1 - alpha), the 1e-4 cut-off, last contributor) and the
//     accumulation of depth, colour, normal and the S features.
// Same operations on the same values in the same order as a one-wave walk: bit-identical outputs, no speculation.  The two
// waves run on different SIMDs of the CU and meet at one workgroup barrier per batch of PB candidates (double-buffered ring);
// the consumer's "every pixel is done" flag stops both after the batch in flight.
// Per-candidate attributes are wave-uniform: both waves fetch theirs with SCALAR loads (constant address space, one candidate
// ahead) and use them as scalar operands -- no LDS staging, no broadcast ds_reads (render_bwd_plain.hip).
// out_weights: the blend weights of a batch go to an LDS panel and are summed lane-parallel once per batch (one atomic per
// candidate per batch).  State dumps for the depth-parallel backward and the list of live segments: as render_fwd.hip.
#include <algorithm>

#include "common.hpp"
#include "stage.hpp"

namespace svgir {

namespace {

typedef const __attribute__((address_space(4))) float cfloat;
typedef const __attribute__((address_space(4))) uint32_t cuint;
typedef const __attribute__((address_space(4))) char cchar;

constexpr int PB = 16;   // candidates per ring slot (SEG % PB == 0: segment boundaries are batch ends)
constexpr int PD = 2;    // scalar-load pipeline depth (candidates in flight)
static_assert(SEG % PB == 0, "segment boundaries must fall on batch boundaries");

template <int S, bool SVGSS>
__global__ void __launch_bounds__(128) render_fwd_plain_kernel(const RenderArgs a) {
    constexpr int SS = S > 0 ? S : 1, NCH = 6 + S;
    __shared__ float sAl[2][PB][64];   // alpha (0: the pixel does not blend this candidate)
    __shared__ float sDz[2][PB][64];   // dx DA + dy DB (depth differencing)
    __shared__ float sW[PB][64 + 4];   // blend weights of the current batch (out_weights)
    __shared__ __attribute__((aligned(16))) float sU[2][PB][16];   // per-candidate uniform data for the consumer: depth r g b | nx ny nz f0 | f1..f4 | slot gid - -
    __shared__ int sStop;

    if ((int)blockIdx.x >= 4 * a.gx * a.gy) return;
    const uint32_t sid = a.sub_order[blockIdx.x];
    const int tile = (int)(sid >> 2), sub = (int)(sid & 3u);
    const int tx = tile % a.gx, ty = tile / a.gx;
    const int lane = threadIdx.x & 63;
    const bool producer = threadIdx.x >= 64;   // (wave-uniform)
    const int px = tx * TILE + (sub & 1) * 8 + (lane & 7), py = ty * TILE + (sub >> 1) * 8 + (lane >> 3);
    const bool inside = px < a.W && py < a.H;
    const float pxf = (float)px, pyf = (float)py;
    const uint32_t r0 = a.ranges[2 * tile], r1 = a.ranges[2 * tile + 1];
    const int len = (int)(r1 - r0);
    if (len == 0) return;   // empty tile: the cull kernel has written its background pixels
    const int total = (int)a.sub_total[sid];
    const bool surface = cfg_flag(a.cfg, 0), normalize_depth = cfg_flag(a.cfg, 1);
    const bool sp = surface && cfg_flag(a.cfg, 2);
    cuint* list_c = (cuint*)(uintptr_t)(a.sub_list + (size_t)4 * r0 + (size_t)sub * len);   // {gid, slot} of candidate i at [2 i]
    cchar* rec_b = (cchar*)(uintptr_t)a.rec;
    cchar* feat_b = (cchar*)(uintptr_t)a.features;
    const int nb = (total + PB - 1) / PB;
    if (threadIdx.x == 0) sStop = 0;

    // ---------------------------------------------- producer ----------------------------------------------
    if (producer) {
        struct PCand { float X, Y, cxx, cxy, cyy, op, DA, DB; };
        // candidate ids of a 64-entry window that starts at the current batch: one coalesced vector load per batch (a batch
        // ahead), read lane-wise into SGPRs -- the record loads of candidate i + PD are issued while candidate i is evaluated
        auto load_ents = [&](int first) -> uint2 { return a.sub_list[(size_t)4 * r0 + (size_t)sub * len + (size_t)min(first + lane, max(total - 1, 0))]; };
        auto fetch = [&](uint32_t ids, int rel) -> PCand {
            const uint32_t gid = (uint32_t)__builtin_amdgcn_readlane((int)ids, rel);
            cfloat* r = (cfloat*)(rec_b + (uint32_t)(gid * (uint32_t)(REC * 4)));
            PCand c;
            c.X = r[R_X]; c.Y = r[R_Y]; c.cxx = r[R_CX]; c.cxy = r[R_CY]; c.cyy = r[R_CZ]; c.op = r[R_OP]; c.DA = r[R_DA]; c.DB = r[R_DB];
            return c;
        };
        uint2 ents = load_ents(0);
        uint32_t ids = ents.x;
        PCand ring[PD];
#pragma unroll
        for (int j = 0; j < PD; j++) ring[j] = fetch(ids, j);
        const float4* rec4 = reinterpret_cast<const float4*>(a.rec);
        auto fill = [&](int b) {
            float* al_row = &sAl[b & 1][0][lane];
            float* dz_row = &sDz[b & 1][0][lane];
            const uint2 ents_next = load_ents((b + 1) * PB);
            {   // the consumer's uniform data of this batch: lane = (candidate lane >> 2, piece lane & 3), vector gathers -> LDS
                const int c = lane >> 2, part = lane & 3;
                const uint32_t gid = (uint32_t)__shfl((int)ents.x, c), slot = (uint32_t)__shfl((int)ents.y, c);
                float* u = &sU[b & 1][c][0];
                if (part < 3) {
                    const float4 q = rec4[(size_t)gid * 6 + (part == 0 ? 1 : part == 1 ? 3 : 4)];
                    if (part == 0) { u[0] = q.z; u[12] = __builtin_bit_cast(float, slot); u[13] = __builtin_bit_cast(float, gid); }   // depth
                    else if (part == 1) { u[1] = q.y; u[2] = q.z; u[3] = q.w; }    // r g b
                    else { u[4] = q.x; u[5] = q.y; u[6] = q.z; }                    // normal
                } else {
#pragma unroll
                    for (int ch = 0; ch < S; ch++) u[7 + ch] = a.features[(size_t)gid * S + ch];
                }
            }
#pragma unroll 1
            for (int k = 0; k < PB; k += PD) {
#pragma unroll
                for (int j = 0; j < PD; j++) {
                    const PCand cur = ring[j];
                    ring[j] = fetch(ids, k + j + PD);   // (window of 64 entries from the batch's first: k + j + PD < 64)
                    const float dx = cur.X - pxf, dy = cur.Y - pyf;
                    const float pw = pair_power(cur.cxx, cur.cxy, cur.cyy, dx, dy);
                    float al;
                    {
#pragma clang fp contract(off)
                        al = cur.op * exp_nonpos(pw);
                    }
                    al = fminf(0.99f, al);
                    const bool pre = (b * PB + k + j < total) && pw <= 0.0f && al >= (1.0f / 255.0f);
                    al_row[(k + j) * 64] = pre ? al : 0.f;
                    dz_row[(k + j) * 64] = dx * cur.DA + dy * cur.DB;
                }
            }
            ents = ents_next; ids = ents.x;
        };
        if (total > 0) fill(0);
        __syncthreads();
        for (int b = 0; b < nb; b++) {
            if (b + 1 < nb) fill(b + 1);
            __syncthreads();
            if (sStop) break;
        }
        return;
    }

    // ---------------------------------------------- consumer ----------------------------------------------
    bool done = !inside;
    float T = 1.0f, D = 0.f;
    float acc[NCH];
#pragma unroll
    for (int i = 0; i < NCH; i++) acc[i] = 0.f;
    uint32_t last_contributor = 0;
    constexpr int NST = 8 + S;
    const uint32_t dump_base = seg_state_base(r0, (uint32_t)len, tile, sub);
    uint32_t ndump = 0;
    auto dump_state = [&](uint32_t j) {
        float* d = a.seg_state + ((size_t)(dump_base + j) * NST) * 64 + lane;
        d[0] = T; d[64] = acc[0]; d[128] = acc[1]; d[192] = acc[2];
        d[256] = surface ? acc[3] : 0.f; d[320] = surface ? acc[4] : 0.f; d[384] = surface ? acc[5] : 0.f; d[448] = D;
#pragma unroll
        for (int ch = 0; ch < S; ch++) d[(8 + ch) * 64] = acc[6 + ch];
    };
    const float spf = sp ? 1.f : 0.f;
    uint32_t head = 0;
    bool wave_done = __all(done);
    __syncthreads();   // batch 0 is in slot 0
    for (int b = 0; b < nb; b++) {
        if (wave_done) { if (lane == 0) sStop = 1; __syncthreads(); break; }
        const int m = min(PB, total - b * PB);
        const float* al_row = &sAl[b & 1][0][lane];
        const float* dz_row = &sDz[b & 1][0][lane];
        const float4* u_row = reinterpret_cast<const float4*>(&sU[b & 1][0][0]);
#pragma unroll 4
        for (int k = 0; k < PB; k++) {
            const float al = al_row[k * 64];
            const float dz = dz_row[k * 64];
            const float4 u0 = u_row[4 * k], u1 = u_row[4 * k + 1], u2 = u_row[4 * k + 2];   // (uniform address: broadcast reads)
            const uint32_t cslot = __builtin_bit_cast(uint32_t, sU[b & 1][k][12]);
            const float cdep = u0.x;
            const float cch[6] = {u0.y, u0.z, u0.w, u1.x, u1.y, u1.z};
            const float cf[5] = {u1.w, u2.x, u2.y, u2.z, u2.w};
            // the sequential part: transmittance chain and cut-off (forward.cu:541-560)
            const bool live = al > 0.f && !done;
            const float test_T = T * (1.f - al);
            const bool term = live && test_T < 0.0001f;
            const bool pass = live && !term;
            done = done || term;
            const float w = pass ? al * T : 0.f;
            T = pass ? test_T : T;
            last_contributor = pass ? cslot + 1u : last_contributor;
            const float dep = cdep - spf * dz;   // depth differencing (common.hpp R_DA / R_DB)
            D += dep * w;
#pragma unroll
            for (int i = 0; i < 6; i++) acc[i] += cch[i] * w;
#pragma unroll
            for (int i = 0; i < S; i++) acc[6 + i] += cf[i] * w;
            sW[k][lane] = w;
        }
        head += (uint32_t)m;
        wave_done = __all(done);
        if (wave_done && lane == 0) sStop = 1;
        // out_weights of the batch: 4 lanes per candidate sum 16 pixels each, two DPP adds, one atomic per candidate
        wave_lds_sync();
        {
            const int c = lane >> 2, part = lane & 3;
            const float4* src = reinterpret_cast<const float4*>(&sW[c][part * 16]);
            const float4 t0 = src[0], t1 = src[1], t2 = src[2], t3 = src[3];
            float v = ((t0.x + t0.y) + (t0.z + t0.w)) + ((t1.x + t1.y) + (t1.z + t1.w)) + (((t2.x + t2.y) + (t2.z + t2.w)) + ((t3.x + t3.y) + (t3.z + t3.w)));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
            if (part == 0 && c < m && v != 0.f) {
                const uint32_t gq = __builtin_bit_cast(uint32_t, sU[b & 1][c][13]);
                atomic_add_f32(&a.out_weights[gq], v);
            }
        }
        wave_lds_sync();
        if (!wave_done && (head & (uint32_t)(SEG - 1)) == 0u) dump_state(ndump++);
        __syncthreads();   // slot b & 1 is free for batch b + 2; batch b + 1 is complete
        if (wave_done) break;
    }
    // the live segments are listed in tile order by seg_build_kernel from these counts (render_fwd.hip)
    if (lane == 0) {
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
    if (inside) {
        const size_t N_ = (size_t)a.W * a.H;
        const size_t pid = (size_t)a.W * py + px;
        T = fminf((float)(1 - 0.000001), T);
        a.final_T[pid] = T;
        a.n_contrib[pid] = (int32_t)last_contributor;
        a.out_color[pid] = acc[0] + T * a.bg[0];
        a.out_color[N_ + pid] = acc[1] + T * a.bg[1];
        a.out_color[2 * N_ + pid] = acc[2] + T * a.bg[2];
#pragma unroll
        for (int ch = 0; ch < S; ch++) a.out_feature[ch * N_ + pid] = acc[6 + ch];
        a.out_normal[pid] = surface ? acc[3] : 0.f;
        a.out_normal[N_ + pid] = surface ? acc[4] : 0.f;
        a.out_normal[2 * N_ + pid] = surface ? acc[5] : 0.f;
        a.out_depth[pid] = normalize_depth ? D / (1.f - T) : D + T * 10.f;
        a.out_opacity[pid] = 1.f - T;
        a.final_D[pid] = D;
    }
}

template <int S, bool SVGSS>
void launch(const RenderArgs& a, hipStream_t s) {
    hipLaunchKernelGGL((render_fwd_plain_kernel<S, SVGSS>), dim3(4 * a.gx * a.gy), dim3(128), 0, s, a);
}

}  // namespace

int launch_render_fwd_plain(const RenderArgs& a, bool svgss, hipStream_t s) {
    if (a.VS != 0) return -1;
#define CASE(SV, SG) if (a.S == SV && svgss == SG) { launch<SV, SG>(a, s); return 0; }
    CASE(0, true) CASE(5, true) CASE(0, false) CASE(5, false) CASE(3, false) CASE(1, false)
#undef CASE
    return -1;
}

}  // namespace svgir
