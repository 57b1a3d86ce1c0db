// svg-ir_amd/csrc/render_bwd.hip -- backward alpha compositing.
//
// Replaces the backward renderCUDA (svgss backward.cu:529-934, rgss backward.cu:431-757): back-to-front replay
// of each pixel's blended splats (T <- T / (1 - alpha) starting from final_T), gradients of every blended quantity
// w.r.t. the per-Gaussian colour / feature / vfeature (x corner weight) / normal (x10, Q4) / depth, and through
// alpha to conic, mean2D (incl. the un-weighted depth-differencing term, Q5) and opacity.
//
// CDNA4 mapping
//   * one wave64 per depth segment of an 8x8-pixel sub-tile (one wave per workgroup, no barriers across waves): SEG
//     consecutive entries of the compact candidate list {Gaussian id, slot} that the forward kernel wrote for the
//     sub-tile (nothing is culled twice), walked in reverse.  The deepest live segment starts from final_T like the
//     reference; every other one starts from the forward state dumped at its far end (T there, and the blend of
//     everything behind = (final accumulators - prefix accumulators) / T), which turns the reference's strictly
//     sequential per-pixel replay into ~2x more, much shorter and evenly sized work items;
//   * candidates are staged CHB at a time into LDS like in the forward (stage.hpp); LDS per wave is kept small
//     (9 KB rgss / 18 KB svgss-train) because the kernel is latency-bound and lives off waves per SIMD;
//   * per-pixel replay state (running accumulators, last values, upstream gradients) lives in VGPRs thanks to
//     compile-time channel counts (the reference keeps ~330 floats per thread in scratch, backward.cu:617-635);
//   * gradient accumulation.  The reference issues 13 + S + VS global float atomics per (pixel, splat) pair
//     (18 / 69 / 84).  All per-Gaussian gradients except the 6 geometric ones have the form
//         dL/dq[g][ch] = sum over pixels of  a_v[pixel] * G[pixel][ch]
//     with a per-sub-tile-constant matrix G (upstream image gradients) and only 1 (+4 for the bilinear corners)
//     per-pair scalars a_v.  The replay is therefore split in two phases per sub-batch of SB candidates:
//       phase A (lane = pixel): replay, alpha gradient; the 1+4 per-pair scalars go to an LDS panel
//                [candidate][vector][pixel]; the 6 geometric gradients are pre-reduced over 8-lane octants with 3
//                DPP steps and stored as [candidate][6][8];
//       phase B: the contraction panel[rows][64 pixels] x G[64 pixels][channels] runs on the matrix pipe as
//                16 (+16 for the vfeature channels) v_mfma_f32_16x16x4_f32 per sub-batch -- exact fp32, G held in
//                registers in the B-operand layout -- and each lane issues the atomics of the (row, channel)
//                results it ends up holding.  (This is the one genuinely dense contraction of the path; the
//                blending itself stays scalar.)
#include "common.hpp"
#include "stage.hpp"

namespace svgir {

namespace {


// sum over the 8 lanes of an aligned octant (lanes differing in their low 3 bits); result in all 8 lanes
__device__ __forceinline__ float octant_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));  // row_half_mirror
    return v;
}

#ifdef RENDER_TIMING
__device__ unsigned long long g_bwd_tm[16];
#define TM_MARK(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tm_acc[i] += t_ - tm_prev; tm_prev = t_; } while (0)
#define TM_COUNT(i, n) tm_acc[i] += (unsigned long long)(n)
#ifdef RENDER_TIMING_FINE
#define TM_FINE(i) TM_MARK(i)
#else
#define TM_FINE(i)
#endif
#else
#define TM_FINE(i)
#define TM_MARK(i)
#define TM_COUNT(i, n)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifdef EXP_NO_ATOMICS   // timing experiment: keep the arithmetic alive, drop (almost) every atomic
#define BWD_ATOMIC(p, v) do { if ((v) == 123.456f) atomic_add_f32((p), (v)); } while (0)
#else
#define BWD_ATOMIC(p, v) atomic_add_f32((p), (v))
#endif

template <int S, int VC>
struct BwdGeom {
    using SG = StageGeom<S, VC>;
    static constexpr int CHB = (VC > 0) ? 16 : 8;        // candidates staged per batch (measured: 8 / 16 / 32 -> 203 / 212 /
                                                         // 227 us rgss cfg2, 426 / 415 / 488 us svgss cfg3)
    static constexpr int SB = (VC > 0) ? 4 : 8;          // candidates per phase-A/phase-B sub-batch
    static constexpr int NC0 = 7 + S;                    // colour3, normal3, depth, feature S  (<= 16)
    static constexpr int NG = NC0 + VC;                  // columns of G
    static constexpr int GROW = NG + 1;
    static constexpr int PROWS = (VC > 0) ? 16 : SB;     // panel rows: (candidate, corner) or candidate
    static constexpr int PS = 68;                        // panel row stride (floats): 16-byte aligned rows
#ifndef BWD_WPE_V
#define BWD_WPE_V 3
#define BWD_WPE_P 5
#endif
    // waves per SIMD the register budget is held to (measured: 5 rgss, 3 svgss-train; the eval widths need > 170 VGPRs)
    static constexpr int WPE = (VC > 13 || S + VC > 17) ? 2 : ((VC > 0) ? BWD_WPE_V : BWD_WPE_P);
    static constexpr size_t off_q = (size_t)CHB * SG::NF * 4;
    static constexpr size_t off_p = off_q + (size_t)SEG * 8;   // the whole segment's {gid, slot} entries
    static constexpr size_t off_pg = off_p + (size_t)PROWS * PS * 4;
    static constexpr size_t run_bytes = off_pg + (size_t)SB * 6 * 8 * 4;
    static constexpr size_t g_bytes = (size_t)64 * GROW * 4;   // G is only staged through LDS once, at setup
    static constexpr size_t lds_bytes = run_bytes > g_bytes ? run_bytes : g_bytes;
    static_assert(NC0 <= 16 && VC <= 16, "one 16-wide MFMA column tile per channel group");
};

template <int S, int VC, bool SVGSS>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(BwdGeom<S, VC>::WPE, BwdGeom<S, VC>::WPE)))
render_bwd_kernel(const RenderBwdArgs a) {
    constexpr int VS = VC * 4;
    using SG = StageGeom<S, VC>;
    using BG = BwdGeom<S, VC>;
    constexpr int SB = BG::SB, NC0 = BG::NC0, GROW = BG::GROW, CHB = BG::CHB, PS = BG::PS;
    constexpr int SS = S > 0 ? S : 1, VV = VC > 0 ? VC : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sD = reinterpret_cast<float*>(smem);                    // [CHB][NF] staged candidates
    uint2* sQ = reinterpret_cast<uint2*>(smem + BG::off_q);        // [SEG] {gid, slot} of the segment, deepest first
    float* sG = reinterpret_cast<float*>(smem);                    // [64][GROW] upstream gradients (setup only; aliases sD..)
    float* sP = reinterpret_cast<float*>(smem + BG::off_p);        // [PROWS][PS] blend-weight panel (MFMA A operand)
    float* sPg = reinterpret_cast<float*>(smem + BG::off_pg);      // [SB][6][8] octant-reduced geometric gradients

    // one wave per live depth segment (common.hpp SEG): seg_map[b] = (sub-tile id << SEG_K_BITS) | k
    const uint32_t sm = a.seg_map[blockIdx.x];
    if (sm == 0xFFFFFFFFu) return;
#ifdef RENDER_TIMING
    unsigned long long tm_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tm_prev = __builtin_amdgcn_s_memtime();
#endif
    const int sid = (int)(sm >> SEG_K_BITS), kseg = (int)(sm & ((1u << SEG_K_BITS) - 1u));
    const int tile = sid >> 2, sub = sid & 3;
    const int count = (int)a.sub_count[sid];
    const int ndump = (int)a.sub_ndump[sid];
    const int seg_lo = kseg * SEG, seg_hi = min(count, seg_lo + SEG);
    if (seg_hi <= seg_lo) return;
    const int tx = tile % a.gx, ty = tile / a.gx;
    const int lane = threadIdx.x;
    const int px = tx * TILE + (sub & 1) * 8 + (lane & 7);
    const int py = ty * TILE + (sub >> 1) * 8 + (lane >> 3);
    const bool inside = px < a.W && py < a.H;
    const float pxf = (float)px, pyf = (float)py;
    const uint32_t r0 = a.ranges[2 * tile], r1 = a.ranges[2 * tile + 1];
    const uint2* __restrict__ sub_in = a.sub_list + (size_t)4 * r0 + (size_t)sub * (r1 - r0);
    const bool surface = cfg_flag(a.cfg, 0), normalize_depth = cfg_flag(a.cfg, 1);
    const bool sp = surface && cfg_flag(a.cfg, 2);
    const bool bgeom = SVGSS ? true : (a.backward_geometry != 0);
    const size_t N_ = (size_t)a.W * a.H;
    const size_t pid = inside ? (size_t)a.W * py + px : 0;

    const float T_final = inside ? a.final_T[pid] : 0.f;
    const float D_final = (inside && normalize_depth) ? a.final_D[pid] : 0.f;
    const uint32_t last_contributor = inside ? (uint32_t)a.n_contrib[pid] : 0u;
    float gC[3], gN[3], gF[SS], gVF[VV], gD = 0.f, gO = 0.f;
#pragma unroll
    for (int i = 0; i < 3; i++) { gC[i] = inside ? a.g_color[i * N_ + pid] : 0.f; gN[i] = inside ? a.g_normal[i * N_ + pid] : 0.f; }
#pragma unroll
    for (int i = 0; i < SS; i++) gF[i] = (inside && i < S) ? a.g_feature[i * N_ + pid] : 0.f;
#pragma unroll
    for (int i = 0; i < VV; i++) gVF[i] = (inside && i < VC) ? a.g_vfeature[i * N_ + pid] : 0.f;
    if (inside) { gD = a.g_depth[pid]; gO = a.g_opacity[pid]; }
    const float bgdot = a.bg[0] * gC[0] + a.bg[1] * gC[1] + a.bg[2] * gC[2];
    const float ddelx_dx = 0.5f * a.W, ddely_dy = 0.5f * a.H;
    const float omt = 1.f - T_final;
    const float gDn = normalize_depth ? gD / omt : gD;  // depth gradient seen by the blended depth
    // d(depth normalisation)/d alpha of the reference, gD*D_final/(1-Tf)^2 * -Tf/(1-alpha)/T_new, equals kdn / T_old
    const float kdn = normalize_depth ? -gD * D_final * T_final / (omt * omt) : 0.f;
    const float kbg = bgdot + (normalize_depth ? 0.f : 10.f * gD);   // background (+ un-normalised depth) term
    const float q5 = sp ? -gD : 0.f;

    // deepest contributor of the wave
    uint32_t wmax = last_contributor;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, d));
    if (wmax == 0) return;

    // G matrix of this sub-tile: row = pixel (lane), columns = [colour3 | normal3 x10 (Q4) | depth | feature S | vfeature VC]
    {
        float* g = sG + lane * GROW;
        g[0] = gC[0]; g[1] = gC[1]; g[2] = gC[2];
        g[3] = surface ? gN[0] * 10.f : 0.f; g[4] = surface ? gN[1] * 10.f : 0.f; g[5] = surface ? gN[2] * 10.f : 0.f;
        g[6] = gDn;
#pragma unroll
        for (int i = 0; i < S; i++) g[7 + i] = gF[i];
#pragma unroll
        for (int i = 0; i < VC; i++) g[NC0 + i] = gVF[i];
    }

    // Phase B is the contraction  out[row][ch] = sum_pixel panel[row][pixel] * G[pixel][ch]  on the matrix pipe
    // (v_mfma_f32_16x16x4_f32, exact fp32).  G is transposed once through LDS into the B-operand layout and then
    // lives in registers: lane l holds G[pixel = 16*(l>>4) + kk][channel = l&15] for kk = 0..15, one set for the
    // plain channels and one for the vfeature channels.
    wave_lds_sync();
    const int colB = lane & 15, grpB = lane >> 4;
    float Bp[16], Bv[16];
#pragma unroll
    for (int kk = 0; kk < 16; kk++) {
        const float* g = sG + (16 * grpB + kk) * GROW;
        Bp[kk] = colB < NC0 ? g[colB] : 0.f;
        Bv[kk] = (VC > 0 && colB < VC) ? g[NC0 + (VC > 0 ? colB : 0)] : 0.f;
    }
    wave_lds_sync();   // sG is dead from here on (its LDS is reused by the staging buffers and the panel)
    // destination of the plain channel colB
    float* pbase = nullptr; int pstride = 0;
    if (colB < 3) { pbase = a.dL_dcolor + colB; pstride = 3; }
    else if (colB < 6) { pbase = a.dL_dnormal + (colB - 3); pstride = 3; }
    else if (colB < 7) { pbase = a.dL_ddepth; pstride = 1; }
    else if (colB < NC0) { pbase = a.dL_dfeature + (colB - 7); pstride = S; }
    // geometric channel owned by this lane in phase B: lane = (candidate lane>>3, value lane&7 < 6)
    float* gbase = nullptr; int gstride = 0;
    {
        const int j = lane & 7;
        if (j < 2) { gbase = a.dL_dmean2D + j; gstride = 3; }
        else if (j < 5) { gbase = a.dL_dconic + (j == 4 ? 3 : j - 2); gstride = 4; }
        else if (j < 6) { gbase = a.dL_dopacity; gstride = 1; }
    }

    // Replay state.  The reference keeps, per channel, the blend of everything behind the current splat (accum_rec)
    // and the last value, and adds (value - accum) * dL_dchannel to dL_dalpha for every channel.  The upstream
    // gradient of a pixel is the same for every splat, so the per-channel recurrences
    //     accum <- last_alpha * last + (1 - last_alpha) * accum
    // collapse into ONE scalar recurrence on A = sum_ch accum_ch * g_ch with s = sum_ch value_ch * g_ch:
    //     A <- last_alpha * s_last + (1 - last_alpha) * A ;   dL_dalpha += s - A.
    float T = T_final;
    float last_alpha = 0.f;
    float A_acc = 0.f, s_last = 0.f;
    if (kseg < ndump) {
        // Not the deepest live segment: start from the forward state dumped at this segment's far end.  With
        // last_alpha = 0 the recurrence takes accum = blend of everything behind = (final - prefix) / T_end.
        constexpr int NST = 8 + S + VC;
        const uint32_t base = blockIdx.x - (uint32_t)kseg;   // state slot of (sub-tile, 0)
        const float* e = a.seg_state + ((size_t)(base + kseg) * NST) * 64 + lane;
        const float* f = a.seg_state + ((size_t)(base + ndump) * NST) * 64 + lane;   // final state
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
#pragma unroll
        for (int i = 0; i < VC; i++) dot += (f[(8 + S + i) * 64] - e[(8 + S + i) * 64]) * gVF[i];
        A_acc = dot * __builtin_amdgcn_rcpf(T);
    }

    // The segment's list entries go to LDS once, deepest first (the replay walks back to front).
    const int nent = seg_hi - seg_lo;
    int nskip = 0;   // entries that lie behind every pixel of this wave (a prefix: slots descend)
    for (int i = lane; i < SEG; i += 64) {
        uint2 e = make_uint2(0u, 0u);
        if (i < nent) { e = sub_in[seg_hi - 1 - i]; sQ[i] = e; }
        nskip += __popcll(__ballot(i < nent && e.y >= wmax));
    }
    wave_lds_sync();
    TM_MARK(0);   // setup: upstream gradients, G matrix, start state
    // Batches of CHB candidates; the gathers of batch b+1 are in flight (registers) while batch b is replayed, so
    // neither their latency nor the completion of this batch's gradient atomics is waited for.
    StageRegs<S, VC, CHB> sr;
    int base = (nskip / CHB) * CHB;
    if (base < nent)
        stage_load<S, VC, CHB>(sr, min((int)CHB, nent - base), [&](int s) { return sQ[base + s].x; }, lane, a.rec,
                               a.features, a.vfeatures);
    for (; base < nent; base += CHB) {
        const int m = min((int)CHB, nent - base);
        wave_lds_sync();  // previous batch fully consumed
        stage_store<S, VC, CHB>(sr, sD, m, lane);
        {
            const int nb = base + CHB;
            if (nb < nent)
                stage_load<S, VC, CHB>(sr, min((int)CHB, nent - nb), [&](int s) { return sQ[nb + s].x; }, lane, a.rec,
                                       a.features, a.vfeatures);
        }
        wave_lds_sync();
        TM_MARK(1);   // staging (LDS stores, prefetch issue)
        TM_COUNT(5, m);

        for (int c0 = 0; c0 < m; c0 += SB) {
            const int nsub = min(SB, m - c0);
            uint32_t live = 0;  // bit cs set: candidate c0+cs has at least one blending pixel (wave-uniform)
            // ---------------- phase A: lane = pixel ----------------
            for (int cs = 0; cs < nsub; cs++) {
                const int c = c0 + cs;
                const float* r = sD + c * SG::NF;
                const float4* q = reinterpret_cast<const float4*>(r);
                // all LDS reads of the candidate are issued up front (one latency exposure)
                const uint32_t slot = sQ[base + c].y;
                const float4 A = q[0];   // x, y, conic.x, conic.y
                const float4 B = q[1];   // conic.z, opacity, depth, J6
                const float4 Jv = q[2];  // J0..J3
                const float4 E = q[3];   // J9, r, g, b
                const float4 Nn = q[4];  // nx, ny, nz, 1/umax
                const float ivm = q[5].x;
                float fl[SS];
#pragma unroll
                for (int ch = 0; ch < S; ch++) fl[ch] = r[SG::F_OFF + ch];
                if (slot >= wmax) continue;  // uniform: behind every pixel of this wave
                TM_FINE(8);    // loop overhead + slot read
                const float dx = A.x - pxf, dy = A.y - pyf;
                float power;
                if (SVGSS) power = -0.5f * ((A.z * dx * dx + B.x * dy * dy) + 2.f * A.w * dx * dy);
                else power = -0.5f * (A.z * dx * dx + B.x * dy * dy) - A.w * dx * dy;
                const float G = __expf(power);
                const float alpha = fminf(0.99f, B.y * G);
                const bool pass = slot < last_contributor && power <= 0.0f && alpha >= (1.0f / 255.0f);
                TM_FINE(9);    // alpha
                if (__ballot(pass) == 0ull) continue;
                live |= 1u << cs;
#ifdef EXP_NO_REPLAY
                if (live != 0xdeadbeefu) { T *= 0.999f; continue; }
#endif

                float vw = 0.f, vc0 = 0.f, vc1 = 0.f, vc2 = 0.f, vc3 = 0.f;
                float ge[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (pass) {
                    const float oma = 1.f - alpha;
                    const float inv_oma = __builtin_amdgcn_rcpf(oma);
                    const float inv_Told = __builtin_amdgcn_rcpf(T);
                    T = T * inv_oma;
                    const float dch = alpha * T;
                    const float inv_keep = 1.f - last_alpha;
                    float dL_dalpha = 0.f;
                    float du = 0.f, dv = 0.f;
                    float cw[4] = {0.f, 0.f, 0.f, 0.f};
                    if (sp) {
                        du = dx * Jv.x + dy * Jv.y; dv = dx * Jv.z + dy * Jv.w;
                        if (SVGSS && VC > 0) {
                            float u = du * Nn.w * 0.5f + 0.5f, v = dv * ivm * 0.5f + 0.5f;
                            u = fminf(0.999f, fmaxf(0.001f, u));
                            v = fminf(0.999f, fmaxf(0.001f, v));
                            cw[0] = (1.f - u) * (1.f - v); cw[1] = u * (1.f - v); cw[2] = (1.f - u) * v; cw[3] = u * v;
                        }
                    }
                    // s = sum over all blended channels of value * upstream gradient (see the replay-state comment)
                    float sdot = E.y * gC[0] + E.z * gC[1] + E.w * gC[2];
                    if (S > 0 && bgeom) {
#pragma unroll
                        for (int ch = 0; ch < S; ch++) sdot += fl[ch] * gF[ch];
                    }
                    if (VC > 0) {
                        // sum_ch (c4[ch] . cw) gVF[ch] = cw . (sum_ch c4[ch] gVF[ch])
                        const float4* vf = reinterpret_cast<const float4*>(r + SG::V_OFF);
                        float h0 = 0.f, h1 = 0.f, h2 = 0.f, h3 = 0.f;
#pragma unroll
                        for (int ch = 0; ch < VC; ch++) {
                            const float4 c4 = vf[ch];
                            h0 += c4.x * gVF[ch]; h1 += c4.y * gVF[ch]; h2 += c4.z * gVF[ch]; h3 += c4.w * gVF[ch];
                        }
                        sdot += (h0 * cw[0] + h1 * cw[1]) + (h2 * cw[2] + h3 * cw[3]);
                    }
                    if (surface) sdot += Nn.x * gN[0] + Nn.y * gN[1] + Nn.z * gN[2];
                    {  // depth
                        float d_cur = B.z;
                        if (sp) d_cur -= du * B.w + dv * E.x;
                        sdot += d_cur * gDn;
                    }
                    A_acc = last_alpha * s_last + inv_keep * A_acc;
                    s_last = sdot;
                    dL_dalpha += kdn * inv_Told + (sdot - A_acc);
                    dL_dalpha *= T;
                    const float tf_oma = T_final * inv_oma;
                    dL_dalpha += (gO - kbg) * tf_oma;
                    last_alpha = alpha;
                    const float dL_ddist = dL_dalpha * B.y * -0.5f * G;
                    vw = dch;
                    vc0 = cw[0] * dch; vc1 = cw[1] * dch; vc2 = cw[2] * dch; vc3 = cw[3] * dch;
                    ge[0] = dL_ddist * 2.f * (A.z * dx + A.w * dy) * ddelx_dx + q5 * (B.w * Jv.x + E.x * Jv.z);  // + Q5
                    ge[1] = dL_ddist * 2.f * (B.x * dy + A.w * dx) * ddely_dy + q5 * (B.w * Jv.y + E.x * Jv.w);
                    ge[2] = dL_ddist * (dx * dx);
                    ge[3] = dL_ddist * (dx * dy);
                    ge[4] = dL_ddist * (dy * dy);
                    ge[5] = G * dL_dalpha;
                }
                TM_FINE(10);   // replay math
                if (VC > 0) {   // rows (candidate, corner); the four corner weights sum to the blend weight
                    float* pr = sP + (cs * 4) * PS + lane;
                    pr[0] = sp ? vc0 : vw; pr[PS] = vc1; pr[2 * PS] = vc2; pr[3 * PS] = vc3;
                } else {
                    sP[cs * PS + lane] = vw;
                }
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    const float s8 = octant_sum(ge[k]);
                    if ((lane & 7) == 0) sPg[(cs * 6 + k) * 8 + (lane >> 3)] = s8;
                }
                TM_FINE(11);   // panel + octant writes
            }
            TM_MARK(2);   // phase A
            TM_COUNT(6, __popc(live));
#ifdef EXP_NO_PHASEB
            if (live != 0xdeadbeefu) continue;
#endif
            if (live == 0) continue;  // uniform
            wave_lds_sync();          // panel (and, the first time, G) visible to the phase-B lanes

            // ---------------- phase B: panel x G on the matrix pipe ----------------
            {
                const int rowA = VC > 0 ? colB : min(colB, SB - 1);
                const float4* ap = reinterpret_cast<const float4*>(sP + rowA * PS + 16 * grpB);
                const float4 a0 = ap[0], a1 = ap[1], a2 = ap[2], a3 = ap[3];
                const float av[16] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w,
                                      a2.x, a2.y, a2.z, a2.w, a3.x, a3.y, a3.z, a3.w};
                f32x4 accP = {0.f, 0.f, 0.f, 0.f}, accV = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 16; kk++) {
                    accP = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], Bp[kk], accP, 0, 0, 0);
                    if (VC > 0) accV = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], Bv[kk], accV, 0, 0, 0);
                }
                // D layout: lane l, register r -> row 4*(l>>4) + r, column l&15
                if (VC > 0 && a.grad_rows) {   // (rows are only used at the svgss widths, api.hip)
                    // ---- gradient rows (common.hpp GradRowGeom): plain stores, summed per Gaussian afterwards ----
                    constexpr int P4 = (NC0 + 3) / 4 * 4, GEO = P4 + VS, RS = (GEO + 6 + 3) / 4 * 4;
                    // slot of candidate cb of this sub-batch: 4 * (first instance of the Gaussian + index of this tile
                    // inside the Gaussian's tile rectangle, emit order) + sub-tile
                    auto slot_of = [&](int cb) -> size_t {
                        const float* rr = sD + (c0 + cb) * SG::NF;
                        const uint32_t ib = __builtin_bit_cast(uint32_t, rr[R_IBASE]);
                        const uint32_t rc = __builtin_bit_cast(uint32_t, rr[R_RECT]);
                        const uint32_t x0 = rc & 1023u, y0 = (rc >> 10) & 1023u, wr = rc >> 20;
                        return (size_t)4 * (ib + ((uint32_t)ty - y0) * wr + ((uint32_t)tx - x0)) + (uint32_t)sub;
                    };
                    if (VC > 0) {
                        const bool mine = grpB < nsub && ((live >> grpB) & 1u);
                        if (mine) {
                            float* row = a.grad_rows + slot_of(grpB) * RS;
                            if (colB < NC0) row[colB] = (accP[0] + accP[1]) + (accP[2] + accP[3]);
                            if (colB < VC)
                                reinterpret_cast<float4*>(row + P4)[colB] =
                                    sp ? make_float4(accV[0], accV[1], accV[2], accV[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int cB = 4 * grpB + r;
                            if (cB < nsub && ((live >> cB) & 1u) && colB < NC0) a.grad_rows[slot_of(cB) * RS + colB] = accP[r];
                        }
                    }
                    {
                        const int cB = lane >> 3, j = lane & 7;
                        if (cB < nsub && ((live >> cB) & 1u)) {
                            const size_t sl = slot_of(cB);
                            if (j < 6) {
                                float v = 0.f;
#pragma unroll
                                for (int p8 = 0; p8 < 8; p8++) v += sPg[(cB * 6 + j) * 8 + p8];
                                a.grad_rows[sl * RS + GEO + j] = v;
                            } else if (j == 6) {
                                a.row_flags[sl] = 1;
                            }
                        }
                    }
                } else if (VC > 0) {

                    // rows = (candidate grpB, corner r)
                    const bool mine = grpB < nsub && ((live >> grpB) & 1u);
                    const int gidB = mine ? (int)sQ[base + c0 + grpB].x : 0;
                    const float v = (accP[0] + accP[1]) + (accP[2] + accP[3]);
                    if (mine && colB < NC0 && v != 0.f) BWD_ATOMIC(pbase + (size_t)gidB * pstride, v);
                    if (mine && sp && colB < VC) {
                        float* dst = a.dL_dvfeature + (size_t)gidB * VS + colB * 4;
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            if (accV[r] != 0.f) BWD_ATOMIC(dst + r, accV[r]);
                    }
                } else {
                    // rows = candidates 4*grpB + r (only rows < SB exist)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int cB = 4 * grpB + r;
                        const bool mine = cB < nsub && ((live >> cB) & 1u);
                        if (mine && colB < NC0 && accP[r] != 0.f)
                            BWD_ATOMIC(pbase + (size_t)sQ[base + c0 + cB].x * pstride, accP[r]);
                    }
                }
                // --- geometric channels: finish the octant partials ---
                if (!(VC > 0 && a.grad_rows)) {
                    const int cB = lane >> 3, j = lane & 7;
                    const bool mine = cB < nsub && ((live >> cB) & 1u) && j < 6;
                    if (mine) {
                        float v = 0.f;
#pragma unroll
                        for (int p8 = 0; p8 < 8; p8++) v += sPg[(cB * 6 + j) * 8 + p8];
                        if (v != 0.f) BWD_ATOMIC(gbase + (size_t)sQ[base + c0 + cB].x * gstride, v);
                    }
                }
            }
            wave_lds_sync();  // panel consumed before the next phase A overwrites it
            TM_MARK(3);   // phase B
        }
    }
#ifdef RENDER_TIMING
    tm_acc[7] += 1;
    if (lane == 0) for (int i = 0; i < 16; i++) atomicAdd(&g_bwd_tm[i], tm_acc[i]);
#endif
}

template <int S, int VC, bool SVGSS>
void launch(const RenderBwdArgs& a, hipStream_t s) {
    using BG = BwdGeom<S, VC>;
    hipLaunchKernelGGL((render_bwd_kernel<S, VC, SVGSS>), dim3(a.seg_cap), dim3(64), BG::lds_bytes, s, a);
}

}  // namespace

int launch_render_bwd(const RenderBwdArgs& a, bool svgss, hipStream_t s) {
    const int VC = a.VS / 4;
#define CASE(SV, VCV, SG) if (a.S == SV && VC == VCV && svgss == SG) { launch<SV, VCV, SG>(a, s); return 0; }
    CASE(0, 0, true) CASE(4, 13, true) CASE(7, 16, true) CASE(3, 2, true) CASE(1, 1, true) CASE(5, 0, true)
    CASE(0, 0, false) CASE(5, 0, false) CASE(3, 0, false) CASE(1, 0, false)
#undef CASE
    return -1;
}

#ifdef RENDER_TIMING
extern "C" int svgir_debug_bwd_timing(unsigned long long* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(svgir::g_bwd_tm), 128) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(svgir::g_bwd_tm), z, 128); }
    return 0;
}
#endif

}  // namespace svgir
