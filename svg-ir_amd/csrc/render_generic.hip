// svg-ir_amd/csrc/render_generic.hip -- composite kernels for channel widths WITHOUT a specialised kernel.
//
// The reference's renderCUDA kernels take the channel counts at run time (S <= 50 / 33, VS/4 <= 20: svgss forward.cu:483-
// 486, backward.cu:617-635); the specialised kernels of render_fwd.hip / render_bwd.hip are compiled for the widths the
// reference's callers use (register-resident accumulators, MFMA gradient contraction).  Every other width runs here:
// the same decomposition (one wave64 per 8x8 sub-tile walking the compact candidate list of the cull kernel), the same
// arithmetic, but run-time channel loops with the per-pixel feature accumulators / upstream gradients in LDS, no depth
// segments (one backward wave replays the whole list of its sub-tile from final_T, like the reference) and one wave
// reduction + atomic per (wave, splat, output).  Correctness path, not a fast path.
#include <algorithm>

#include "common.hpp"
#include "stage.hpp"

namespace svgir {

namespace {

struct PairGeom {   // everything of a (pixel, splat) pair that does not depend on the blend state
    float dx, dy, power, G, alpha, dep, cw[4];
    bool pre;
};

template <bool SVGSS>
__device__ __forceinline__ PairGeom pair_geom(const float4* __restrict__ q, float pxf, float pyf, bool sp, bool corners) {
    PairGeom g;
    const float4 A = q[0], B = q[1];
    g.dx = A.x - pxf; g.dy = A.y - pyf;
    g.power = pair_power(A.z, A.w, B.x, g.dx, g.dy);
    g.G = exp_nonpos(g.power);
    g.alpha = fminf(0.99f, B.y * g.G);
    g.pre = g.power <= 0.0f && g.alpha >= (1.0f / 255.0f);
    g.dep = B.z;
    g.cw[0] = g.cw[1] = g.cw[2] = g.cw[3] = 0.f;
    if (sp) {
        g.dep -= g.dx * B.w + g.dy * q[3].x;   // depth differencing (common.hpp R_DA / R_DB)
        if (SVGSS && corners) {
            const float4 J = q[2];
            const float du = g.dx * J.x + g.dy * J.y, dv = g.dx * J.z + g.dy * J.w;
            float u = du * q[4].w * 0.5f + 0.5f, v = dv * q[5].x * 0.5f + 0.5f;
            u = fminf(0.999f, fmaxf(0.001f, u));
            v = fminf(0.999f, fmaxf(0.001f, v));
            g.cw[0] = (1.f - u) * (1.f - v); g.cw[1] = u * (1.f - v); g.cw[2] = (1.f - u) * v; g.cw[3] = u * v;
        }
    }
    return g;
}

template <bool SVGSS>
__global__ void __launch_bounds__(64) render_fwd_generic_kernel(const RenderArgs a) {
    extern __shared__ __attribute__((aligned(16))) float acc[];   // [(S + VC)][64] feature / vfeature accumulators
    const int S = a.S, VC = a.VS / 4;
    if ((int)blockIdx.x >= 4 * a.gx * a.gy) return;
    const uint32_t sid = a.sub_order[blockIdx.x];
    const int tile = (int)(sid >> 2), sub = (int)(sid & 3u);
    const int tx = tile % a.gx, ty = tile / a.gx;
    const int lane = threadIdx.x;
    const int px = tx * TILE + (sub & 1) * 8 + (lane & 7), py = ty * TILE + (sub >> 1) * 8 + (lane >> 3);
    const bool inside = px < a.W && py < a.H;
    const float pxf = (float)px, pyf = (float)py;
    const uint32_t r0 = a.ranges[2 * tile], r1 = a.ranges[2 * tile + 1];
    const int len = (int)(r1 - r0);
    if (len == 0) return;   // empty tile: the cull kernel has written its background pixels
    const int total = (int)a.sub_total[sid];
    const bool surface = cfg_flag(a.cfg, 0), normalize_depth = cfg_flag(a.cfg, 1);
    const bool sp = surface && cfg_flag(a.cfg, 2);
    const uint2* __restrict__ list = a.sub_list + (size_t)4 * r0 + (size_t)sub * len;
    const float4* __restrict__ rec4 = reinterpret_cast<const float4*>(a.rec);
    for (int ch = 0; ch < S + VC; ch++) acc[ch * 64 + lane] = 0.f;
    bool done = !inside;
    float T = 1.0f, D = 0.f, C[3] = {0.f, 0.f, 0.f}, N[3] = {0.f, 0.f, 0.f};
    uint32_t last_contributor = 0;
    int i = 0;
    for (; i < total && !__all(done); i++) {
        const uint2 e = list[i];
        const float4* q = rec4 + (size_t)e.x * 6;
        const PairGeom g = pair_geom<SVGSS>(q, pxf, pyf, sp, VC > 0);
        const bool live = g.pre && !done;
        const float test_T = T * (1.f - g.alpha);
        const bool term = live && test_T < 0.0001f;
        const bool pass = live && !term;
        done = done || term;
        const float w = pass ? g.alpha * T : 0.f;
        T = pass ? test_T : T;
        last_contributor = pass ? e.y + 1u : last_contributor;
        const float4 E = q[3], Nn = q[4];
        D += g.dep * w;
        C[0] += E.y * w; C[1] += E.z * w; C[2] += E.w * w;
        if (surface) { N[0] += Nn.x * w; N[1] += Nn.y * w; N[2] += Nn.z * w; }
        for (int ch = 0; ch < S; ch++) acc[ch * 64 + lane] += a.features[(size_t)e.x * S + ch] * w;
        for (int ch = 0; ch < VC; ch++) {
            const float4 c4 = reinterpret_cast<const float4*>(a.vfeatures)[(size_t)e.x * VC + ch];
            acc[(S + ch) * 64 + lane] += w * (c4.x * g.cw[0] + c4.y * g.cw[1] + c4.z * g.cw[2] + c4.w * g.cw[3]);
        }
        const float wsum = wave_scan_last(w);
        if (lane == 63 && wsum != 0.f) atomic_add_f32(&a.out_weights[e.x], wsum);
    }
    // no depth segments: the backward replays the whole list from the end (one list entry per sub-tile)
    if (lane == 0) {
        a.sub_count[sid] = (uint32_t)i; a.sub_ndump[sid] = 0u;
        if (i != 0) atomicAdd(a.seg_block + (sid >> 10) * SEG_BLOCK_STRIDE + seg_class((uint32_t)i), 1u);   // one live segment (render_fwd.hip seg_build_kernel)
    }
    if (inside) {
        const size_t N_ = (size_t)a.W * a.H;
        const size_t pid = (size_t)a.W * py + px;
        T = fminf((float)(1 - 0.000001), T);
        a.final_T[pid] = T;
        a.n_contrib[pid] = (int32_t)last_contributor;
        a.out_color[pid] = C[0] + T * a.bg[0];
        a.out_color[N_ + pid] = C[1] + T * a.bg[1];
        a.out_color[2 * N_ + pid] = C[2] + T * a.bg[2];
        for (int ch = 0; ch < S; ch++) a.out_feature[ch * N_ + pid] = acc[ch * 64 + lane];
        for (int ch = 0; ch < VC; ch++) a.out_vfeature[ch * N_ + pid] = acc[(S + ch) * 64 + lane];
        a.out_normal[pid] = surface ? N[0] : 0.f;
        a.out_normal[N_ + pid] = surface ? N[1] : 0.f;
        a.out_normal[2 * N_ + pid] = surface ? N[2] : 0.f;
        a.out_depth[pid] = normalize_depth ? D / (1.f - T) : D + T * 10.f;
        a.out_opacity[pid] = 1.f - T;
        a.final_D[pid] = D;
    }
}

template <bool SVGSS>
__global__ void __launch_bounds__(64) render_bwd_generic_kernel(const RenderBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float gbuf[];   // [(S + VC)][64] upstream feature / vfeature gradients
    const int S = a.S, VC = a.VS / 4, VS = a.VS;
    const int lane = threadIdx.x;
    const bool surface = cfg_flag(a.cfg, 0), normalize_depth = cfg_flag(a.cfg, 1);
    const bool sp = surface && cfg_flag(a.cfg, 2);
    const bool bgeom = SVGSS ? true : (a.backward_geometry != 0);
    const size_t N_ = (size_t)a.W * a.H;
    const float ddelx_dx = 0.5f * a.W, ddely_dy = 0.5f * a.H;
    const float4* __restrict__ rec4 = reinterpret_cast<const float4*>(a.rec);
    const uint32_t nlive = min(a.seg_count[0], (uint32_t)a.seg_cap);
    for (uint32_t item = blockIdx.x; item < nlive; item += gridDim.x) {
        wave_lds_sync();
        const int sid = (int)(a.seg_list[item] >> SEG_K_BITS);
        const int tile = sid >> 2, sub = sid & 3;
        const int count = (int)a.sub_count[sid];
        const int tx = tile % a.gx, ty = tile / a.gx;
        const int px = tx * TILE + (sub & 1) * 8 + (lane & 7), py = ty * TILE + (sub >> 1) * 8 + (lane >> 3);
        const bool inside = px < a.W && py < a.H;
        const float pxf = (float)px, pyf = (float)py;
        const uint32_t r0 = a.ranges[2 * tile], r1 = a.ranges[2 * tile + 1];
        const uint2* __restrict__ list = a.sub_list + (size_t)4 * r0 + (size_t)sub * (r1 - r0);
        const size_t pid = inside ? (size_t)a.W * py + px : 0;
        const float T_final = inside ? a.final_T[pid] : 0.f;
        const float D_final = (inside && normalize_depth) ? a.final_D[pid] : 0.f;
        const uint32_t last_contributor = inside ? (uint32_t)a.n_contrib[pid] : 0u;
        float gC[3], gN[3], gD = 0.f, gO = 0.f;
        for (int c = 0; c < 3; c++) { gC[c] = (inside && a.g_color) ? a.g_color[c * N_ + pid] : 0.f; gN[c] = (inside && a.g_normal) ? a.g_normal[c * N_ + pid] : 0.f; }
        if (inside) { gD = a.g_depth ? a.g_depth[pid] : 0.f; gO = a.g_opacity ? a.g_opacity[pid] : 0.f; }
        for (int ch = 0; ch < S; ch++) gbuf[ch * 64 + lane] = (inside && a.g_feature) ? a.g_feature[ch * N_ + pid] : 0.f;
        for (int ch = 0; ch < VC; ch++) gbuf[(S + ch) * 64 + lane] = (inside && a.g_vfeature) ? a.g_vfeature[ch * N_ + pid] : 0.f;
        const float bgdot = a.bg[0] * gC[0] + a.bg[1] * gC[1] + a.bg[2] * gC[2];
        const float omt = 1.f - T_final;
        const float gDn = normalize_depth ? gD / omt : gD;
        const float kdn = normalize_depth ? -gD * D_final * T_final / (omt * omt) : 0.f;
        const float gO_kbg = gO - (bgdot + (normalize_depth ? 0.f : 10.f * gD));
        const float q5g = sp ? -gD : 0.f;
        float T = T_final, last_alpha = 0.f, A_acc = 0.f, s_last = 0.f;
        auto reduce_to = [&](float v, float* dst) {
            const float t = wave_scan_last(v);
            if (lane == 63 && t != 0.f) atomic_add_f32(dst, t);
        };
        for (int i = count - 1; i >= 0; i--) {
            const uint2 e = list[i];
            const float4* q = rec4 + (size_t)e.x * 6;
            const PairGeom g = pair_geom<SVGSS>(q, pxf, pyf, sp, VC > 0);
            const bool pass = g.pre && e.y < last_contributor;
            if (__ballot(pass) == 0ull) continue;   // uniform
            const float4 A = q[0], B = q[1], E = q[3], Nn = q[4];
            // s = sum over all blended channels of value * upstream gradient (render_bwd.hip: the scalar replay recurrence)
            float sdot = E.y * gC[0] + E.z * gC[1] + E.w * gC[2] + g.dep * gDn;
            if (surface) sdot += Nn.x * gN[0] + Nn.y * gN[1] + Nn.z * gN[2];
            if (bgeom) for (int ch = 0; ch < S; ch++) sdot += a.features[(size_t)e.x * S + ch] * gbuf[ch * 64 + lane];
            for (int ch = 0; ch < VC; ch++) {
                const float4 c4 = reinterpret_cast<const float4*>(a.vfeatures)[(size_t)e.x * VC + ch];
                sdot += (c4.x * g.cw[0] + c4.y * g.cw[1] + c4.z * g.cw[2] + c4.w * g.cw[3]) * gbuf[(S + ch) * 64 + lane];
            }
            const float ioma = __builtin_amdgcn_rcpf(1.f - g.alpha);
            const float inv_Told = __builtin_amdgcn_rcpf(T);
            const float Tn = T * ioma;
            const float An = last_alpha * s_last + (1.f - last_alpha) * A_acc;
            float dL_dalpha = (kdn * inv_Told + (sdot - An)) * Tn + gO_kbg * (T_final * ioma);
            T = pass ? Tn : T; A_acc = pass ? An : A_acc; s_last = pass ? sdot : s_last; last_alpha = pass ? g.alpha : last_alpha;
            dL_dalpha = pass ? dL_dalpha : 0.f;
            const float vw = pass ? g.alpha * Tn : 0.f;
            const float dL_ddist = dL_dalpha * (B.y * -0.5f) * g.G;
            float ge0 = dL_ddist * 2.f * (A.z * g.dx + A.w * g.dy) * ddelx_dx, ge1 = dL_ddist * 2.f * (B.x * g.dy + A.w * g.dx) * ddely_dy;
            if (sp) { ge0 += pass ? q5g * B.w : 0.f; ge1 += pass ? q5g * E.x : 0.f; }
            const size_t gi = e.x;
            reduce_to(pass ? ge0 : 0.f, a.dL_dmean2D + gi * 3);
            reduce_to(pass ? ge1 : 0.f, a.dL_dmean2D + gi * 3 + 1);
            reduce_to(pass ? dL_ddist * (g.dx * g.dx) : 0.f, a.dL_dconic + gi * 4);
            reduce_to(pass ? dL_ddist * (g.dx * g.dy) : 0.f, a.dL_dconic + gi * 4 + 1);
            reduce_to(pass ? dL_ddist * (g.dy * g.dy) : 0.f, a.dL_dconic + gi * 4 + 3);
            reduce_to(pass ? g.G * dL_dalpha : 0.f, a.dL_dopacity + gi);
            for (int c = 0; c < 3; c++) reduce_to(vw * gC[c], a.dL_dcolor + gi * 3 + c);
            if (surface) for (int c = 0; c < 3; c++) reduce_to(vw * gN[c] * 10.f, a.dL_dnormal + gi * 3 + c);   // Q4
            reduce_to(vw * gDn, a.dL_ddepth + gi);
            for (int ch = 0; ch < S; ch++) reduce_to(vw * gbuf[ch * 64 + lane], a.dL_dfeature + gi * S + ch);
            if (sp) {
                for (int ch = 0; ch < VC; ch++) {
                    const float gv = vw * gbuf[(S + ch) * 64 + lane];
                    for (int k = 0; k < 4; k++) reduce_to(gv * g.cw[k], a.dL_dvfeature + gi * VS + 4 * ch + k);
                }
            }
        }
    }
}

}  // namespace

void launch_render_fwd_generic(const RenderArgs& a, bool svgss, hipStream_t s) {
    const size_t lds = (size_t)std::max(1, a.S + a.VS / 4) * 64 * 4;
    if (svgss) hipLaunchKernelGGL(render_fwd_generic_kernel<true>, dim3(4 * a.gx * a.gy), dim3(64), lds, s, a);
    else hipLaunchKernelGGL(render_fwd_generic_kernel<false>, dim3(4 * a.gx * a.gy), dim3(64), lds, s, a);
}

void launch_render_bwd_generic(const RenderBwdArgs& a, bool svgss, hipStream_t s) {
    const size_t lds = (size_t)std::max(1, a.S + a.VS / 4) * 64 * 4;
    const int grid = std::min(a.seg_cap, 4 * a.gx * a.gy);
    if (svgss) hipLaunchKernelGGL(render_bwd_generic_kernel<true>, dim3(grid), dim3(64), lds, s, a);
    else hipLaunchKernelGGL(render_bwd_generic_kernel<false>, dim3(grid), dim3(64), lds, s, a);
}

}  // namespace svgir
