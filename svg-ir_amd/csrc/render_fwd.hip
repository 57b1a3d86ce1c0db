// svg-ir_amd/csrc/render_fwd.hip -- forward alpha compositing.
//
// Replaces renderCUDA (svgss forward.cu:401-750, rgss forward.cu:323-535): front-to-back blending of the
// depth-ordered splat list of each 16x16 tile; per-pixel depth by depth differencing; svgss additionally blends
// VS/4 "vfeature" channels, each the bilinear interpolation of 4 corner values in the surfel's tangent plane.
//
// CDNA4 mapping (see stage.hpp for the staging details)
//   * one wave64 per 8x8-pixel sub-tile, one wave per workgroup: 4x more independent work items than the
//     reference's 256-thread tile blocks, no workgroup barriers, and a compact pixel footprint so that a splat is
//     only visited by the waves it can actually touch (measured: ~17 % of the (wave, splat) pairs of a tile list);
//   * the wave scans the tile list 64 entries per step, one splat per lane: coalesced id load, 32-byte header
//     gather, exact minimum of the conic form over the 8x8 rectangle against the 1/255 alpha threshold; survivors are
//     compacted with a ballot + popcount into an LDS ring and -- with their slot in the tile list -- into a global
//     per-sub-tile list that the backward kernel reuses (it never re-culls);
//   * candidates are staged CH at a time into LDS (record + features + vfeatures, 16-byte loads, all in flight
//     together) and consumed with wave-uniform broadcast reads; channel counts are template parameters so every
//     accumulator lives in a VGPR (the reference keeps >640 floats per thread in scratch, forward.cu:483-493);
//   * the per-(pixel,splat) out_weights atomic of the reference (forward.cu:653) becomes one DPP wave reduction
//     + one atomic per (wave, splat);
//   * after every SEG-th candidate (and once at the end) the wave dumps its blend state -- T and every accumulator,
//     [state][channel][64 pixels] -- and registers the live backward segments of its sub-tile (common.hpp SEG): the
//     backward is parallel over depth segments and starts each one from these states.
#include "common.hpp"
#include "stage.hpp"

namespace svgir {

namespace {


template <int S, int VC, bool SVGSS>
__global__ void __launch_bounds__(64) render_fwd_kernel(const RenderArgs a) {
    using SG = StageGeom<S, VC>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sD = reinterpret_cast<float*>(smem);                              // [CH][NF]
    uint2* sQ = reinterpret_cast<uint2*>(smem + (size_t)SG::CH * SG::NF * 4);  // [QN] {gid, slot}

    int tile, sub;
    sub_tile_of_block(blockIdx.x, a.gx * a.gy, a.tile_order, tile, sub);
    if (tile < 0) return;
    const int tx = tile % a.gx, ty = tile / a.gx;
    const int lane = threadIdx.x;
    const int bx = tx * TILE + (sub & 1) * 8, by = ty * TILE + (sub >> 1) * 8;
    const int px = bx + (lane & 7), py = by + (lane >> 3);
    const bool inside = px < a.W && py < a.H;
    const float pxf = (float)px, pyf = (float)py;
    const float wx0 = (float)bx, wy0 = (float)by;
    const uint32_t r0 = a.ranges[2 * tile], r1 = a.ranges[2 * tile + 1];
    const int len = (int)(r1 - r0);
    const bool surface = cfg_flag(a.cfg, 0), normalize_depth = cfg_flag(a.cfg, 1);
    const bool sp = surface && cfg_flag(a.cfg, 2);
    const float4* __restrict__ rec4 = reinterpret_cast<const float4*>(a.rec);
    uint2* __restrict__ sub_out = a.sub_list + (size_t)4 * r0 + (size_t)sub * len;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    bool done = !inside;
    float T = 1.0f, D = 0.f;
    float C[3] = {0.f, 0.f, 0.f}, N[3] = {0.f, 0.f, 0.f};
    float F[S > 0 ? S : 1];
    float VF[VC > 0 ? VC : 1];
#pragma unroll
    for (int i = 0; i < (S > 0 ? S : 1); i++) F[i] = 0.f;
#pragma unroll
    for (int i = 0; i < (VC > 0 ? VC : 1); i++) VF[i] = 0.f;
    uint32_t last_contributor = 0;

    // segment-boundary state dumps (see common.hpp SEG)
    constexpr int NST = 8 + S + VC;
    const uint32_t dump_base = (uint32_t)(((size_t)4 * r0 + (size_t)sub * len) / SEG) + (uint32_t)(4 * tile + sub);
    uint32_t ndump = 0;
    auto dump_state = [&](uint32_t j) {
        float* d = a.seg_state + ((size_t)(dump_base + j) * NST) * 64 + lane;
        d[0] = T; d[64] = C[0]; d[128] = C[1]; d[192] = C[2];
        d[256] = N[0]; d[320] = N[1]; d[384] = N[2]; d[448] = D;
#pragma unroll
        for (int ch = 0; ch < S; ch++) d[(8 + ch) * 64] = F[ch];
#pragma unroll
        for (int ch = 0; ch < VC; ch++) d[(8 + S + ch) * 64] = VF[ch];
    };

    uint32_t head = 0, tail = 0;  // candidate ring indices (wave-uniform)
    bool wave_done = __all(done);
    for (int scan = 0; scan < len && !wave_done; scan += 64) {
        // ---- lane-parallel cull of 64 list entries ----
        const int i = scan + lane;
        bool cand = false;
        uint32_t gid = 0;
        if (i < len) {
            gid = a.point_list[r0 + i];
            const float4 A = rec4[(size_t)gid * 6];
            const float4 B = rec4[(size_t)gid * 6 + 1];
            cand = splat_may_touch(A.x, A.y, A.z, A.w, B.x, B.y, wx0, wy0, wx0 + 7.f, wy0 + 7.f);
        }
        const unsigned long long mask = __ballot(cand);
        if (cand) {
            const uint32_t pos = tail + (uint32_t)__popcll(mask & lt_mask);
            const uint2 e = make_uint2(gid, (uint32_t)i);
            sQ[pos & (SG::QN - 1)] = e;
            sub_out[pos] = e;
        }
        tail += (uint32_t)__popcll(mask);
        const bool last_scan = scan + 64 >= len;

        // ---- stage + blend queued candidates, CH at a time ----
        while (!wave_done && (tail - head >= (uint32_t)SG::CH || (last_scan && tail != head))) {
            const int m = min((int)SG::CH, (int)(tail - head));
            wave_lds_sync();  // ring writes visible; previous batch fully consumed
            stage_candidates<S, VC, SG::CH>(sD, m, [&](int s) { return sQ[(head + s) & (SG::QN - 1)].x; }, lane, a.rec,
                                    a.features, a.vfeatures);
            wave_lds_sync();
            for (int c = 0; c < m; c++) {
                const float4* q = reinterpret_cast<const float4*>(sD + c * SG::NF);
                const float4 A = q[0];   // x, y, conic.x, conic.y
                const float4 B = q[1];   // conic.z, opacity, depth, J6
                const float4 J = q[2];   // J0..J3
                const float4 E = q[3];   // J9, r, g, b
                const float4 Nn = q[4];  // nx, ny, nz, 1/umax
                const float dx = A.x - pxf, dy = A.y - pyf;
                float power;
                if (SVGSS) power = -0.5f * ((A.z * dx * dx + B.x * dy * dy) + 2.f * A.w * dx * dy);
                else power = -0.5f * (A.z * dx * dx + B.x * dy * dy) - A.w * dx * dy;
                const float alpha = fminf(0.99f, B.y * __expf(power));
                bool pass = !done && power <= 0.0f && alpha >= (1.0f / 255.0f);
                const float test_T = T * (1.f - alpha);
                bool newly_done = false;
                if (pass && test_T < 0.0001f) { done = true; pass = false; newly_done = true; }
                if (__ballot(pass) != 0ull) {
                    const float w = pass ? alpha * T : 0.f;
                    // the wave reduction for out_weights is issued first: its dependent DPP steps (2 wait states
                    // each) interleave with the independent blend FMAs below instead of stalling at the end
                    const float wsum = wave_scan_last(w);
                    float dep = B.z;
                    float w0 = 0.f, w1 = 0.f, w2 = 0.f, w3 = 0.f;
                    if (sp) {
                        const float du = dx * J.x + dy * J.y;
                        const float dv = dx * J.z + dy * J.w;
                        dep -= du * B.w + dv * E.x;
                        if (SVGSS && VC > 0) {
                            const float iv = q[5].x;
                            float u = du * Nn.w * 0.5f + 0.5f, v = dv * iv * 0.5f + 0.5f;
                            u = fminf(0.999f, fmaxf(0.001f, u));
                            v = fminf(0.999f, fmaxf(0.001f, v));
                            // pre-multiplied by the blend weight
                            w0 = (1.f - u) * (1.f - v) * w; w1 = u * (1.f - v) * w; w2 = (1.f - u) * v * w; w3 = u * v * w;
                        }
                    }
                    D += dep * w;
                    C[0] += E.y * w; C[1] += E.z * w; C[2] += E.w * w;
                    if (surface) { N[0] += Nn.x * w; N[1] += Nn.y * w; N[2] += Nn.z * w; }
                    if (S > 0) {
                        const float* f = sD + c * SG::NF + SG::F_OFF;
#pragma unroll
                        for (int ch = 0; ch < S; ch++) F[ch] += f[ch] * w;
                    }
                    if (VC > 0) {
                        const float4* vf = reinterpret_cast<const float4*>(sD + c * SG::NF + SG::V_OFF);
#pragma unroll
                        for (int ch = 0; ch < VC; ch++) {
                            const float4 c4 = vf[ch];
                            VF[ch] += c4.x * w0 + c4.y * w1 + c4.z * w2 + c4.w * w3;
                        }
                    }
                    const uint2 e = sQ[(head + c) & (SG::QN - 1)];
                    if (pass) {
                        T = test_T;
                        last_contributor = e.y + 1u;
                    }
                    if (lane == 63) atomic_add_f32(&a.out_weights[e.x], wsum);
                }
                if (__any(newly_done) && __all(done)) { wave_done = true; break; }
            }
            head += (uint32_t)m;
            // batches are CH-aligned and CH divides SEG: segment boundaries are batch ends
            if (!wave_done && (head & (uint32_t)(SEG - 1)) == 0u) dump_state(ndump++);
        }
    }
    if (lane == 0) { a.sub_count[4 * tile + sub] = tail; a.sub_ndump[4 * tile + sub] = ndump; }
    if (tail != 0) {
        if (ndump != 0) dump_state(ndump);   // final state (only needed by segments that do not start from the end)
        // live segments: those that hold at least one processed candidate
        const uint32_t nseg = min((tail + (uint32_t)SEG - 1u) / (uint32_t)SEG, ndump + 1u);
        for (uint32_t k = lane; k < nseg; k += 64)
            a.seg_map[dump_base + k] = ((uint32_t)(4 * tile + sub) << SEG_K_BITS) | k;
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
#pragma unroll
        for (int ch = 0; ch < S; ch++) a.out_feature[ch * N_ + pid] = F[ch];
#pragma unroll
        for (int ch = 0; ch < VC; ch++) a.out_vfeature[ch * N_ + pid] = VF[ch];
        a.out_normal[pid] = surface ? N[0] : 0.f;
        a.out_normal[N_ + pid] = surface ? N[1] : 0.f;
        a.out_normal[2 * N_ + pid] = surface ? N[2] : 0.f;
        a.out_depth[pid] = normalize_depth ? D / (1.f - T) : D + T * 10.f;
        a.out_opacity[pid] = 1.f - T;
        a.final_D[pid] = D;
    }
}

template <int S, int VC, bool SVGSS>
void launch(const RenderArgs& a, hipStream_t s) {
    using SG = StageGeom<S, VC>;
    hipLaunchKernelGGL((render_fwd_kernel<S, VC, SVGSS>), dim3(sub_tile_grid(a.gx * a.gy)), dim3(64), SG::lds_bytes(),
                       s, a);
}

}  // namespace

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

}  // namespace svgir
