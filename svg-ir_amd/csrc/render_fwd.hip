// svg-ir_amd/csrc/render_fwd.hip -- forward per-tile alpha compositing.
//
// Replaces renderCUDA (svgss forward.cu:401-750, rgss forward.cu:323-535): front-to-back blending of the
// depth-ordered splat list of each 16x16 tile; per-pixel depth by depth differencing; svgss additionally blends
// VS/4 "vfeature" channels, each the bilinear interpolation of 4 corner values in the surfel's tangent plane.
//
// CDNA4 mapping
//   * one 256-thread workgroup per tile = 4 wave64, each wave owns an 8x8 pixel quadrant (compact footprint =>
//     more wave-level culling than the reference's 16x2 warp rows);
//   * the splat list is consumed in batches; for each batch ALL per-splat data -- the 96-byte record written by the
//     preprocess stage plus the S feature and VS vfeature floats -- is gathered into LDS once per tile with 16-byte
//     loads issued by all 256 threads (one latency exposure per batch, every byte of the algorithmic gather
//     R*(4+G) is touched exactly once per tile).  The reference stages only the geometry and re-reads
//     features/vfeatures from global memory per (pixel, splat) (forward.cu:635-646);
//   * every wave walks the staged batch reading wave-uniform LDS addresses (broadcast ds_read_b128, no bank
//     conflicts).  Before that, the wave culls the batch LANE-PARALLEL: each lane tests one staged splat against the
//     wave's 8x8 pixel rectangle (exact minimum of the conic form over the rectangle vs. the 1/255 alpha threshold,
//     stage.hpp) and the ballot mask is then walked with scalar bit scans, so splats that cannot touch the wave
//     cost 1/64 of an iteration instead of one;
//   * channel counts are template parameters: accumulators live in VGPRs (the reference spills >640 floats of
//     per-thread arrays to scratch, forward.cu:483-493);
//   * the per-(pixel,splat) out_weights atomic of the reference (forward.cu:653) becomes one DPP wave reduction
//     + one atomic per (wave, splat).
#include "common.hpp"
#include "stage.hpp"

namespace svgir {

namespace {

template <int S, int VC, bool SVGSS>
__global__ void __launch_bounds__(BLOCK) render_fwd_kernel(const RenderArgs a) {
    using SG = StageGeom<S, VC>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sD = reinterpret_cast<float*>(smem);                               // [BATCH][NF]
    int* sId = reinterpret_cast<int*>(smem + (size_t)SG::BATCH * SG::NF * 4);  // [BATCH]

    const int tile = blockIdx.x;
    const int tx = tile % a.gx, ty = tile / a.gx;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int px = tx * TILE + (wave & 1) * 8 + (lane & 7);
    const int py = ty * TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < a.W && py < a.H;
    const float pxf = (float)px, pyf = (float)py;
    const float wx0 = (float)(tx * TILE + (wave & 1) * 8), wy0 = (float)(ty * TILE + (wave >> 1) * 8);
    const uint32_t r0 = a.ranges[2 * tile], r1 = a.ranges[2 * tile + 1];
    const bool surface = cfg_flag(a.cfg, 0), normalize_depth = cfg_flag(a.cfg, 1);
    const bool sp = surface && cfg_flag(a.cfg, 2);

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

    for (uint32_t base = r0; base < r1; base += SG::BATCH) {
        // all four waves finished => stop fetching (forward.cu:499-501); also the barrier that frees the LDS batch
        if (__syncthreads_and(done)) break;
        const int n = min((int)SG::BATCH, (int)(r1 - base));
        if (t < n) sId[t] = (int)a.point_list[base + t];
        __syncthreads();
        stage_batch<S, VC>(sD, sId, n, a.rec, a.features, a.vfeatures);
        __syncthreads();
        if (__all(done)) continue;  // this wave is finished; keep taking part in the barriers

        bool wave_done = false;
        for (int rnd = 0; rnd * 64 < n && !wave_done; rnd++) {
            // lane-parallel conservative cull of 64 staged splats against this wave's 8x8 pixel block
            const int js = rnd * 64 + lane;
            bool cand = false;
            if (js < n) {
                const float4* qs = reinterpret_cast<const float4*>(sD + js * SG::NF);
                const float4 A = qs[0];
                const float4 B = qs[1];
                cand = splat_may_touch(A.x, A.y, A.z, A.w, B.x, B.y, wx0, wy0, wx0 + 7.f, wy0 + 7.f);
            }
            unsigned long long mask = __ballot(cand);
            while (mask) {
                const int j = rnd * 64 + __builtin_ctzll(mask);
                mask &= mask - 1;
                const float4* q = reinterpret_cast<const float4*>(sD + j * SG::NF);
                const float4 A = q[0];   // x, y, conic.x, conic.y
                const float4 B = q[1];   // conic.z, opacity, depth, J6
                const float dx = A.x - pxf, dy = A.y - pyf;
                float power;
                if (SVGSS) power = -0.5f * ((A.z * dx * dx + B.x * dy * dy) + 2.f * A.w * dx * dy);
                else power = -0.5f * (A.z * dx * dx + B.x * dy * dy) - A.w * dx * dy;
                const float alpha = fminf(0.99f, B.y * __expf(power));
                bool pass = !done && power <= 0.0f && alpha >= (1.0f / 255.0f);
                const float test_T = T * (1.f - alpha);
                bool newly_done = false;
                if (pass && test_T < 0.0001f) { done = true; pass = false; newly_done = true; }
                if (__ballot(pass) == 0ull) {
                    if (__any(newly_done) && __all(done)) { wave_done = true; break; }
                    continue;
                }
                const float w = pass ? alpha * T : 0.f;
                const float4 J = q[2];   // J0..J3
                const float4 E = q[3];   // J9, r, g, b
                const float4 Nn = q[4];  // nx, ny, nz, 1/umax
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
                    const float* f = sD + j * SG::NF + SG::F_OFF;
#pragma unroll
                    for (int ch = 0; ch < S; ch++) F[ch] += f[ch] * w;
                }
                if (VC > 0) {
                    const float4* vf = reinterpret_cast<const float4*>(sD + j * SG::NF + SG::V_OFF);
#pragma unroll
                    for (int ch = 0; ch < VC; ch++) {
                        const float4 c4 = vf[ch];
                        VF[ch] += c4.x * w0 + c4.y * w1 + c4.z * w2 + c4.w * w3;
                    }
                }
                if (pass) {
                    T = test_T;
                    last_contributor = (base - r0) + (uint32_t)j + 1u;
                }
                const float wsum = wave_scan_last(w);
                if (lane == 63) atomic_add_f32(&a.out_weights[sId[j]], wsum);
                if (__any(newly_done) && __all(done)) { wave_done = true; break; }
            }
        }
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
    hipLaunchKernelGGL((render_fwd_kernel<S, VC, SVGSS>), dim3(a.gx * a.gy), dim3(BLOCK), SG::lds_bytes(), s, a);
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
