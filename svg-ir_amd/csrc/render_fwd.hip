// svg-ir_amd/csrc/render_fwd.hip -- forward per-tile alpha compositing.
//
// Replaces renderCUDA (svgss forward.cu:401-750, rgss forward.cu:323-535): front-to-back blending of the
// depth-ordered splat list of each 16x16 tile; per-pixel depth by depth differencing; svgss additionally blends
// VS/4 "vfeature" channels, each the bilinear interpolation of 4 corner values in the surfel's tangent plane.
//
// CDNA4 mapping
//   * one 256-thread workgroup per tile = 4 wave64, each wave owns an 8x8 pixel quadrant (compact footprint =>
//     more wave-level culling than the reference's 16x2 warp rows);
//   * per batch of 256 splats only the 24-byte "header" (mean2D, conic, opacity) + id is staged in LDS with one
//     coalesced-by-record gather per thread; every wave walks the batch reading headers as LDS broadcasts;
//   * a wave ballot decides whether ANY of its 64 pixels blends the splat; only then is the rest of the 96-byte
//     record and the S + VS feature floats fetched -- with wave-uniform addresses, i.e. scalar loads through
//     the constant cache into SGPRs, so the blend is `v_fmac vacc, s_feature, v_weight` with no LDS/VGPR staging;
//   * channel counts are template parameters: accumulators live in VGPRs (the reference spills >640 floats of
//     per-thread arrays to scratch, forward.cu:483-493);
//   * the per-(pixel,splat) out_weights atomic of the reference (forward.cu:653) becomes one DPP wave reduction
//     + one atomic per (wave, splat).
#include "common.hpp"

namespace svgir {

namespace {

template <int S, int VC, bool SVGSS>
__global__ void __launch_bounds__(BLOCK) render_fwd_kernel(const RenderArgs a) {
    __shared__ float4 sA[BLOCK];  // x, y, conic.x, conic.y
    __shared__ float2 sB[BLOCK];  // conic.z, opacity
    __shared__ int sId[BLOCK];

    const int tile = blockIdx.x;
    const int tx = tile % a.gx, ty = tile / a.gx;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int px = tx * TILE + (wave & 1) * 8 + (lane & 7);
    const int py = ty * TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < a.W && py < a.H;
    const float pxf = (float)px, pyf = (float)py;
    const uint32_t r0 = a.ranges[2 * tile], r1 = a.ranges[2 * tile + 1];
    const float* __restrict__ rec = a.rec;
    const float* __restrict__ feat = a.features;
    const float* __restrict__ vfeat = a.vfeatures;
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

    for (uint32_t base = r0; base < r1; base += BLOCK) {
        // all four waves finished => stop fetching (forward.cu:499-501); also the barrier that frees the LDS batch
        if (__syncthreads_and(done)) break;
        const int n = min((int)BLOCK, (int)(r1 - base));
        if (t < n) {
            const int id = (int)a.point_list[base + t];
            const float4* r = reinterpret_cast<const float4*>(rec + (size_t)id * REC);
            const float4 h0 = r[0];
            const float4 h1 = r[1];
            sA[t] = h0;
            sB[t] = make_float2(h1.x, h1.y);
            sId[t] = id;
        }
        __syncthreads();
        if (__all(done)) continue;  // this wave is finished; keep taking part in the barriers

        for (int j = 0; j < n; j++) {
            const float4 A = sA[j];
            const float2 B = sB[j];
            const float dx = A.x - pxf, dy = A.y - pyf;
            float power;
            if (SVGSS) power = -0.5f * ((A.z * dx * dx + B.x * dy * dy) + 2.f * A.w * dx * dy);
            else power = -0.5f * (A.z * dx * dx + B.x * dy * dy) - A.w * dx * dy;
            const float alpha = fminf(0.99f, B.y * __expf(power));
            bool pass = !done && power <= 0.0f && alpha >= (1.0f / 255.0f);
            const float test_T = T * (1.f - alpha);
            if (pass && test_T < 0.0001f) { done = true; pass = false; }
            if (__ballot(pass) == 0ull) {
                if (__all(done)) break;
                continue;
            }
            // wave-uniform fetch of the rest of the record
            const int gid = __builtin_amdgcn_readfirstlane(sId[j]);
            const float* __restrict__ r = rec + (size_t)gid * REC;
            const float w = pass ? alpha * T : 0.f;
            float dep = r[R_DEPTH];
            float w0 = 0.f, w1 = 0.f, w2 = 0.f, w3 = 0.f;
            if (sp) {
                const float du = dx * r[R_J0] + dy * r[R_J1];
                const float dv = dx * r[R_J2] + dy * r[R_J3];
                dep -= du * r[R_J6] + dv * r[R_J9];
                if (SVGSS && VC > 0) {
                    float u = du * r[R_IU] * 0.5f + 0.5f, v = dv * r[R_IV] * 0.5f + 0.5f;
                    u = fminf(0.999f, fmaxf(0.001f, u));
                    v = fminf(0.999f, fmaxf(0.001f, v));
                    // pre-multiplied by the blend weight
                    w0 = (1.f - u) * (1.f - v) * w; w1 = u * (1.f - v) * w; w2 = (1.f - u) * v * w; w3 = u * v * w;
                }
            }
            D += dep * w;
            C[0] += r[R_R] * w; C[1] += r[R_G] * w; C[2] += r[R_B] * w;
            if (surface) { N[0] += r[R_NX] * w; N[1] += r[R_NY] * w; N[2] += r[R_NZ] * w; }
            if (S > 0) {
                const float* __restrict__ f = feat + (size_t)gid * S;
#pragma unroll
                for (int ch = 0; ch < S; ch++) F[ch] += f[ch] * w;
            }
            if (VC > 0) {
                const float* __restrict__ vf = vfeat + (size_t)gid * (VC * 4);
#pragma unroll
                for (int ch = 0; ch < VC; ch++)
                    VF[ch] += vf[4 * ch] * w0 + vf[4 * ch + 1] * w1 + vf[4 * ch + 2] * w2 + vf[4 * ch + 3] * w3;
            }
            if (pass) {
                T = test_T;
                last_contributor = (base - r0) + (uint32_t)j + 1u;
            }
            const float wsum = wave_scan_last(w);
            if (lane == 63) atomic_add_f32(&a.out_weights[gid], wsum);
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
    hipLaunchKernelGGL((render_fwd_kernel<S, VC, SVGSS>), dim3(a.gx * a.gy), dim3(BLOCK), 0, s, a);
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
