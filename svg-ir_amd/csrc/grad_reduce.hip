// svg-ir_amd/csrc/grad_reduce.hip -- per-Gaussian sum of the gradient rows written by the backward composite.
//
// The reference accumulates the composite's per-Gaussian gradients with one global float atomic per (pixel, splat,
// output) (backward.cu:880-930); the first version here issued one per (wave, splat, output) -- still 18 / 69 / 84
// memory-side atomics per pair, 30 % of the backward composite at the svgss widths.  Now a backward wave stores the
// complete gradient row of its (instance, sub-tile) pair (common.hpp GradRowGeom) -- compact rows, one per pair that survived
// the cull -- and notes it in the reverse map row_of[4 * emit-order instance index + sub-tile] = row + 1; the entries of one
// Gaussian are contiguous there, and this kernel adds up the Gaussian's rows in that order: streaming stores + one pass, no
// atomics, bit-reproducible gradients.
//
// A wave sums the rows of GK Gaussians: per Gaussian the 64 lanes look at 64 reverse-map entries at a time (ballot) and walk the set
// bits; lane l accumulates row element l (and l + 64 for rows longer than 64 floats).  The GK Gaussians' load chains advance together.
#include "common.hpp"

namespace svgir {

namespace {

#ifndef GRAD_REDUCE_RB
#define GRAD_REDUCE_RB 8
#endif
constexpr int RB = GRAD_REDUCE_RB;   // gradient rows in flight per Gaussian
// GK = Gaussians per wave: their dependent chains (id -> radius / tiles / first instance -> reverse map -> rows) advance TOGETHER, one
// memory round trip per link for all of them -- a wave per Gaussian pays the four links one after the other for a handful of rows, and with
// hundreds of thousands of blended Gaussians the kernel lasts its waves' latency chains, not its bytes (cfg5, 267 k blended of 2 M: 337 ->
// 269 us with GK = 4; cfg5_dense 367 -> 296 us).  With few Gaussians the waves are what is scarce (cfg3_train, 57 k blended: 59 -> 67 us
// with GK = 4), so the launcher picks GK from the model size.
template <int GK>
__global__ void __launch_bounds__(BLOCK) grad_reduce_kernel(const GradReduceArgs a, const GradRowGeom rg) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    // (with a list: only the Gaussians that received a blend weight own gradient rows; the others keep the caller's zeros)
    const int n = a.list ? (int)min(*a.list_count, (uint32_t)a.P) : a.P;
    const int first = w * GK;
    if (first >= n) return;
    const bool listed = a.list != nullptr;
    // ---- link 1: the Gaussians of this wave
    int g[GK];
#pragma unroll
    for (int k = 0; k < GK; k++) { const int i = min(first + k, n - 1); g[k] = listed ? (int)a.list[i] : i; }
    // ---- link 2: three independent loads per Gaussian
    int32_t radius[GK]; uint32_t ntiles[GK], ibase[GK];
#pragma unroll
    for (int k = 0; k < GK; k++) {
        radius[k] = a.radii[g[k]];
        ntiles[k] = a.tiles[2 * g[k]];   // (common.hpp GeomLayout::tiles: {count, rectangle})
        ibase[k] = __builtin_bit_cast(uint32_t, a.rec[(size_t)g[k] * REC + R_IBASE]);
    }
    // A LISTED Gaussian always gets its row written -- zeros if it owns none: with a list the binder may keep dL_dfeatures / dL_dvfeatures
    // outside the cleared allocation (svgir_grads: scratch feature gradients of the fused shading), and the consumer behind this kernel
    // reads the row of every listed Gaussian.  The list (forward: out_weights > 0) and the rows (backward: the replayed alpha / T tests)
    // come from two evaluations that agree bit for bit today; this keeps a disagreement from ever reading uninitialised memory.
    uint32_t nslots[GK];
#pragma unroll
    for (int k = 0; k < GK; k++) nslots[k] = (first + k < n && radius[k] > 0 && ntiles[k] != 0) ? 4u * ntiles[k] : 0u;
    // ---- link 3: the first 64 reverse-map entries of each (a Gaussian that touches more than 16 tiles continues below)
    uint32_t rv[GK];
#pragma unroll
    for (int k = 0; k < GK; k++) rv[k] = (uint32_t)lane < nslots[k] ? a.row_of[(size_t)4 * ibase[k] + lane] : 0u;
    unsigned long long m[GK];
#pragma unroll
    for (int k = 0; k < GK; k++) m[k] = __ballot(rv[k] != 0u);
    // ---- link 4: up to RB rows of every Gaussian, all loads issued before the first add; the adds keep slot order (reproducible sums)
    float acc0[GK], acc1[GK];
    bool any[GK];
    {
        float v0[GK][RB], v1[GK][RB];
        int b[GK][RB];
#pragma unroll
        for (int k = 0; k < GK; k++) {
            any[k] = m[k] != 0ull;
#pragma unroll
            for (int j = 0; j < RB; j++) {
                b[k][j] = m[k] ? __builtin_ctzll(m[k]) : -1;
                m[k] &= m[k] - 1;   // (0 & anything stays 0)
            }
#pragma unroll
            for (int j = 0; j < RB; j++) {
                const uint32_t ri = (uint32_t)__builtin_amdgcn_readlane((int)rv[k], b[k][j] >= 0 ? b[k][j] : 0) - 1u;
                const float* row = a.grad_rows + (size_t)(b[k][j] >= 0 ? ri : 0u) * (size_t)rg.RS;
                v0[k][j] = (b[k][j] >= 0 && lane < rg.RS) ? row[lane] : 0.f;
                v1[k][j] = (b[k][j] >= 0 && lane + 64 < rg.RS) ? row[lane + 64] : 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < GK; k++) {
            acc0[k] = 0.f; acc1[k] = 0.f;
#pragma unroll
            for (int j = 0; j < RB; j++)
                if (b[k][j] >= 0) { acc0[k] += v0[k][j]; acc1[k] += v1[k][j]; }
        }
    }
    // ---- the rest (more than RB rows among the first 64 entries, or more than 64 entries): one Gaussian at a time, as before
#pragma unroll
    for (int k = 0; k < GK; k++) {
        if (m[k] == 0ull && nslots[k] <= 64u) continue;   // (uniform)
        const size_t base = (size_t)4 * ibase[k];
        uint32_t rvk = rv[k];
        unsigned long long mk = m[k];
        for (uint32_t s0 = 0; s0 < nslots[k]; s0 += 64) {
            if (s0 != 0) {
                const uint32_t s = s0 + (uint32_t)lane;
                rvk = s < nslots[k] ? a.row_of[base + s] : 0u;
                mk = __ballot(rvk != 0u);
                any[k] = any[k] || mk != 0ull;
            }
            while (mk) {
                int bb[RB];
#pragma unroll
                for (int j = 0; j < RB; j++) {
                    bb[j] = mk ? __builtin_ctzll(mk) : -1;
                    mk &= mk - 1;
                }
                float v0[RB], v1[RB];
#pragma unroll
                for (int j = 0; j < RB; j++) {
                    const uint32_t ri = (uint32_t)__builtin_amdgcn_readlane((int)rvk, bb[j] >= 0 ? bb[j] : bb[0]) - 1u;
                    const float* row = a.grad_rows + (size_t)ri * (size_t)rg.RS;
                    v0[j] = lane < rg.RS ? row[lane] : 0.f;
                    v1[j] = lane + 64 < rg.RS ? row[lane + 64] : 0.f;
                }
#pragma unroll
                for (int j = 0; j < RB; j++)
                    if (bb[j] >= 0) { acc0[k] += v0[j]; acc1[k] += v1[j]; }
            }
        }
    }
    // ---- scatter the summed rows to the output tensors (the caller zero-fills them; this is the only writer)
#pragma unroll
    for (int k = 0; k < GK; k++) {
        if (first + k >= n) continue;
        if (!any[k] && !listed) continue;   // outputs stay at the caller's zeros
        const int gg = g[k];
        auto put = [&](int e, float v) {
            if (e < 3) a.dL_dcolor[(size_t)gg * 3 + e] = v;
            else if (e < 6) a.dL_dnormal[(size_t)gg * 3 + (e - 3)] = v;
            else if (e < 7) a.dL_ddepth[gg] = v;
            else if (e < rg.NC0) a.dL_dfeature[(size_t)gg * a.S + (e - 7)] = v;
            else if (e < rg.P4) {}
            else if (e < rg.GEO) a.dL_dvfeature[(size_t)gg * a.VS + (e - rg.P4)] = v;
            else if (e < rg.GEO + 2) a.dL_dmean2D[(size_t)gg * 3 + (e - rg.GEO)] = v;
            else if (e < rg.GEO + 5) { const int j = e - rg.GEO - 2; a.dL_dconic[(size_t)gg * 4 + (j == 2 ? 3 : j)] = v; }
            else if (e < rg.GEO + 6) a.dL_dopacity[gg] = v;
        };
        if (lane < rg.RS) put(lane, acc0[k]);
        if (lane + 64 < rg.RS) put(lane + 64, acc1[k]);
    }
}

}  // namespace

void launch_grad_reduce(const GradReduceArgs& a, hipStream_t s) {
    const GradRowGeom rg = grad_row_geom(a.S, a.VS);
    // (grids are sized for all P: with a list the surplus workgroups exit at once)
    auto grid = [&](int gk) { const int per = (BLOCK / 64) * gk; return dim3((a.P + per - 1) / per); };
    if (a.P >= 1000000) hipLaunchKernelGGL(grad_reduce_kernel<4>, grid(4), dim3(BLOCK), 0, s, a, rg);
    else if (a.P >= 500000) hipLaunchKernelGGL(grad_reduce_kernel<2>, grid(2), dim3(BLOCK), 0, s, a, rg);
    else hipLaunchKernelGGL(grad_reduce_kernel<1>, grid(1), dim3(BLOCK), 0, s, a, rg);
}

}  // namespace svgir
