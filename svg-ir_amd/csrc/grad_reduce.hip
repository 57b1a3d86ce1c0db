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
// One wave per Gaussian: the 64 lanes first look at 64 reverse-map entries at a time (ballot), then walk the set bits;
// lane l accumulates row element l (and l + 64 for rows longer than 64 floats).
#include "common.hpp"

namespace svgir {

namespace {

#ifndef GRAD_REDUCE_RB
#define GRAD_REDUCE_RB 8
#endif
constexpr int RB = GRAD_REDUCE_RB;   // gradient rows in flight per wave

__global__ void __launch_bounds__(BLOCK) grad_reduce_kernel(const GradReduceArgs a, const GradRowGeom rg) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    // (with a list: only the Gaussians that received a blend weight own gradient rows; the others keep the caller's zeros)
    if (w >= (a.list ? (int)min(*a.list_count, (uint32_t)a.P) : a.P)) return;
    const int g = a.list ? (int)a.list[w] : w;
    // three independent loads first (one memory latency), then the early exits
    const int32_t radius = a.radii[g];
    const uint32_t ntiles = a.tiles[g];
    const size_t base = (size_t)4 * __builtin_bit_cast(uint32_t, a.rec[(size_t)g * REC + R_IBASE]);
    // A LISTED Gaussian always gets its row written -- zeros if it owns none: with a list the binder may keep dL_dfeatures / dL_dvfeatures
    // outside the cleared allocation (svgir_grads: scratch feature gradients of the fused shading), and the consumer behind this kernel
    // reads the row of every listed Gaussian.  The list (forward: out_weights > 0) and the rows (backward: the replayed alpha / T tests)
    // come from two evaluations that agree bit for bit today; this keeps a disagreement from ever reading uninitialised memory.
    const bool listed = a.list != nullptr;
    const bool visible = radius > 0 && ntiles != 0;
    if (!visible && !listed) return;
    const uint32_t nslots = visible ? 4u * ntiles : 0u;
    float acc0 = 0.f, acc1 = 0.f;
    bool any = false;
    for (uint32_t s0 = 0; s0 < nslots; s0 += 64) {
        const uint32_t s = s0 + (uint32_t)lane;
        const uint32_t rv = s < nslots ? a.row_of[base + s] : 0u;   // row + 1, or 0
        unsigned long long m = __ballot(rv != 0u);
        any = any || m != 0ull;
        // up to RB valid rows per step: all their loads are issued before the first add (one memory latency per RB rows
        // instead of one per row -- most Gaussians have fewer than RB valid rows, i.e. one round trip); the adds keep slot
        // order, so the result is reproducible
        while (m) {
            int b[RB];
#pragma unroll
            for (int k = 0; k < RB; k++) {
                b[k] = m ? __builtin_ctzll(m) : -1;
                m &= m - 1;   // (0 & anything stays 0)
            }
            float v0[RB], v1[RB];
#pragma unroll
            for (int k = 0; k < RB; k++) {
                const uint32_t ri = (uint32_t)__builtin_amdgcn_readlane((int)rv, b[k] >= 0 ? b[k] : b[0]) - 1u;
                const float* row = a.grad_rows + (size_t)ri * (size_t)rg.RS;
                v0[k] = lane < rg.RS ? row[lane] : 0.f;
                v1[k] = lane + 64 < rg.RS ? row[lane + 64] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < RB; k++)
                if (b[k] >= 0) { acc0 += v0[k]; acc1 += v1[k]; }
        }
    }
    if (!any && !listed) return;   // outputs stay at the caller's zeros
    // scatter the summed row to the output tensors (the caller zero-fills them; this is the only writer)
    auto put = [&](int e, float v) {
        if (e < 3) a.dL_dcolor[(size_t)g * 3 + e] = v;
        else if (e < 6) a.dL_dnormal[(size_t)g * 3 + (e - 3)] = v;
        else if (e < 7) a.dL_ddepth[g] = v;
        else if (e < rg.NC0) a.dL_dfeature[(size_t)g * a.S + (e - 7)] = v;
        else if (e < rg.P4) {}
        else if (e < rg.GEO) a.dL_dvfeature[(size_t)g * a.VS + (e - rg.P4)] = v;
        else if (e < rg.GEO + 2) a.dL_dmean2D[(size_t)g * 3 + (e - rg.GEO)] = v;
        else if (e < rg.GEO + 5) { const int j = e - rg.GEO - 2; a.dL_dconic[(size_t)g * 4 + (j == 2 ? 3 : j)] = v; }
        else if (e < rg.GEO + 6) a.dL_dopacity[g] = v;
    };
    if (lane < rg.RS) put(lane, acc0);
    if (lane + 64 < rg.RS) put(lane + 64, acc1);
}

}  // namespace

void launch_grad_reduce(const GradReduceArgs& a, hipStream_t s) {
    const GradRowGeom rg = grad_row_geom(a.S, a.VS);
    const int per = BLOCK / 64;
    hipLaunchKernelGGL(grad_reduce_kernel, dim3((a.P + per - 1) / per), dim3(BLOCK), 0, s, a, rg);
}

}  // namespace svgir
