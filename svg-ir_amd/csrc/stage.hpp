// svg-ir_amd/csrc/stage.hpp -- LDS staging of a batch of splats, shared by the forward and backward composite
// kernels.
//
// Staged layout of one splat (floats):  [0,24) the record written by preprocess.hip (common.hpp RecField),
// [24, 24+S4) the S feature floats padded to a float4 boundary, [24+S4, NF) the VS vfeature floats.
// Every slot starts on a 16-byte boundary, so the walking waves read it with broadcast ds_read_b128.
#pragma once
#include "common.hpp"

namespace svgir {

template <int S, int VC>
struct StageGeom {
    static constexpr int S4 = (S + 3) / 4 * 4;      // feature slots padded to a float4 boundary
    static constexpr int NF = REC + S4 + VC * 4;     // floats per staged splat
    static constexpr int NF4 = NF / 4;
    static constexpr int C4 = 6 + VC;                // float4 chunks gathered with 16-byte loads (record + vfeatures)
    static constexpr int BATCH = NF <= 32 ? 256 : (NF <= 64 ? 192 : 128);  // <= 48 KB of LDS per workgroup
    static constexpr int F_OFF = REC;                // features
    static constexpr int V_OFF = REC + S4;           // vfeatures
    static constexpr size_t lds_bytes() { return (size_t)BATCH * NF * 4 + (size_t)BATCH * 4; }
};

#if defined(__HIPCC__)
// Gathers splats [0, n) of the current batch into LDS.  ids[s] must already hold the Gaussian id of slot s
// (written before a barrier).  All 256 threads take part; loads are 16 bytes wide except the S feature floats
// (rows of S floats are not 16-byte aligned in the caller's [P,S] tensor).
template <int S, int VC>
__device__ __forceinline__ void stage_batch(float* __restrict__ sD, const int* __restrict__ ids, int n,
                                            const float* __restrict__ rec, const float* __restrict__ feat,
                                            const float* __restrict__ vfeat) {
    using SG = StageGeom<S, VC>;
    float4* sD4 = reinterpret_cast<float4*>(sD);
    const float4* rec4 = reinterpret_cast<const float4*>(rec);
    const float4* vf4 = reinterpret_cast<const float4*>(vfeat);
    const int total = n * SG::C4;
    for (int k = threadIdx.x; k < total; k += BLOCK) {
        const int s = k / SG::C4, part = k - s * SG::C4;
        const size_t id = (size_t)ids[s];
        if (part < 6) sD4[s * SG::NF4 + part] = rec4[id * 6 + part];
        else sD4[s * SG::NF4 + SG::V_OFF / 4 + (part - 6)] = vf4[id * VC + (part - 6)];
    }
    if (S > 0) {
        const int totf = n * S;
        for (int k = threadIdx.x; k < totf; k += BLOCK) {
            const int s = k / S, c = k - s * S;
            sD[s * SG::NF + SG::F_OFF + c] = feat[(size_t)ids[s] * S + c];
        }
    }
}
#endif

}  // namespace svgir
