// svg-ir_amd/csrc/shade_tables.hpp -- the two small tables the shading kernels read (csrc/shade.hip): f(env) as one float4 per texel and
// the per-sample part of the incident-direction lattice.  They are a few thousand entries, i.e. a launch of their own costs more than
// the work (~6 us of stream time): svgir_shade_forward / _backward build them with one small prologue launch, the fused path
// (svgir_params.shade) lets kernels that run anyway carry them -- the preprocess kernel in the forward, the live-segment kernel in the
// backward (extra work items behind their own).
#pragma once
#include "common.hpp"

namespace svgir {

// (struct ShadeTables: common.hpp)

#if defined(__HIPCC__)
constexpr float kLatticeDelta = 2.39996322972865332f;   // fp32(pi * (3 - sqrt(5)))

__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }

// entry i of both tables (i < ShadeTables::entries())
__device__ __forceinline__ void shade_table_entry(const ShadeTables& t, int i) {
    if (i < t.ntexel) {
        const float r = t.env[3 * i], g = t.env[3 * i + 1], b = t.env[3 * i + 2];
        t.env_tab[i] = t.softplus ? make_float4(softplus_f(r), softplus_f(g), softplus_f(b), 0.f) : make_float4(r, g, b, 0.f);
    }
    if (t.lat_tab && i < t.Ns) {
#pragma clang fp contract(off)
        const float fi = (float)i;
        const float z = fmaxf(1.f - 2.f * fi / (float)(2 * t.Ns - 1), 0.17364817766693033f);   // sin(10 deg)
        const float rad = sqrtf(1.f - z * z);
        const float th = kLatticeDelta * fi;
        float sn, cs;
        sincosf(th, &sn, &cs);
        t.lat_tab[i] = make_float4(sn, cs, z, rad);
    }
}
#endif

}  // namespace svgir
