// svg-ir_amd/csrc/lbvh.hpp -- the few device helpers the two LBVH builders (bvh.hip, pbgi.hip) share.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

namespace svgir {

// order-preserving float <-> uint (for atomicMin / atomicMax on floats)
__device__ __forceinline__ uint32_t f2ord(float f) {
    const uint32_t u = __builtin_bit_cast(uint32_t, f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
    return __builtin_bit_cast(float, (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}
// 10 bits -> every third bit of 30
__device__ __forceinline__ uint32_t expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

}  // namespace svgir
