// svg-ir_amd/csrc/dev_trace.hpp -- per-wave timeline probes for kernel development.  Compiled out of the product:
// every macro is empty unless the translation unit is built with -DSVGIR_DEV (scripts/build_variant.sh <name> -DSVGIR_DEV),
// and only such a build exports svgir_dev_trace_read().
#pragma once
#if defined(SVGIR_DEV) && defined(__HIPCC__)
#include <hip/hip_runtime.h>
namespace svgir {
constexpr int DEV_TRACE_WORDS = 8;            // u64 words per wave record
constexpr int DEV_TRACE_CAP = 1 << 17;        // records per kernel slot
static __device__ unsigned long long g_dev_trace[2][DEV_TRACE_CAP * DEV_TRACE_WORDS];   // slot 0 forward, 1 backward
static __device__ unsigned int g_dev_trace_n[2];
struct DevTrace {
    unsigned long long t0, r0, acc[4], prev;
    __device__ __forceinline__ void begin() {
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); prev = t0;
        acc[0] = acc[1] = acc[2] = acc[3] = 0;
    }
    __device__ __forceinline__ void mark(int i) {
        const unsigned long long t = __builtin_amdgcn_s_memtime(); acc[i] += t - prev; prev = t;
    }
    __device__ __forceinline__ void end(int slot, unsigned int a, unsigned int b, unsigned int c) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0) {
            const unsigned int i = atomicAdd(&g_dev_trace_n[slot], 1u);
            if (i < (unsigned)DEV_TRACE_CAP) {
                unsigned long long* d = g_dev_trace[slot] + (size_t)i * DEV_TRACE_WORDS;
                unsigned int hw;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
                d[0] = t1 - t0; d[1] = r0; d[2] = r1; d[3] = ((unsigned long long)a << 32) | b;
                d[4] = ((unsigned long long)c << 32) | hw; d[5] = acc[0]; d[6] = acc[1];
                d[7] = (acc[2] << 32) | (acc[3] & 0xffffffffull);
            }
        }
    }
};
}  // namespace svgir
#define DEV_TRACE_DECL() svgir::DevTrace dev_tr; dev_tr.begin()
#define DEV_TRACE_MARK(i) dev_tr.mark(i)
#define DEV_TRACE_END(slot, a, b, c) dev_tr.end(slot, a, b, c)
#else
#define DEV_TRACE_DECL()
#define DEV_TRACE_MARK(i)
#define DEV_TRACE_END(slot, a, b, c)
#endif
