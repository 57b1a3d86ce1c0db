// Development probe: throughput of global integer atomics (returning / non-returning) onto a few thousand counters,
// uniform and skewed -- sizing the atomics-based tile binning.   hipcc --offload-arch=gfx950 -O3 atomic_probe.hip -o atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__device__ __forceinline__ uint32_t hsh(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
template <bool RET, bool SKEW>
__global__ void k(uint32_t* cnt, int T, int per, uint32_t* sink) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (int j = 0; j < per; j++) {
        uint32_t h = hsh(i * 16u + j);
        uint32_t t = h % (uint32_t)T;
        if (SKEW && (h >> 28) < 4u) t = (h >> 8) % 16u;   // 25 % of the traffic onto 16 hot counters
        if (RET) acc += atomicAdd(&cnt[t], 1u); else atomicAdd(&cnt[t], 1u);
    }
    if (RET && acc == 0xdeadbeefu) sink[0] = acc;
}
template <bool RET, bool SKEW>
void run(const char* name, int n, int per, int T, uint32_t* cnt, uint32_t* sink) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int it = 0; it < 5; it++) {
        hipMemset(cnt, 0, T * 4);
        hipEventRecord(a);
        hipLaunchKernelGGL((k<RET, SKEW>), dim3(n / 256), dim3(256), 0, 0, cnt, T, per, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    printf("%-34s threads %8d x %2d atomics onto %6d counters: %8.1f us  (%.2f G atomics/s)\n", name, n, per, T, best * 1e3, (double)n * per / (best * 1e-3) / 1e9);
}
int main() {
    uint32_t *cnt, *sink; hipMalloc(&cnt, 1 << 22); hipMalloc(&sink, 64);
    for (int T : {2500, 40000}) {
        for (int n : {90112, 2000128}) {
            run<false, false>("non-returning uniform", n, 8, T, cnt, sink);
            run<false, true>("non-returning skewed", n, 8, T, cnt, sink);
            run<true, false>("returning uniform", n, 8, T, cnt, sink);
            run<true, true>("returning skewed", n, 8, T, cnt, sink);
        }
    }
    return 0;
}
