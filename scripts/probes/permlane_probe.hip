// scripts/probes/permlane_probe.hip -- what v_permlane16_swap / v_permlane32_swap (gfx950) do to a register swapped with itself:
// prints, per lane, the sum of the two results (expected: rows pairwise summed / wave halves summed).
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/permlane_probe.hip -o build/probes/permlane_probe && build/probes/permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out) {
    const int l = threadIdx.x;
    const float v = (float)(1 << (l >> 4)) * 100.f + (float)(l & 15);   // row r holds 100 * 2^r + lane-in-row
    // (inline asm: with the builtins' two-element result this compiler adds element 0 to itself)
    float x = v, y = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
    const float s16 = x + y;
    float p = s16, q = s16;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(p), "+v"(q));
    const float s32 = p + q;
    out[l] = s16; out[64 + l] = s32;
}
int main() {
    float* d; float h[128];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int r = 0; r < 4; r++) printf("row %d: s16 lane0 %.0f lane5 %.0f | s32 lane0 %.0f lane5 %.0f\n", r, h[16 * r], h[16 * r + 5], h[64 + 16 * r], h[64 + 16 * r + 5]);
    // expected: s16 rows 0,1 = 300 + 2 l ; rows 2,3 = 1200 + 2 l ; s32 all rows = 1500 + 4 l
    bool ok = true;
    for (int l = 0; l < 64; l++) {
        const int r = l >> 4, i = l & 15;
        ok = ok && h[l] == (r < 2 ? 300.f : 1200.f) + 2.f * i && h[64 + l] == 1500.f + 4.f * i;
    }
    printf("%s\n", ok ? "OK: swap-with-self + add = pairwise row sums, then wave-half sums" : "MISMATCH");
    return ok ? 0 : 1;
}
