// Standalone micro-benchmark: issue rate of wave64 VALU instructions on gfx950 (cycles per instruction per SIMD),
// for independent and dependent v_fma_f32 chains, packed v_pk_fma_f32, v_exp_f32, DPP moves and ds_write/ds_read,
// at 1..8 waves per SIMD.  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 scripts/probes/valu_rate_probe.hip -o build/valu_rate_probe && build/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ITER = 4096;

template <int MODE>
__global__ void __launch_bounds__(64) probe(float* out, unsigned long long* cyc) {
    __shared__ float lds[64 * 8];
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float m = 0.999f, c = 1e-4f;
    f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const f32x2 pm = {m, m}, pc = {c, c};
    lds[threadIdx.x] = a0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < ITER; i++) {
        if (MODE == 0) {   // 8 independent fma chains
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if (MODE == 1) {   // one dependent chain of 8
            asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         : "+v"(a0) : "v"(m), "v"(c));
        } else if (MODE == 2) {   // 8 independent packed fmas (4 registers pairs x 2)
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm), "v"(pc));
        } else if (MODE == 3) {   // 8 independent exp2
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                         "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 4) {   // 8 independent DPP adds
            asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %6, %6, %6 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 6) {   // 8 fma + 8 independent SALU, interleaved
            unsigned s0 = i, s1 = i + 1;
            asm volatile("v_fma_f32 %0, %0, %10, %11\n s_add_u32 %8, %8, 1\n v_fma_f32 %1, %1, %10, %11\n s_add_u32 %9, %9, 3\n v_fma_f32 %2, %2, %10, %11\n s_add_u32 %8, %8, 1\n v_fma_f32 %3, %3, %10, %11\n s_add_u32 %9, %9, 3\n"
                         "v_fma_f32 %4, %4, %10, %11\n s_add_u32 %8, %8, 1\n v_fma_f32 %5, %5, %10, %11\n s_add_u32 %9, %9, 3\n v_fma_f32 %6, %6, %10, %11\n s_add_u32 %8, %8, 1\n v_fma_f32 %7, %7, %10, %11\n s_add_u32 %9, %9, 3\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0), "+s"(s1) : "v"(m), "v"(c));
        } else if (MODE == 7) {   // 8 fma + 16 SALU (2 per fma)
            unsigned s0 = i, s1 = i + 1;
            asm volatile("v_fma_f32 %0, %0, %10, %11\n s_add_u32 %8, %8, 1\n s_add_u32 %9, %9, 3\n v_fma_f32 %1, %1, %10, %11\n s_add_u32 %8, %8, 1\n s_add_u32 %9, %9, 3\n v_fma_f32 %2, %2, %10, %11\n s_add_u32 %8, %8, 1\n s_add_u32 %9, %9, 3\n v_fma_f32 %3, %3, %10, %11\n s_add_u32 %8, %8, 1\n s_add_u32 %9, %9, 3\n"
                         "v_fma_f32 %4, %4, %10, %11\n s_add_u32 %8, %8, 1\n s_add_u32 %9, %9, 3\n v_fma_f32 %5, %5, %10, %11\n s_add_u32 %8, %8, 1\n s_add_u32 %9, %9, 3\n v_fma_f32 %6, %6, %10, %11\n s_add_u32 %8, %8, 1\n s_add_u32 %9, %9, 3\n v_fma_f32 %7, %7, %10, %11\n s_add_u32 %8, %8, 1\n s_add_u32 %9, %9, 3\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0), "+s"(s1) : "v"(m), "v"(c));
        } else if (MODE == 8) {   // 8 fma with an SGPR operand
            float sc = __builtin_bit_cast(float, (unsigned)(0x3f7fbe77u));
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sc), "v"(c));
        } else if (MODE == 9) {   // 8 broadcast ds_read_b128 (uniform address) + wait
            f32x4 r0, r1, r2, r3, r4, r5, r6, r7;
            unsigned addr = (unsigned)(size_t)lds;
            asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:16\n ds_read_b128 %2, %8 offset:32\n ds_read_b128 %3, %8 offset:48\n"
                         "ds_read_b128 %4, %8 offset:64\n ds_read_b128 %5, %8 offset:80\n ds_read_b128 %6, %8 offset:96\n ds_read_b128 %7, %8 offset:112\n s_waitcnt lgkmcnt(0)\n"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(addr) : "memory");
            a0 += r0.x + r1.x + r2.x + r3.x + r4.x + r5.x + r6.x + r7.x;
        } else if (MODE == 10) {   // 8 dependent-accumulator f32 MFMAs (2 chains of 4)
            f32x4 d0 = {a0, a1, a2, a3}, d1 = {a4, a5, a6, a7};
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %2, %3, %0\n v_mfma_f32_16x16x4_f32 %1, %2, %3, %1\n v_mfma_f32_16x16x4_f32 %0, %2, %3, %0\n v_mfma_f32_16x16x4_f32 %1, %2, %3, %1\n"
                         "v_mfma_f32_16x16x4_f32 %0, %2, %3, %0\n v_mfma_f32_16x16x4_f32 %1, %2, %3, %1\n v_mfma_f32_16x16x4_f32 %0, %2, %3, %0\n v_mfma_f32_16x16x4_f32 %1, %2, %3, %1\n s_nop 7\n s_nop 7\n"
                         : "+v"(d0), "+v"(d1) : "v"(m), "v"(c));
            a0 = d0.x; a1 = d0.y; a2 = d0.z; a3 = d0.w; a4 = d1.x; a5 = d1.y; a6 = d1.z; a7 = d1.w;
        } else if (MODE == 11) {   // 4 MFMAs interleaved with 28 independent fmas (7 per MFMA)
            f32x4 d0 = {p0.x, p0.y, p1.x, p1.y};
            asm volatile("v_mfma_f32_16x16x4_f32 %8, %9, %10, %8\n v_fma_f32 %0, %0, %9, %10\n v_fma_f32 %1, %1, %9, %10\n v_fma_f32 %2, %2, %9, %10\n v_fma_f32 %3, %3, %9, %10\n v_fma_f32 %4, %4, %9, %10\n v_fma_f32 %5, %5, %9, %10\n v_fma_f32 %6, %6, %9, %10\n"
                         "v_mfma_f32_16x16x4_f32 %8, %9, %10, %8\n v_fma_f32 %0, %0, %9, %10\n v_fma_f32 %1, %1, %9, %10\n v_fma_f32 %2, %2, %9, %10\n v_fma_f32 %3, %3, %9, %10\n v_fma_f32 %4, %4, %9, %10\n v_fma_f32 %5, %5, %9, %10\n v_fma_f32 %6, %6, %9, %10\n"
                         "v_mfma_f32_16x16x4_f32 %8, %9, %10, %8\n v_fma_f32 %0, %0, %9, %10\n v_fma_f32 %1, %1, %9, %10\n v_fma_f32 %2, %2, %9, %10\n v_fma_f32 %3, %3, %9, %10\n v_fma_f32 %4, %4, %9, %10\n v_fma_f32 %5, %5, %9, %10\n v_fma_f32 %6, %6, %9, %10\n"
                         "v_mfma_f32_16x16x4_f32 %8, %9, %10, %8\n v_fma_f32 %0, %0, %9, %10\n v_fma_f32 %1, %1, %9, %10\n v_fma_f32 %2, %2, %9, %10\n v_fma_f32 %3, %3, %9, %10\n v_fma_f32 %4, %4, %9, %10\n v_fma_f32 %5, %5, %9, %10\n v_fma_f32 %6, %6, %9, %10\n s_nop 7\n s_nop 7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(d0) : "v"(m), "v"(c));
            p0.x = d0.x;
        } else if (MODE == 5) {   // 8 ds_write_b32 (lane = column, conflict-free)
            float* p = lds + threadIdx.x;
            asm volatile("ds_write_b32 %0, %1\n ds_write_b32 %0, %2 offset:256\n ds_write_b32 %0, %3 offset:512\n ds_write_b32 %0, %4 offset:768\n"
                         "ds_write_b32 %0, %5 offset:1024\n ds_write_b32 %0, %6 offset:1280\n ds_write_b32 %0, %7 offset:1536\n ds_write_b32 %0, %8 offset:1792\n s_waitcnt lgkmcnt(0)\n"
                         :: "v"((unsigned)(size_t)p), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "memory");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + lds[threadIdx.x];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, float* out, unsigned long long* cyc) {
    printf("%-28s", name);
    for (int wps : {1, 2, 3, 4, 6, 8}) {
        const int grid = 256 * 4 * wps;   // one-wave workgroups: wps waves per SIMD when the chip is full
        hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(64), 0, 0, out, cyc);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(grid);
        hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
        double s = 0;
        for (auto v : h) s += (double)v;
        // cycles per instruction per SIMD = (mean wave cycles) / (instructions per wave) / (waves sharing the SIMD)
        printf("  w%d: %.2f", wps, s / grid / (8.0 * ITER) / wps);
    }
    printf("   (SIMD cycles per wave64 instruction)\n");
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 4 * 8 * 64 * 4);
    hipMalloc(&cyc, 256 * 4 * 8 * 8);
    run<0>("v_fma_f32 independent x8", out, cyc);
    run<1>("v_fma_f32 dependent chain", out, cyc);
    run<2>("v_pk_fma_f32 independent", out, cyc);
    run<3>("v_exp_f32 independent", out, cyc);
    run<4>("v_add_f32 dpp independent", out, cyc);
    run<5>("ds_write_b32 x8 + wait", out, cyc);
    run<6>("8 fma + 8 salu (per 8 slots)", out, cyc);
    run<7>("8 fma + 16 salu (per 8)", out, cyc);
    run<8>("v_fma_f32 sgpr operand", out, cyc);
    run<9>("ds_read_b128 bcast x8 + wait", out, cyc);
    run<10>("mfma 16x16x4 f32 x8", out, cyc);
    run<11>("4 mfma + 28 fma (per 8)", out, cyc);
    return 0;
}
