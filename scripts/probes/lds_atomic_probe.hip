// LDS atomic throughput on gfx950: ds_add_f32 vs ds_add_u32 vs ds_add_u64 onto a 6144-entry workgroup-private image,
// pseudo-random addresses (the access pattern of shade_bwd's env-gradient scatter).  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int NTEX = 6144, ITER = 256;
template <int MODE, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) probe(float* out, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* f = reinterpret_cast<float*>(smem);
    unsigned* u = reinterpret_cast<unsigned*>(smem);
    unsigned long long* q = reinterpret_cast<unsigned long long*>(smem);
    for (int i = threadIdx.x; i < NTEX * (MODE == 2 || MODE == 4 ? 2 : 1); i += WAVES * 64) u[i] = 0;
    __syncthreads();
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; it++) {
        s = s * 1664525u + 1013904223u;
        const unsigned idx = (s >> 8) % NTEX;
        const float v = (float)(s & 255u) * 1e-3f;
        if (MODE == 0) atomicAdd(&f[idx], v);
        else if (MODE == 1) atomicAdd(&u[idx], (unsigned)(v * 65536.f));
        else if (MODE == 2) atomicAdd(&q[idx], (unsigned long long)(long long)__builtin_rintf(v * 1099511627776.f));
        else if (MODE == 3) f[idx] = v;   // plain store for scale
        else if (MODE == 4) atomicAdd(reinterpret_cast<double*>(smem) + idx, (double)v);
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0;
    for (int i = threadIdx.x; i < NTEX; i += WAVES * 64) acc += MODE == 2 ? (float)q[i] : f[i];
    out[blockIdx.x * WAVES * 64 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE, int WAVES>
void run(const char* name, float* out, unsigned long long* cyc, int wg_per_cu) {
    const int grid = 256 * wg_per_cu;
    const size_t lds = (size_t)NTEX * (MODE == 2 || MODE == 4 ? 8 : 4);
    hipLaunchKernelGGL((probe<MODE, WAVES>), dim3(grid), dim3(WAVES * 64), lds, 0, out, cyc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    double m = 0;
    for (auto v : h) m += (double)v;
    m /= grid;
    // s_memtime ticks at 100 MHz on this part: report ns per wave instruction per CU
    printf("%-34s %d waves/WG x %d WG/CU: %8.1f memtime ticks per WG, %.2f ticks per wave-instruction per CU\n", name, WAVES, wg_per_cu, m,
           m / ((double)ITER * WAVES * wg_per_cu));
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 4 * 1024 * 4);
    hipMalloc(&cyc, 256 * 4 * 8);
    for (int rep = 0; rep < 2; rep++) {
        run<0, 4>("ds_add_f32", out, cyc, 3);
        run<1, 4>("ds_add_u32", out, cyc, 3);
        run<2, 4>("ds_add_u64", out, cyc, 3);
        run<3, 4>("ds_write_b32", out, cyc, 3);
        run<4, 4>("ds_add_f64", out, cyc, 3);
        run<4, 12>("ds_add_f64", out, cyc, 1);
        run<0, 12>("ds_add_f32", out, cyc, 1);
        run<2, 12>("ds_add_u64", out, cyc, 1);
        run<2, 16>("ds_add_u64", out, cyc, 1);
    }
    return 0;
}
