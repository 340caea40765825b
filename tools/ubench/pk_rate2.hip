// Micro-benchmark 2: VALU issue rate AND dependent-issue latency with DISTINCT source registers
// (pk_rate.hip reuses two register pairs per op, which hides any register-file port limit).
// Forms: packed f32 / scalar f32 / f64 FMA with three distinct VGPR sources, with one SGPR source;
// ILP = independent chains per wave (1 = pure dependent chain -> latency), waves per SIMD 1..8.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

template <int V, int ILP>
__global__ void __launch_bounds__(256) k(double *out, int iters, double seed)
{
    double a[8], b[8], c[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed * (i + 1 + threadIdx.x); b[i] = seed * 0.5 * (i + 3); c[i] = seed * 0.25 * (i + 7); }
    const double sc = seed * 1.0001;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 32 / ILP; ++r) {
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if (V == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));
                if (V == 1) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));
                if (V == 2) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "s"(sc), "v"(c[i]));
                if (V == 3) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(((float *)&a[i])[0]) : "v"(((float *)&b[i])[0]), "v"(((float *)&c[i])[0]));
                if (V == 4) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
                if (V == 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));
                if (V == 6) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "s"(sc), "v"(c[i]));
                if (V == 7) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
                if (V == 8) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
                if (V == 9) asm volatile("v_add_f64 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
            }
        }
    }
    double s = a[0];
    for (int i = 1; i < 8; ++i) s += a[i];
    if (s == 1.2345) out[0] = s;
}
template <int V, int ILP> static void run(const char *name, int blocks_per_cu)
{
    double *out; CK(hipMalloc(&out, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000, grid = 256 * blocks_per_cu;            // blocks of 4 waves: blocks_per_cu waves per SIMD
    hipLaunchKernelGGL((k<V, ILP>), dim3(grid), dim3(256), 0, 0, out, 100, 1.0);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<V, ILP>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double winstr = (double)grid * 4 * iters * 32;
    const double per_simd_per_s = winstr / (256.0 * 4) / (ms * 1e-3);
    printf("%-44s ILP %d  waves/SIMD %d  one per %6.2f clk per SIMD  (%.2f clk per wave)\n", name, ILP, blocks_per_cu,
           2.4e9 / per_simd_per_s, 2.4e9 / per_simd_per_s * blocks_per_cu);
    CK(hipFree(out));
}
template <int V> static void sweep(const char *name)
{
    run<V, 8>(name, 8); run<V, 8>(name, 4); run<V, 8>(name, 2); run<V, 8>(name, 1);
    run<V, 2>(name, 4); run<V, 2>(name, 2); run<V, 2>(name, 1);
    run<V, 1>(name, 4); run<V, 1>(name, 2); run<V, 1>(name, 1);
}
int main()
{
    sweep<3>("v_fma_f32 3 distinct VGPR");
    sweep<0>("v_pk_fma_f32 3 distinct VGPR pairs");
    sweep<5>("v_pk_fma_f32 3 distinct + op_sel swap");
    sweep<6>("v_pk_fma_f32 SGPR pair + 2 VGPR pairs");
    sweep<4>("v_pk_mul_f32 2 distinct");
    sweep<7>("v_pk_add_f32 2 distinct");
    sweep<1>("v_fma_f64 3 distinct VGPR pairs");
    sweep<2>("v_fma_f64 SGPR pair + 2 VGPR pairs");
    sweep<8>("v_mul_f64 2 distinct");
    sweep<9>("v_add_f64 2 distinct");
    return 0;
}
