// Micro-benchmark: dependent-issue latency of v_fma_f64 / v_fma_f32 / v_pk_fma_f32 on gfx950 -- one wave per SIMD running
// NCH independent dependency chains; clocks per instruction per wave from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

template <int KIND, int NCH>
__global__ void __launch_bounds__(64) k(double *out, long long *cyc, int iters, double seed)
{
    double a[NCH];
    float f[NCH];
    v2f p[NCH];
    for (int i = 0; i < NCH; ++i) { a[i] = seed * (i + 1 + threadIdx.x); f[i] = (float)a[i]; p[i] = v2f{f[i], f[i] * 0.5f}; }
    const double c = 0.999999, d = 1e-9;
    const float cf = 0.999999f, df = 1e-9f;
    const v2f cp = {cf, cf}, dp = {df, df};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
                if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(cf), "v"(df));
                if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(cp), "v"(dp));
                if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(dp));
            }
    }
    const long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < NCH; ++i) s += a[i] + f[i] + p[i].x + p[i].y;
    if (s == 1.2345) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int KIND, int NCH> static void run(const char *name)
{
    double *out; long long *cyc; CK(hipMalloc(&out, 8)); CK(hipMalloc(&cyc, 8));
    const int iters = 2000;
    hipLaunchKernelGGL((k<KIND, NCH>), dim3(1), dim3(64), 0, 0, out, cyc, 10, 1.0);
    hipLaunchKernelGGL((k<KIND, NCH>), dim3(1), dim3(64), 0, 0, out, cyc, iters, 1.0);
    long long h = 0; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-16s %d chain(s): %6.2f counter ticks per instruction (one wave alone)\n", name, NCH, (double)h / ((double)iters * 16 * NCH));
}
int main()
{
    run<0, 1>("v_fma_f64"); run<0, 2>("v_fma_f64"); run<0, 3>("v_fma_f64"); run<0, 4>("v_fma_f64");
    run<1, 1>("v_fma_f32"); run<1, 2>("v_fma_f32"); run<1, 4>("v_fma_f32");
    run<2, 1>("v_pk_fma_f32"); run<2, 2>("v_pk_fma_f32"); run<2, 4>("v_pk_fma_f32");
    run<3, 1>("v_pk_add_f32"); run<3, 2>("v_pk_add_f32"); run<3, 4>("v_pk_add_f32");
    return 0;
}
