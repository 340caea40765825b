// Micro-benchmark: issue rate of the packed-float32 VALU ops with and without operand modifiers.
// Each wave runs 8 independent dependency chains of one instruction form; result = wave-instructions
// per clock per SIMD (1/4 = one instruction every 4 clocks).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int V>
__global__ void __launch_bounds__(256) k(v2f *out, int iters, v2f seed)
{
    v2f a[8], b = seed;
    for (int i = 0; i < 8; ++i) a[i] = seed * (float)(i + 1 + threadIdx.x);
    const v2f sc = {1.0001f, 0.9999f};
    for (int it = 0; it < iters; ++it) {
#define ONE(i)                                                                                            \
        if (V == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));                                  \
        if (V == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1 op_sel_hi:[0,1,1]" : "+v"(a[i]) : "v"(b));                \
        if (V == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %1 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "+v"(a[i]) : "v"(b)); \
        if (V == 3) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "+v"(a[i]) : "v"(b)); \
        if (V == 4) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "+v"(a[i]) : "v"(b)); \
        if (V == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                      \
        if (V == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "s"(sc));                                     \
        if (V == 7) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i].x) : "v"(b.x));                                 \
        if (V == 8) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[0,1]" : "+v"(a[i]) : "v"(b));                      \
        if (V == 9) asm volatile("v_pk_fma_f32 %0, %0, %1, %1 neg_hi:[0,1,0]" : "+v"(a[i]) : "v"(b));
        REP8(ONE) REP8(ONE) REP8(ONE) REP8(ONE)
    }
    v2f s = a[0];
    for (int i = 1; i < 8; ++i) s += a[i];
    if (s.x == 1.2345f) out[0] = s;
}
template <int V> static void run(const char *name)
{
    v2f *out; CK(hipMalloc(&out, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000, grid = 256 * 8;            // 8 blocks of 4 waves per CU: 8 waves per SIMD
    v2f seed = {1.0f, 0.5f};
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, out, 100, seed);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, out, iters, seed);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double winstr = (double)grid * 4 * iters * 32;        // wave-instructions
    const double per_simd_per_s = winstr / (256.0 * 4) / (ms * 1e-3);
    printf("%-64s %8.3f ms  %.3f G wave-instr/s/SIMD  = one per %.2f clk @2.4 GHz\n", name, ms, per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s);
}
int main()
{
    run<0>("v_pk_fma_f32 plain");
    run<1>("v_pk_fma_f32 op_sel_hi:[0,1,1] (broadcast .x of src0)");
    run<2>("v_pk_fma_f32 op_sel:[1,0,0] op_sel_hi:[0,1,1] (swap src0)");
    run<9>("v_pk_fma_f32 neg_hi:[0,1,0]");
    run<3>("v_pk_mul_f32 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]");
    run<8>("v_pk_mul_f32 op_sel_hi:[0,1]");
    run<6>("v_pk_mul_f32 with an SGPR-pair operand");
    run<5>("v_pk_add_f32 plain");
    run<4>("v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]");
    run<7>("v_fma_f32 (scalar float)");
    return 0;
}
