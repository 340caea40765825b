// Micro-benchmark: sustained rate of v_mfma_f32_32x32x2_f32 (the exact-f32 matrix op the direct
// FIR uses) on this part, as a function of resident waves per SIMD and of LDS operand traffic.
// Gives the practical ceiling the FIR kernel's 157.3 TF nominal peak should be read against.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int LDSREADS>
__global__ void __launch_bounds__(256) mfma_loop(float *out, int iters, float a0, float b0)
{
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = a0 + i * 1e-9f;
    __syncthreads();
    floatx16 acc[4];
    for (int t = 0; t < 4; ++t)
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    const float *p = lds + (threadIdx.x & 63) * 33;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (LDSREADS) b = p[2 * q + (it & 31) * 32];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (LDSREADS) a = p[1056 * t + 2 * q + (it & 31) * 33];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
            }
        }
    }
    float s = 0;
    for (int t = 0; t < 4; ++t)
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    if (s == 12345.678f) out[0] = s;
}

template <int L>
static void run(const char *name, int wg_per_cu, float *out)
{
    const int iters = 2000;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int grid = 256 * wg_per_cu;
    mfma_loop<L><<<grid, 256>>>(out, 10, 1.f, 1.f);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(a);
        mfma_loop<L><<<grid, 256>>>(out, iters, 1.f, 1.f);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    const double flop = (double)grid * 4 /*waves*/ * iters * 64.0 * 4096.0;
    printf("%-28s waves/SIMD=%d  %8.3f ms  %7.1f TFLOP/s  (%.1f %% of 157.3)\n", name, wg_per_cu, best,
           flop / (best * 1e-3) / 1e12, 100.0 * flop / (best * 1e-3) / 157.3e12);
}

int main()
{
    float *out;
    hipMalloc(&out, 4);
    for (int w : {1, 2, 4}) run<0>("mfma only", w, out);
    for (int w : {1, 2, 4}) run<1>("mfma + LDS operands (FIR mix)", w, out);
    // long run: does the clock sag over ~50 ms of sustained matrix work?
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        mfma_loop<0><<<512, 256>>>(out, 40000, 1.f, 1.f);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        const double flop = 512.0 * 4 * 40000 * 64.0 * 4096.0;
        printf("sustained %.1f ms: %7.1f TFLOP/s\n", ms, flop / (ms * 1e-3) / 1e12);
    }
    return 0;
}
