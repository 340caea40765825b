// Where does the direct-FIR kernel's time go?  Runs the library's own kernel (included from
// torchfx_amd/csrc/fir.hip) with the output stores and/or the global loads compiled out.
#include "../../torchfx_amd/csrc/fir.hip"
namespace tfx {
void set_last_error(const std::string &) {}
bool prof_on() { return false; }
void prof_begin(const char *, hipStream_t) {}
void prof_end(hipStream_t) {}
}  // namespace tfx
using namespace tfx;

template <int FIR_KC, int DBG>
static void run(const char *name, const float *x, float *y, const float *k, int64_t C, int64_t T, int K)
{
    constexpr int XW = FIR_NOUT + FIR_KC + 32;
    constexpr int XW_PAD = XW + (XW >> 5) + 1;
    const size_t shmem = (((XW_PAD + 3) & ~3) + 31 + FIR_KC + 33) * sizeof(float);
    hipFuncSetAttribute((const void *)fir_direct_mfma_kernel<FIR_KC, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    const int64_t tiles = (T + FIR_NOUT - 1) / FIR_NOUT;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((fir_direct_mfma_kernel<FIR_KC, DBG>), dim3((unsigned)(C * tiles)), dim3(256), shmem, 0, x, y, k, C, T, K,
                           (K + FIR_KC - 1) / FIR_KC, tiles);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
    }
    printf("%-36s %7.3f ms  %6.1f TFLOP/s\n", name, best, 2.0 * K * C * T / (best * 1e-3) / 1e12);
}

int main()
{
    const int64_t C = 64, T = 2880000;
    const int K = 1024;
    float *x, *y, *k;
    hipMalloc(&x, C * T * 4);
    hipMalloc(&y, C * T * 4);
    hipMalloc(&k, 4096 * 4);
    hipMemset(x, 0, C * T * 4);
    hipMemset(k, 0, 4096 * 4);
    run<1024, 0>("full kernel", x, y, k, C, T, K);
    run<1024, 1>("no output stores", x, y, k, C, T, K);
    run<1024, 2>("no global loads", x, y, k, C, T, K);
    run<1024, 3>("no loads, no stores (LDS + MFMA)", x, y, k, C, T, K);
    run<1024, 0>("full kernel, K=4096", x, y, k, C, T, 4096);
    run<1024, 0>("K=101, 1024-tap chunks", x, y, k, C, T, 101);
    run<128, 0>("K=101, 128-tap chunks", x, y, k, C, T, 101);
    run<512, 0>("K=400, 512-tap chunks", x, y, k, C, T, 400);
    run<128, 0>("K=400, 128-tap chunks", x, y, k, C, T, 400);
    return 0;
}
