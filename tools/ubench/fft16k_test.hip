// Development check of fftpk16k.h: forward / inverse 16384-point workgroup transform against a float64 host FFT.
#include <hip/hip_runtime.h>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../torchfx_amd/csrc/fftpk16k.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
using namespace tfx::pk;

__global__ void __launch_bounds__(1024, 4) k(const v2f *z, const v2f *gtab, v2f *spec, v2f *back)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    v2f *L = (v2f *)smem;
    const int j = threadIdx.x;
    Tab16k tb = fill_tab16k(L + X16K_SLOTS, gtab, j);
    __syncthreads();
    const v2f Wc = {0.92387953251128675613f, 0.38268343236508977173f}, Wr = {0.70710678118654752440f, 0.70710678118654752440f};
    v2f v[16];
    for (int t = 0; t < 16; ++t) v[t] = z[j + 1024 * t];
    fft16384_fwd(v, L, tb, j, Wc, Wr);
    for (int e = 0; e < 4; ++e)
        for (int k3 = 0; k3 < 4; ++k3) spec[(j & 255) + 256 * (4 * (j >> 8) + e) + 4096 * k3] = v[4 * e + k3];
    fft16384_inv(v, L, tb, j, Wc, Wr);
    for (int t = 0; t < 16; ++t) back[j + 1024 * t] = v[t];
}

static void fft(std::vector<std::complex<double>> &a)
{
    const size_t n = a.size();
    if (n == 1) return;
    std::vector<std::complex<double>> e(n / 2), o(n / 2);
    for (size_t i = 0; i < n / 2; ++i) { e[i] = a[2 * i]; o[i] = a[2 * i + 1]; }
    fft(e); fft(o);
    for (size_t k = 0; k < n / 2; ++k) {
        const std::complex<double> w = std::polar(1.0, -2.0 * M_PI * (double)k / (double)n) * o[k];
        a[k] = e[k] + w; a[k + n / 2] = e[k] - w;
    }
}
int main()
{
    const int N = 16384;
    std::vector<v2f> z(N), tab(X16K_TABLES);
    std::vector<std::complex<double>> ref(N);
    srand(1);
    for (int i = 0; i < N; ++i) { z[i] = v2f{(float)rand() / RAND_MAX - 0.5f, (float)rand() / RAND_MAX - 0.5f}; ref[i] = {z[i].x, z[i].y}; }
    auto W = [](double num, double den) { const double a = -2.0 * M_PI * num / den; return v2f{(float)cos(a), (float)sin(a)}; };
    for (int a = 0; a < 16; ++a) for (int b = 0; b < 16; ++b) { tab[16 * a + b] = W(a * b, 4096); tab[256 + 16 * a + b] = W(a * b, 256); }
    for (int d = 0; d < 4; ++d) for (int b = 0; b < 16; ++b) { tab[512 + 16 * d + b] = W(d * b, 16384); tab[576 + 16 * d + b] = W(d * b, 1024); tab[640 + 16 * d + b] = W(d * b, 64); }
    fft(ref);
    v2f *dz, *dt, *ds, *db;
    CK(hipMalloc(&dz, N * 8)); CK(hipMalloc(&dt, X16K_TABLES * 8)); CK(hipMalloc(&ds, N * 8)); CK(hipMalloc(&db, N * 8));
    CK(hipMemcpy(dz, z.data(), N * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dt, tab.data(), X16K_TABLES * 8, hipMemcpyHostToDevice));
    const size_t shm = (size_t)(X16K_SLOTS + X16K_TABLES) * 8;
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL(k, dim3(1), dim3(1024), shm, 0, dz, dt, ds, db);
    CK(hipDeviceSynchronize());
    std::vector<v2f> spec(N), back(N);
    CK(hipMemcpy(spec.data(), ds, N * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(back.data(), db, N * 8, hipMemcpyDeviceToHost));
    double es = 0, eb = 0, ms = 0;
    int bad = -1;
    for (int i = 0; i < N; ++i) {
        const double d = std::abs(std::complex<double>(spec[i].x, spec[i].y) - ref[i]);
        if (d > es) { es = d; bad = i; }
        ms = std::max(ms, std::abs(ref[i]));
        eb = std::max(eb, std::abs(std::complex<double>(back[i].x / N, back[i].y / N) - std::complex<double>(z[i].x, z[i].y)));
    }
    printf("forward: max |err| %.3e (max |X| %.1f, worst bin %d)   round trip: max |err| %.3e   shm %zu B\n", es, ms, bad, eb, shm);
    return (es < 1e-3 * 1 && eb < 1e-5) ? 0 : 1;
}
