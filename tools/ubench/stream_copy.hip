// Micro-benchmark: does the cascade kernel's access pattern (one persistent wavefront per (row, time segment) stream,
// each walking its segment in 16 KB tiles) cost HBM bandwidth against a linear sweep?  y = 1.5 * x, float32.
//   linear     : non-persistent workgroups, 16 B per lane x UNR, launched in address order (what an elementwise op does)
//   streams    : nstreams persistent waves, stream s owns [s * seg, (s + 1) * seg), tile = 64 lanes x 16 B x NU
//                (NU = 16 -> 16 KB like the float64 cascade's LC = 64 tile; 8 -> 8 KB like LC = 32; 32 / 64 -> larger)
//   wgstreams  : one stream per WORKGROUP of 4 waves, tile = 4 x (64 x 16 B x NU): the 4 waves touch one contiguous chunk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int UNR>
__global__ void __launch_bounds__(256) linear_k(const float4 *__restrict__ x, float4 *__restrict__ y, size_t n4)
{
    size_t i = ((size_t)blockIdx.x * UNR) * 256 + threadIdx.x;
    float4 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) if (i + u * 256 < n4) v[u] = x[i + u * 256];
#pragma unroll
    for (int u = 0; u < UNR; ++u) if (i + u * 256 < n4) { float4 w = v[u]; w.x *= 1.5f; w.y *= 1.5f; w.z *= 1.5f; w.w *= 1.5f; y[i + u * 256] = w; }
}

template <int NU, bool WG>
__global__ void __launch_bounds__(256) streams_k(const float4 *__restrict__ x, float4 *__restrict__ y, size_t seg4, size_t n4, int nstreams)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t sid = WG ? blockIdx.x : (size_t)blockIdx.x * 4 + wave;
    if (sid >= (size_t)nstreams) return;
    const size_t b = sid * seg4, e = (b + seg4 < n4) ? b + seg4 : n4;
    constexpr size_t TILE4 = (size_t)64 * NU * (WG ? 4 : 1);
    for (size_t t = b; t < e; t += TILE4) {
        const size_t o = t + (WG ? (size_t)wave * 64 * NU : 0) + lane;
        float4 v[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) v[u] = (o + u * 64 < e) ? x[o + u * 64] : make_float4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < NU; ++u) { float4 w = v[u]; w.x *= 1.5f; w.y *= 1.5f; w.z *= 1.5f; w.w *= 1.5f; if (o + u * 64 < e) y[o + u * 64] = w; }
    }
}

static double time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main()
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (size_t T : {(size_t)2880000, (size_t)28800000}) {
        const size_t C = 64, n = C * T, n4 = n / 4;
        float4 *x, *y; CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4));
        CK(hipMemset(x, 0, n * 4));
        auto report = [&](const char *name, double ms) { printf("T=%zu %-44s %8.3f ms  %.2f TB/s\n", T, name, ms, 8.0 * n / ms / 1e9); };
        auto run = [&](auto launch, const char *name) {
            for (int w = 0; w < 3; ++w) launch();
            double best = 1e9;
            for (int r = 0; r < 7; ++r) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); double ms = time_ms(e0, e1); if (ms < best) best = ms; }
            report(name, best);
        };
        run([&] { hipLaunchKernelGGL(linear_k<4>, dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, 0, x, y, n4); }, "linear, 4 x 16 B per lane");
        run([&] { hipLaunchKernelGGL(linear_k<1>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, x, y, n4); }, "linear, 1 x 16 B per lane");
        run([&] { hipLaunchKernelGGL(linear_k<16>, dim3((unsigned)((n4 + 4095) / 4096)), dim3(256), 0, 0, x, y, n4); }, "linear, 16 x 16 B per lane");
        for (int nstreams : {2048, 3072, 4096, 8192}) {
            const size_t seg4 = ((n4 + nstreams - 1) / nstreams + 63) / 64 * 64;
            char nm[96];
            snprintf(nm, sizeof nm, "%d wave streams, 8 KB tiles", nstreams);
            run([&] { hipLaunchKernelGGL((streams_k<8, false>), dim3((nstreams + 3) / 4), dim3(256), 0, 0, x, y, seg4, n4, nstreams); }, nm);
            snprintf(nm, sizeof nm, "%d wave streams, 16 KB tiles", nstreams);
            run([&] { hipLaunchKernelGGL((streams_k<16, false>), dim3((nstreams + 3) / 4), dim3(256), 0, 0, x, y, seg4, n4, nstreams); }, nm);
            snprintf(nm, sizeof nm, "%d wave streams, 32 KB tiles", nstreams);
            run([&] { hipLaunchKernelGGL((streams_k<32, false>), dim3((nstreams + 3) / 4), dim3(256), 0, 0, x, y, seg4, n4, nstreams); }, nm);
        }
        for (int nstreams : {512, 1024, 2048}) {
            const size_t seg4 = ((n4 + nstreams - 1) / nstreams + 63) / 64 * 64;
            char nm[96];
            snprintf(nm, sizeof nm, "%d workgroup streams, 4 x 8 KB tiles", nstreams);
            run([&] { hipLaunchKernelGGL((streams_k<8, true>), dim3(nstreams), dim3(256), 0, 0, x, y, seg4, n4, nstreams); }, nm);
            snprintf(nm, sizeof nm, "%d workgroup streams, 4 x 16 KB tiles", nstreams);
            run([&] { hipLaunchKernelGGL((streams_k<16, true>), dim3(nstreams), dim3(256), 0, 0, x, y, seg4, n4, nstreams); }, nm);
        }
        CK(hipFree(x)); CK(hipFree(y));
    }
    return 0;
}
