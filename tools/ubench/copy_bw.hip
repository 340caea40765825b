// Micro-benchmark: best float4 copy / read / write rate on this box as a function of grid size and
// per-thread unroll -- the practical HBM ceiling the streaming passes should be read against
// (MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>

template <int U>
__global__ void __launch_bounds__(256) copy_k(const float4 *__restrict__ s, float4 *__restrict__ d, size_t n)
{
    size_t i = ((size_t)blockIdx.x * U) * 256 + threadIdx.x;
    const size_t st = (size_t)gridDim.x * U * 256;
    for (; i + (U - 1) * 256 < n; i += st) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = s[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) d[i + u * 256] = v[u];
    }
}
template <int U>
__global__ void __launch_bounds__(256) read_k(const float4 *__restrict__ s, float *out, size_t n)
{
    size_t i = ((size_t)blockIdx.x * U) * 256 + threadIdx.x;
    const size_t st = (size_t)gridDim.x * U * 256;
    float acc = 0;
    for (; i + (U - 1) * 256 < n; i += st) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = s[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 1.2345f) out[0] = acc;
}
template <int U>
__global__ void __launch_bounds__(256) write_k(float4 *__restrict__ d, size_t n)
{
    size_t i = ((size_t)blockIdx.x * U) * 256 + threadIdx.x;
    const size_t st = (size_t)gridDim.x * U * 256;
    for (; i + (U - 1) * 256 < n; i += st) {
#pragma unroll
        for (int u = 0; u < U; ++u) d[i + u * 256] = make_float4(1.f, 2.f, 3.f, 4.f);
    }
}

template <typename F> static float best_ms(F f)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    float best = 1e9f;
    for (int r = 0; r < 6; ++r) {
        hipEventRecord(a);
        f();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (r && ms < best) best = ms;
    }
    return best;
}

int main()
{
    const size_t bytes = (size_t)4 << 30, n = bytes / 16;
    float4 *s, *d;
    float *out;
    hipMalloc(&s, bytes);
    hipMalloc(&d, bytes);
    hipMalloc(&out, 4);
    hipMemset(s, 1, bytes);
    hipMemset(d, 0, bytes);
    for (int wgs_per_cu : {2, 4, 8, 16, 32, 0}) {
        const unsigned g1 = wgs_per_cu ? 256u * wgs_per_cu : (unsigned)(n / 256);
        const unsigned g4 = wgs_per_cu ? 256u * wgs_per_cu : (unsigned)(n / 1024);
        const float c1 = best_ms([&] { copy_k<1><<<g1, 256>>>(s, d, n); });
        const float c4 = best_ms([&] { copy_k<4><<<g4, 256>>>(s, d, n); });
        const float r4 = best_ms([&] { read_k<4><<<g4, 256>>>(s, out, n); });
        const float w4 = best_ms([&] { write_k<4><<<g4, 256>>>(d, n); });
        printf("WGs/CU %-10s copy U1 %5.2f TB/s  copy U4 %5.2f TB/s (r+w bytes)  read U4 %5.2f  write U4 %5.2f\n",
               wgs_per_cu ? std::to_string(wgs_per_cu).c_str() : "one-shot", 2.0 * bytes / (c1 * 1e-3) / 1e12,
               2.0 * bytes / (c4 * 1e-3) / 1e12, bytes / (r4 * 1e-3) / 1e12, bytes / (w4 * 1e-3) / 1e12);
    }
    return 0;
}
