// Micro-benchmark: what does the column passes' ACCESS PATTERN cost, independent of the FFT work?
// A column-pass workgroup touches R row segments of W bytes each, R*stride apart (four-step FFT,
// N = R * N2, stride = N2 samples).  Variants: segment width 128 / 256 / 512 B (R = 256 / 128 / 64
// at constant 32 KiB per frame per workgroup), read only / write only / read->write (real in,
// complex out like pass A; complex in, real out like pass C).
//   hipcc --offload-arch=gfx950 -O3 seg_pattern.hip -o seg_pattern && ./seg_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// mode 0: read real segments (2 frames), 1: write real segments (2 frames), 2: read real -> write complex rows,
// 3: read complex rows -> write real segments
template <int WD, int MODE>
__global__ void __launch_bounds__(256) seg_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                  float2 *__restrict__ T, int64_t N, int R, int64_t frame_stride, float *sink)
{
    constexpr int RQ = 256 / WD;                 // rows covered per sweep of the workgroup
    const int tid = threadIdx.x, col = tid % WD, q = tid / WD;
    const int64_t N2 = N / R;
    const int ncb = (int)(N2 / WD);
    const int64_t pair = blockIdx.x / ncb;
    const int cb = blockIdx.x % ncb;
    const int64_t n2 = (int64_t)cb * WD + col;
    const float *xa = x + (2 * pair) * frame_stride, *xb = xa + frame_stride;
    float *ya = y + (2 * pair) * frame_stride, *yb = ya + frame_stride;
    float2 *Tp = T + pair * N;
    float acc = 0.f;
    if (MODE == 0 || MODE == 2) {
        float2 v[32];                             // up to 32 rows per thread (R / RQ <= 32)
        const int per = R / RQ;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (i < per) {
                const int64_t n = (int64_t)(q + RQ * i) * N2 + n2;
                v[i] = make_float2(xa[n], xb[n]);
            }
        }
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 32; ++i) if (i < per) acc += v[i].x + v[i].y;
            if (acc == 1.2345f) sink[0] = acc;
        } else {
#pragma unroll
            for (int i = 0; i < 32; ++i) if (i < per) Tp[(int64_t)(q + RQ * i) * N2 + n2] = v[i];
        }
    } else if (MODE == 1) {
        const int per = R / RQ;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (i < per) {
                const int64_t n = (int64_t)(q + RQ * i) * N2 + n2;
                ya[n] = (float)i; yb[n] = (float)col;
            }
        }
    } else {
        float2 v[32];
        const int per = R / RQ;
#pragma unroll
        for (int i = 0; i < 32; ++i) if (i < per) v[i] = Tp[(int64_t)(q + RQ * i) * N2 + n2];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (i < per) {
                const int64_t n = (int64_t)(q + RQ * i) * N2 + n2;
                ya[n] = v[i].x; yb[n] = v[i].y;
            }
        }
    }
}

// plain streaming references on the same buffers (one-shot grid, 16 B per lane)
__global__ void __launch_bounds__(256) lin_read(const float4 *__restrict__ p, float *sink)
{
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    float4 a = p[i], b = p[i + 256], c = p[i + 512], d = p[i + 768];
    float s = a.x + b.y + c.z + d.w;
    if (s == 1.2345f) sink[0] = s;
}
__global__ void __launch_bounds__(256) lin_copy(const float4 *__restrict__ p, float4 *__restrict__ o)
{
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    float4 a = p[i], b = p[i + 256], c = p[i + 512], d = p[i + 768];
    o[i] = a; o[i + 256] = b; o[i + 512] = c; o[i + 768] = d;
}

template <int WD, int MODE>
static double run(const float *x, float *y, float2 *T, int64_t N, int R, int64_t fs, int64_t pairs, float *sink)
{
    const int64_t N2 = N / R;
    const unsigned grid = (unsigned)(pairs * (N2 / WD));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((seg_kernel<WD, MODE>), dim3(grid), dim3(256), 0, 0, x, y, T, N, R, fs, sink);
    CK(hipEventRecord(a));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((seg_kernel<WD, MODE>), dim3(grid), dim3(256), 0, 0, x, y, T, N, R, fs, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms / reps;
}

int main()
{
    const int64_t N = 1 << 20, pairs = 256;            // 256 pairs: 2 GiB of real frames, 2 GiB workspace
    const int64_t fs = N;                              // frames back to back
    float *x, *y, *sink; float2 *T;
    CK(hipMalloc(&x, 2 * pairs * fs * 4)); CK(hipMalloc(&y, 2 * pairs * fs * 4)); CK(hipMalloc(&T, pairs * N * 8)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(x, 0, 2 * pairs * fs * 4)); CK(hipMemset(T, 0, pairs * N * 8));
    const double real_gb = 2.0 * pairs * N * 4 / 1e9, cpx_gb = pairs * N * 8.0 / 1e9;
    printf("N = 2^20, %lld frame pairs: %.2f GB real frames, %.2f GB complex workspace\n", (long long)pairs, real_gb, cpx_gb);
    {
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        const unsigned g = (unsigned)(2 * pairs * fs * 4 / 16384);
        hipLaunchKernelGGL(lin_read, dim3(g), dim3(256), 0, 0, (const float4 *)x, sink);
        CK(hipEventRecord(a));
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(lin_read, dim3(g), dim3(256), 0, 0, (const float4 *)x, sink);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("linear read  %.3f ms  %.2f TB/s\n", ms / 5, real_gb / (ms / 5));
        hipLaunchKernelGGL(lin_copy, dim3(g), dim3(256), 0, 0, (const float4 *)x, (float4 *)y);
        CK(hipEventRecord(a));
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(lin_copy, dim3(g), dim3(256), 0, 0, (const float4 *)x, (float4 *)y);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms, a, b));
        printf("linear copy  %.3f ms  %.2f TB/s (r+w)\n", ms / 5, 2 * real_gb / (ms / 5));
    }
#define ROW(WD, R)                                                                                         \
    {                                                                                                      \
        const double t0 = run<WD, 0>(x, y, T, N, R, fs, pairs, sink), t1 = run<WD, 1>(x, y, T, N, R, fs, pairs, sink); \
        const double t2 = run<WD, 2>(x, y, T, N, R, fs, pairs, sink), t3 = run<WD, 3>(x, y, T, N, R, fs, pairs, sink); \
        printf("seg %4d B x %3d rows (stride %6lld B): read %.3f ms %.2f TB/s | write %.3f ms %.2f TB/s | A-like r->w %.3f ms %.2f TB/s | C-like r->w %.3f ms %.2f TB/s\n", \
               WD * 4, R, (long long)(N / R * 4), t0, real_gb / t0, t1, real_gb / t1, t2, (real_gb + cpx_gb) / t2, t3, (real_gb + cpx_gb) / t3); \
    }
    ROW(32, 256)
    ROW(64, 128)
    ROW(128, 64)
    ROW(16, 256)
    return 0;
}
