// Micro-benchmark, round 3: WHY does "one persistent wavefront per (row, segment) stream" top out at ~5.0 TB/s when a linear sweep by
// small workgroups reaches 6.2 (stream_copy.hip)?  y = 1.5 x, float32, 64 x 2.88 M and 64 x 28.8 M.  Variants of the persistent kernel:
//   burst     (round 2)  per tile: NU loads, then NU stores
//   nt        burst with nontemporal loads and stores
//   pipe      software pipeline: the loads of tile i+1 are issued BEFORE the stores of tile i (two register sets)
//   interleave  stream s walks tiles s, s + S, s + 2 S, ... (the address order of a linear sweep, persistent waves): what a
//             look-back formulation of the recurrence would produce
//   shared    G = 4 / 16 waves share one segment (wave w takes tiles w, w + G, ...): what a workgroup-level scan would read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float4 scale(float4 w) { w.x *= 1.5f; w.y *= 1.5f; w.z *= 1.5f; w.w *= 1.5f; return w; }
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int NU, int MODE>   // MODE 0 burst, 1 nt, 2 pipe, 3 interleave
__global__ void __launch_bounds__(256) streams_k(const float4 *__restrict__ x, float4 *__restrict__ y, size_t seg4, size_t n4, int nstreams)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t sid = (size_t)blockIdx.x * 4 + wave;
    if (sid >= (size_t)nstreams) return;
    constexpr size_t TILE4 = (size_t)64 * NU;
    if (MODE == 3) {
        for (size_t t = sid * TILE4; t < n4; t += (size_t)nstreams * TILE4) {
            const size_t o = t + lane;
            float4 v[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) v[u] = (o + u * 64 < n4) ? x[o + u * 64] : make_float4(0, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < NU; ++u) if (o + u * 64 < n4) y[o + u * 64] = scale(v[u]);
        }
        return;
    }
    if (MODE == 4 || MODE == 5) {
        // a GROUP of G waves (4 = one workgroup, 5: 16 = four workgroups in a row) shares one segment: wave w takes tiles w, w + G, ...
        // of it -- what a workgroup-level scan (lanes x waves) would read: G x fewer address streams, G x longer bursts
        const size_t G = MODE == 4 ? 4 : 16;
        const size_t grp = sid / G, w = sid % G;
        const size_t gseg4 = seg4 * G;
        const size_t gb = grp * gseg4, ge = (gb + gseg4 < n4) ? gb + gseg4 : n4;
        for (size_t t = gb + w * TILE4; t < ge; t += G * TILE4) {
            const size_t o = t + lane;
            float4 v[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) { floatx4 q = (o + u * 64 < ge) ? __builtin_nontemporal_load((const floatx4 *)(x + o + u * 64)) : (floatx4){0, 0, 0, 0}; v[u] = make_float4(q.x, q.y, q.z, q.w); }
#pragma unroll
            for (int u = 0; u < NU; ++u) if (o + u * 64 < ge) { const float4 r = scale(v[u]); __builtin_nontemporal_store((floatx4){r.x, r.y, r.z, r.w}, (floatx4 *)(y + o + u * 64)); }
        }
        return;
    }
    const size_t b = sid * seg4, e = (b + seg4 < n4) ? b + seg4 : n4;
    if (MODE == 2) {
        float4 v[NU], w[NU];
        size_t t = b;
        const size_t o0 = t + lane;
#pragma unroll
        for (int u = 0; u < NU; ++u) v[u] = (o0 + u * 64 < e) ? x[o0 + u * 64] : make_float4(0, 0, 0, 0);
        for (; t < e; t += TILE4) {
            const size_t o = t + lane, on = o + TILE4;
#pragma unroll
            for (int u = 0; u < NU; ++u) w[u] = (on + u * 64 < e) ? x[on + u * 64] : make_float4(0, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < NU; ++u) if (o + u * 64 < e) y[o + u * 64] = scale(v[u]);
#pragma unroll
            for (int u = 0; u < NU; ++u) v[u] = w[u];
        }
        return;
    }
    for (size_t t = b; t < e; t += TILE4) {
        const size_t o = t + lane;
        float4 v[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if (MODE == 1) { floatx4 q = (o + u * 64 < e) ? __builtin_nontemporal_load((const floatx4 *)(x + o + u * 64)) : (floatx4){0, 0, 0, 0}; v[u] = make_float4(q.x, q.y, q.z, q.w); }
            else v[u] = (o + u * 64 < e) ? x[o + u * 64] : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) if (o + u * 64 < e) {
            const float4 r = scale(v[u]);
            if (MODE == 1) __builtin_nontemporal_store((floatx4){r.x, r.y, r.z, r.w}, (floatx4 *)(y + o + u * 64));
            else y[o + u * 64] = r;
        }
    }
}

template <int UNR, bool NT>
__global__ void __launch_bounds__(256) linear_k(const float4 *__restrict__ x, float4 *__restrict__ y, size_t n4)
{
    size_t i = ((size_t)blockIdx.x * UNR) * 256 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < UNR; ++u) if (i + u * 256 < n4) {
        if (NT) { floatx4 q = __builtin_nontemporal_load((const floatx4 *)(x + i + u * 256)); q *= 1.5f; __builtin_nontemporal_store(q, (floatx4 *)(y + i + u * 256)); }
        else y[i + u * 256] = scale(x[i + u * 256]);
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
        auto run = [&](auto launch, const char *name) {
            for (int w = 0; w < 3; ++w) launch();
            double best = 1e9;
            for (int r = 0; r < 7; ++r) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); double ms = time_ms(e0, e1); if (ms < best) best = ms; }
            printf("T=%zu %-52s %8.3f ms  %.2f TB/s\n", T, name, best, 8.0 * n / best / 1e9);
        };
        run([&] { hipLaunchKernelGGL((linear_k<1, false>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, x, y, n4); }, "linear, 1 x 16 B per lane");
        run([&] { hipLaunchKernelGGL((linear_k<1, true>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, x, y, n4); }, "linear, 1 x 16 B per lane, nontemporal");
        for (int nstreams : {2048, 4096}) {
            const size_t seg4 = ((n4 + nstreams - 1) / nstreams + 63) / 64 * 64;
            char nm[96];
#define RUN(NU, MODE, label) snprintf(nm, sizeof nm, "%d streams, %d KB tiles, %s", nstreams, NU, label); \
            run([&] { hipLaunchKernelGGL((streams_k<NU, MODE>), dim3((nstreams + 3) / 4), dim3(256), 0, 0, x, y, seg4, n4, nstreams); }, nm);
            RUN(8, 0, "burst") RUN(8, 1, "nontemporal") RUN(8, 2, "pipelined") RUN(8, 3, "tile-interleaved") RUN(8, 4, "nt, 4 waves share a segment") RUN(8, 5, "nt, 16 waves share a segment") RUN(4, 4, "nt, 4 waves share a segment") RUN(4, 5, "nt, 16 waves share a segment")
            RUN(2, 0, "burst") RUN(2, 1, "nontemporal") RUN(2, 3, "tile-interleaved")
            RUN(16, 1, "nontemporal") RUN(16, 3, "tile-interleaved")
        }
        CK(hipFree(x)); CK(hipFree(y));
    }
    return 0;
}
