// Micro-benchmark: LDS cost of the exchange patterns of the 4096-point workgroup FFT (olslds.hip), 4 workgroups of 256
// threads per CU (38.9 KB of LDS each), nothing but the LDS instructions and the barriers.  Reports clocks per CU for what
// one workgroup does per exchange.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int V>
__global__ void __launch_bounds__(256, 4) k(v2f *out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    v2f *l2 = (v2f *)smem;
    const int j = threadIdx.x, kb = j & 15, jh = j >> 4;
    v2f v[16];
    for (int i = 0; i < 16; ++i) v[i] = v2f{(float)(i + j), (float)(i - j)};
    for (int i = j; i < 4864; i += 256) l2[i] = v2f{0.f, 0.f};
    __syncthreads();
    typedef const char __attribute__((address_space(3))) *lds_ptr;
#define LA(p) ((unsigned)(uintptr_t)(lds_ptr)(const char *)(p))
    const unsigned w64 = LA(l2 + j + jh), w128 = LA(l2 + 2 * (j + jh)), r2 = LA(l2 + kb + 272 * jh), r3 = LA(l2 + 17 * j),
                   r2p = LA(l2 + 2 * kb + (jh & 1) + 544 * (jh >> 1)), tw = LA(l2 + 4352 + jh), twp = LA(l2 + 4352 + 18 * jh),
                   r128 = LA(l2 + 16 * j + 2 * (j >> 2));
    for (int it = 0; it < iters; ++it) {
        if (V == 0 || V == 2) {
#pragma unroll
            for (int t = 0; t < 16; ++t) asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(w64), "v"(v[t]), "n"(t * 272 * 8) : "memory");
        }
        if (V == 1 || V == 3) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                v4f q = {v[2 * t].x, v[2 * t].y, v[2 * t + 1].x, v[2 * t + 1].y};
                asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(w128), "v"(q), "n"(t * 544 * 8) : "memory");
            }
        }
        if (V <= 3) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __syncthreads(); }
        if (V == 0 || V == 4) {
#pragma unroll
            for (int t = 0; t < 16; ++t) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[t]) : "v"(r2), "n"(t * 17 * 8) : "memory");
        }
        if (V == 1) {
#pragma unroll
            for (int t = 0; t < 16; ++t) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[t]) : "v"(r2p), "n"(t * 34 * 8) : "memory");
        }
        if (V == 8) {
#pragma unroll
            for (int t = 0; t < 16; ++t) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[t]) : "v"(r3), "n"(t * 8) : "memory");
        }
        if (V == 5) {
#pragma unroll
            for (int t = 0; t < 16; ++t) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[t]) : "v"(tw), "n"(t * 16 * 8) : "memory");
        }
        if (V == 6) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                v4f q;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q) : "v"(twp), "n"(t * 16) : "memory");
                v[2 * t] = v2f{q.x, q.y}; v[2 * t + 1] = v2f{q.z, q.w};
            }
        }
        if (V == 7) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                v4f q;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q) : "v"(r128), "n"(t * 16) : "memory");
                v[2 * t] = v2f{q.x, q.y}; v[2 * t + 1] = v2f{q.z, q.w};
            }
        }
        if (V == 0 || V == 1 || V >= 4) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __syncthreads(); }
        if (V >= 4) {
#pragma unroll
            for (int t = 0; t < 16; ++t) asm volatile("" : "+v"(v[t]));
        }
    }
    v2f s = v[0];
    for (int i = 1; i < 16; ++i) s += v[i];
    if (s.x == 1.2345f) out[0] = s;
}
template <int V> static void run(const char *name)
{
    v2f *out; CK(hipMalloc(&out, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4000, grid = 256 * 4;
    const size_t shm = 4864 * 8;
    CK(hipFuncSetAttribute((const void *)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), shm, 0, out, 50);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), shm, 0, out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // 4 workgroups per CU: time per iteration / 4 = CU time per workgroup-iteration
    const double ns = ms * 1e6 / iters / 4;
    printf("%-72s %8.3f ms  %7.1f ns per workgroup-step per CU = %6.0f clk @2.4 GHz\n", name, ms, ns, ns * 2.4);
}
int main()
{
    run<0>("16 ds_write_b64 + barrier + 16 ds_read_b64 (stage-2 pattern) + barrier");
    run<1>("8 ds_write_b128 (pair layout) + barrier + 16 ds_read_b64 + barrier");
    run<2>("16 ds_write_b64 + barrier");
    run<3>("8 ds_write_b128 + barrier");
    run<4>("16 ds_read_b64 stride 17 (stage 2) + barrier");
    run<8>("16 ds_read_b64 stride 1 (stage 3) + barrier");
    run<7>("8 ds_read_b128 contiguous + barrier");
    run<5>("16 ds_read_b64 twiddles (4 addresses per wave) + barrier");
    run<6>("8 ds_read_b128 twiddles (4 addresses per wave) + barrier");
    return 0;
}
