// Micro-benchmark (feasibility of an L2-resident overlap-save): one persistent workgroup per CU; the workgroups of
// an XCD (found at run time from HW_REG_XCC_ID) process frame pairs of N = 2^18 through three phases with the real
// access patterns but NO arithmetic --
//   A  strided reads of two real frames (256 rows x 32-column blocks)  -> complex workspace W[xcd] (2 MB, stays in that XCD's L2)
//   B  rows of W read-modify-written (+ optionally a 1 MB spectrum read)
//   C  columns of W -> two real output frames
// separated by XCD-local barriers (L2-local atomics, L1-bypassing loads; no agent-scope release, so the dirty
// workspace lines are never written back).  Reports us per pair per XCD and lets rocprofv3 count the HBM bytes:
// if the workspace really lives in L2 the traffic is x + y (+ spectrum), ~9-14 B/sample instead of ~26.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int N1 = 256, N2 = 1024, NN = N1 * N2;       // 2^18
constexpr int CB = 32;

struct Ctl {
    unsigned reg[8];         // workgroups registered per XCD
    unsigned ready;          // all registered
    unsigned bar[8];         // monotonic XCD barrier counters
    unsigned err;
};

__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 ld_nt(const float2 *p)           // L1-bypassing 8-byte load
{
    const v2f v = __builtin_nontemporal_load((const v2f *)p);
    return make_float2(v.x, v.y);
}
__device__ __forceinline__ unsigned ld_l2(const unsigned *p)      // bypass L1
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// barrier among the G workgroups of one XCD: counter lives in that XCD's L2
__device__ __forceinline__ void xcd_barrier(Ctl *ctl, unsigned xcc, unsigned G, unsigned &gen)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(&ctl->bar[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        gen += G;
        unsigned spins = 0;
        while (ld_l2(&ctl->bar[xcc]) < gen) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { ctl->err = 1; break; }
        }
    }
    __syncthreads();
}

template <bool SPECTRUM>
__global__ void __launch_bounds__(512, 2) pipeline(const float *__restrict__ x, float *__restrict__ y, float2 *W,
                                                   const float2 *__restrict__ H, Ctl *ctl, int pairs_per_xcd, int64_t hop,
                                                   int nblocks)
{
    extern __shared__ char smem[];                       // only to force one workgroup per CU
    __shared__ unsigned s_slot, s_G;
    const unsigned xcc = xcc_id();
    if (threadIdx.x == 0) {
        s_slot = atomicAdd(&ctl->reg[xcc], 1u);
        __threadfence();
        atomicAdd(&ctl->ready, 1u);
        unsigned spins = 0;
        while (__hip_atomic_load(&ctl->ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nblocks) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > (1u << 22)) { ctl->err = 2; break; }
        }
        __threadfence();
        s_G = __hip_atomic_load(&ctl->reg[xcc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const unsigned slot = s_slot, G = s_G;
    unsigned gen = 0;
    float2 *Wx = W + (size_t)xcc * NN;
    const int tid = threadIdx.x, col = tid & 31, q = tid >> 5;          // q < 16
    for (int it = 0; it < pairs_per_xcd; ++it) {
        const int64_t pair = (int64_t)it * 8 + xcc;
        const float *xa = x + pair * 2 * hop, *xb = xa + hop;
        float *ya = y + pair * 2 * hop, *yb = ya + hop;
        // ---- A: 32 column blocks of 32 columns; rows q + 16 t
        for (unsigned t = slot; t < N2 / CB; t += G) {
            const int n2 = t * CB + col;
            float2 v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = make_float2(xa[(int64_t)(q + 16 * r) * N2 + n2], xb[(int64_t)(q + 16 * r) * N2 + n2]);
#pragma unroll
            for (int r = 0; r < 16; ++r) Wx[(size_t)(q + 16 * r) * N2 + n2] = v[r];
        }
        xcd_barrier(ctl, xcc, G, gen);
        // ---- B: 256 rows, 8 rows per task (one wave per row), in place; L1-bypassing loads
        for (unsigned t = slot; t < N1 / 8; t += G) {
            const int row = t * 8 + (tid >> 6), lane = tid & 63;
            float2 *base = Wx + (size_t)row * N2;
            float2 v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = ld_nt(&base[lane + 64 * r]);
            if (SPECTRUM) {
#pragma unroll
                for (int r = 0; r < 8; ++r) { const float2 h = H[(size_t)(row >> 1) * N2 + lane + 64 * (r + 8 * (row & 1))]; v[r].x += h.x; v[r + 8].y += h.y; }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) base[lane + 64 * r] = make_float2(v[r].y, v[r].x);
        }
        xcd_barrier(ctl, xcc, G, gen);
        // ---- C: columns -> output frames (first 3/4 of the rows: the valid part)
        for (unsigned t = slot; t < N2 / CB; t += G) {
            const int n2 = t * CB + col;
            float2 v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = ld_nt(&Wx[(size_t)(q + 16 * r) * N2 + n2]);
#pragma unroll
            for (int r = 0; r < 12; ++r) { ya[(int64_t)(q + 16 * r) * N2 + n2] = v[r].x; yb[(int64_t)(q + 16 * r) * N2 + n2] = v[r].y; }
        }
        xcd_barrier(ctl, xcc, G, gen);
    }
}

int main(int argc, char **argv)
{
    const int pairs_per_xcd = argc > 1 ? atoi(argv[1]) : 400;
    const int64_t hop = 196608;                                  // 3/4 N: valid samples per frame
    const int64_t npairs = (int64_t)pairs_per_xcd * 8;
    const size_t xbytes = (size_t)(npairs * 2 * hop + NN) * 4;
    float *x, *y; float2 *W, *H; Ctl *ctl;
    CK(hipMalloc(&x, xbytes)); CK(hipMalloc(&y, xbytes)); CK(hipMalloc(&W, (size_t)8 * NN * 8)); CK(hipMalloc(&H, (size_t)NN * 4));
    CK(hipMalloc(&ctl, sizeof(Ctl)));
    CK(hipMemset(x, 0, xbytes)); CK(hipMemset(H, 0, (size_t)NN * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int nblocks = 256;
    for (int spec = 0; spec < 2; ++spec) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(ctl, 0, sizeof(Ctl)));
            CK(hipEventRecord(a));
            if (spec) hipLaunchKernelGGL(pipeline<true>, dim3(nblocks), dim3(512), 81920, 0, x, y, W, H, ctl, pairs_per_xcd, hop, nblocks);
            else hipLaunchKernelGGL(pipeline<false>, dim3(nblocks), dim3(512), 81920, 0, x, y, W, H, ctl, pairs_per_xcd, hop, nblocks);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            Ctl h; CK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
            const double samples = (double)npairs * 2 * hop;
            printf("spectrum=%d rep %d: %.3f ms for %lld pairs (%d per XCD): %.2f us per pair per XCD, %.1f Gsamples/s, err=%u, groups:", spec, rep, ms,
                   (long long)npairs, pairs_per_xcd, ms * 1e3 / pairs_per_xcd, samples / ms / 1e6, h.err);
            for (int i = 0; i < 8; ++i) printf(" %u", h.reg[i]);
            printf("\n");
        }
    }
    printf("bytes by construction per pair: x 2.10 MB read, y 1.57 MB written (+ spectrum 1.05 MB read); workspace 2.10 MB per XCD\n");
    return 0;
}
