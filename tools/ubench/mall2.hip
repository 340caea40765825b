// Micro-benchmark: steady-state bandwidth of a buffer that fits L2 (16 MB), the Infinity Cache (64-192 MB)
// or neither (2 GB), swept REPS times inside one launch (no per-launch ramp/tail in the number).
//   read: float4 loads;  rmw: in-place read-modify-write (like the overlap-save row pass)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) sweep(float4 *__restrict__ p, size_t n4, int reps, float *sink)
{
    // every workgroup owns a contiguous stripe set: chunk c = blockIdx + gridDim * i, 4 KiB per chunk
    float acc = 0.f;
    const size_t nchunks = n4 / 256;
    for (int r = 0; r < reps; ++r) {
        for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
            float4 v = p[c * 256 + threadIdx.x];
            if (MODE == 0) acc += v.x + v.y + v.z + v.w;
            else { v.x += 1.f; p[c * 256 + threadIdx.x] = v; }
        }
    }
    if (MODE == 0 && acc == 1.2345f) sink[0] = acc;
}

int main()
{
    float *sink; CK(hipMalloc(&sink, 4));
    const size_t maxb = (size_t)4 << 30;
    float4 *buf; CK(hipMalloc(&buf, maxb)); CK(hipMemset(buf, 0, maxb));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (size_t mb : {8, 16, 24, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 4096}) {
        const size_t bytes = mb << 20, n4 = bytes / 16;
        const int reps = (int)((8ull << 30) / bytes) < 2 ? 2 : (int)((8ull << 30) / bytes);   // ~8 GiB of traffic per measurement
        for (int grid : {2048}) {
            float ms0, ms1;
            hipLaunchKernelGGL(sweep<0>, dim3(grid), dim3(256), 0, 0, buf, n4, 2, sink);
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(sweep<0>, dim3(grid), dim3(256), 0, 0, buf, n4, reps, sink);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms0, a, b));
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(sweep<1>, dim3(grid), dim3(256), 0, 0, buf, n4, reps, sink);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms1, a, b));
            printf("%5zu MB x %4d sweeps: read %6.2f TB/s | in-place rmw %6.2f TB/s (r+w bytes)\n", mb, reps,
                   (double)bytes * reps / ms0 / 1e9, 2.0 * bytes * reps / ms1 / 1e9);
        }
    }
    return 0;
}
