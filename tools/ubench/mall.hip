// Micro-benchmark: does the Infinity Cache keep a buffer written by one kernel for the next one?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void wr(float4 *p, size_t n, float v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) p[i] = make_float4(v, v, v, v);
}
__global__ void rd(const float4 *p, size_t n, float *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    float acc = 0;
    for (; i < n; i += st) { float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) out[0] = acc;
}
__global__ void rw(float4 *p, size_t n) {   // in-place read-modify-write (like the row pass)
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) { float4 v = p[i]; v.x += 1.f; p[i] = v; }
}
int main() {
    float *out; hipMalloc(&out, 4);
    size_t big = (size_t)2 << 30;
    float4 *flush; hipMalloc(&flush, big);
    float4 *buf; hipMalloc(&buf, big);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (size_t mb : {8, 16, 32, 64, 128, 192, 256, 384, 512, 1024, 2048}) {
        size_t bytes = mb << 20, n = bytes / 16;
        float t_hot = 0, t_cold = 0, t_rw_hot = 0, t_w = 0;
        for (int rep = 0; rep < 5; ++rep) {
            float ms;
            // hot: write then read immediately
            hipEventRecord(a); wr<<<2048, 256>>>(buf, n, 1.f); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b); t_w += ms;
            hipEventRecord(a); rd<<<2048, 256>>>(buf, n, out); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b); t_hot += ms;
            // rw in place right after a write
            wr<<<2048, 256>>>(buf, n, 1.f);
            hipEventRecord(a); rw<<<2048, 256>>>(buf, n); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b); t_rw_hot += ms;
            // cold: flush caches with a 2 GB write, then read
            wr<<<2048, 256>>>(flush, big / 16, 2.f);
            hipEventRecord(a); rd<<<2048, 256>>>(buf, n, out); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b); t_cold += ms;
        }
        printf("%5zu MB: write %7.2f TB/s | read-after-write %7.2f TB/s | read cold %7.2f TB/s | rmw-after-write %7.2f TB/s (r+w bytes)\n",
               mb, bytes / (t_w / 5 * 1e-3) / 1e12, bytes / (t_hot / 5 * 1e-3) / 1e12, bytes / (t_cold / 5 * 1e-3) / 1e12,
               2.0 * bytes / (t_rw_hot / 5 * 1e-3) / 1e12);
    }
    return 0;
}
