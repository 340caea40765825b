// Micro-benchmark, round 3: a radix-16 butterfly in registers, three ways of writing it (profiles/r03_experiments.txt):
//   struct   float2 structs, the compiler packs what it can (v_mov / v_pk_mov re-pairing for the mixed-sign lanes)
//   vector   ext_vector_type(2) + shufflevector
//   asm      the mixed-sign adds / fmas as single v_pk_* instructions with op_sel / neg modifiers (non-volatile inline asm)
// ITER dependent butterflies per thread, 256 threads x 4 waves per SIMD resident; reports ns per butterfly-wave and the VALU count.
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float2 cpx;

// ---- variant 0: current code
__device__ __forceinline__ cpx cadd(cpx a, cpx b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cpx csub(cpx a, cpx b) { return make_float2(a.x - b.x, a.y - b.y); }
template <bool INV> __device__ __forceinline__ void dft4(cpx &a0, cpx &a1, cpx &a2, cpx &a3)
{
    const cpx s02 = cadd(a0, a2), d02 = csub(a0, a2), s13 = cadd(a1, a3), d13 = csub(a1, a3);
    const cpx id = INV ? make_float2(-d13.y, d13.x) : make_float2(d13.y, -d13.x);
    a0 = cadd(s02, s13); a2 = csub(s02, s13); a1 = cadd(d02, id); a3 = csub(d02, id);
}
template <bool INV> __device__ __forceinline__ void dft16(cpx (&v)[16])
{
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
#pragma unroll
    for (int t1 = 0; t1 < 4; ++t1) dft4<INV>(v[t1], v[t1 + 4], v[t1 + 8], v[t1 + 12]);
    auto tw = [&](cpx &x, float c, float sn) { const float s_ = INV ? -sn : sn; x = make_float2(x.x * c + x.y * s_, x.y * c - x.x * s_); };
    tw(v[1 + 4], C1, S1);  tw(v[1 + 8], R2, R2);  tw(v[1 + 12], S1, C1);
    tw(v[2 + 4], R2, R2);  tw(v[2 + 8], 0.f, 1.f); tw(v[2 + 12], -R2, R2);
    tw(v[3 + 4], S1, C1);  tw(v[3 + 8], -R2, R2); tw(v[3 + 12], -C1, -S1);
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) dft4<INV>(v[4 * k2], v[4 * k2 + 1], v[4 * k2 + 2], v[4 * k2 + 3]);
}
// ---- variant 1: vector-native
__device__ __forceinline__ v2f mul_mi(v2f d) { return __builtin_shufflevector(d, -d, 1, 2); }   // (d.y, -d.x) = -i d
__device__ __forceinline__ v2f mul_pi(v2f d) { return __builtin_shufflevector(-d, d, 1, 2); }   // (-d.y, d.x) = +i d
template <bool INV> __device__ __forceinline__ void dft4v(v2f &a0, v2f &a1, v2f &a2, v2f &a3)
{
    const v2f s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, d13 = a1 - a3;
    const v2f id = INV ? mul_pi(d13) : mul_mi(d13);
    a0 = s02 + s13; a2 = s02 - s13; a1 = d02 + id; a3 = d02 - id;
}
// x * (c - i s) forward, (c + i s) inverse:  (x.x c + x.y s, x.y c - x.x s) = x * c + (x.y, -x.x) * s
template <bool INV> __device__ __forceinline__ v2f twv(v2f x, float c, float s) { const float s_ = INV ? -s : s; return x * c + mul_mi(x) * s_; }
template <bool INV> __device__ __forceinline__ void dft16v(v2f (&v)[16])
{
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
#pragma unroll
    for (int t1 = 0; t1 < 4; ++t1) dft4v<INV>(v[t1], v[t1 + 4], v[t1 + 8], v[t1 + 12]);
    v[5] = twv<INV>(v[5], C1, S1); v[9] = twv<INV>(v[9], R2, R2); v[13] = twv<INV>(v[13], S1, C1);
    v[6] = twv<INV>(v[6], R2, R2); v[10] = INV ? mul_pi(v[10]) : mul_mi(v[10]); v[14] = twv<INV>(v[14], -R2, R2);
    v[7] = twv<INV>(v[7], S1, C1); v[11] = twv<INV>(v[11], -R2, R2); v[15] = twv<INV>(v[15], -C1, -S1);
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) dft4v<INV>(v[4 * k2], v[4 * k2 + 1], v[4 * k2 + 2], v[4 * k2 + 3]);
}
// ---- variant 2: vector-native + the two mixed-sign rotations as single packed instructions (non-volatile asm)
// a + (b.y, -b.x)  and  a + (-b.y, b.x)
__device__ __forceinline__ v2f add_mi(v2f a, v2f b) { v2f r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ v2f add_pi(v2f a, v2f b) { v2f r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
// t + (x.y, -x.x) * s   (s splat):  fma with src0 lanes swapped and neg_hi on src0
__device__ __forceinline__ v2f fma_mi(v2f x, v2f s, v2f t) { v2f r; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(x), "v"(s), "v"(t)); return r; }
template <bool INV> __device__ __forceinline__ void dft4a(v2f &a0, v2f &a1, v2f &a2, v2f &a3)
{
    const v2f s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, d13 = a1 - a3;
    a0 = s02 + s13; a2 = s02 - s13;
    a1 = INV ? add_pi(d02, d13) : add_mi(d02, d13);
    a3 = INV ? add_mi(d02, d13) : add_pi(d02, d13);
}
template <bool INV> __device__ __forceinline__ v2f twa(v2f x, float c, float s) { const float s_ = INV ? -s : s; return fma_mi(x, (v2f){s_, s_}, x * c); }
template <bool INV> __device__ __forceinline__ void dft16a(v2f (&v)[16])
{
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
#pragma unroll
    for (int t1 = 0; t1 < 4; ++t1) dft4a<INV>(v[t1], v[t1 + 4], v[t1 + 8], v[t1 + 12]);
    v[5] = twa<INV>(v[5], C1, S1); v[9] = twa<INV>(v[9], R2, R2); v[13] = twa<INV>(v[13], S1, C1);
    v[6] = twa<INV>(v[6], R2, R2); v[10] = twa<INV>(v[10], 0.f, 1.f); v[14] = twa<INV>(v[14], -R2, R2);
    v[7] = twa<INV>(v[7], S1, C1); v[11] = twa<INV>(v[11], -R2, R2); v[15] = twa<INV>(v[15], -C1, -S1);
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) dft4a<INV>(v[4 * k2], v[4 * k2 + 1], v[4 * k2 + 2], v[4 * k2 + 3]);
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int ITER = 2000;
__global__ void __launch_bounds__(256, 4) b0(cpx *p) { cpx v[16]; for (int i = 0; i < 16; ++i) v[i] = p[threadIdx.x + 256 * i]; for (int it = 0; it < ITER; ++it) { dft16<false>(v); dft16<true>(v); for (int i = 0; i < 16; ++i) { v[i].x *= 0.0625f; v[i].y *= 0.0625f; } } for (int i = 0; i < 16; ++i) p[threadIdx.x + 256 * i] = v[i]; }
__global__ void __launch_bounds__(256, 4) b1(v2f *p) { v2f v[16]; for (int i = 0; i < 16; ++i) v[i] = p[threadIdx.x + 256 * i]; for (int it = 0; it < ITER; ++it) { dft16v<false>(v); dft16v<true>(v); for (int i = 0; i < 16; ++i) v[i] *= 0.0625f; } for (int i = 0; i < 16; ++i) p[threadIdx.x + 256 * i] = v[i]; }
__global__ void __launch_bounds__(256, 4) b2(v2f *p) { v2f v[16]; for (int i = 0; i < 16; ++i) v[i] = p[threadIdx.x + 256 * i]; for (int it = 0; it < ITER; ++it) { dft16a<false>(v); dft16a<true>(v); for (int i = 0; i < 16; ++i) v[i] *= 0.0625f; } for (int i = 0; i < 16; ++i) p[threadIdx.x + 256 * i] = v[i]; }

// complex multiplies: struct code (compiler) vs the two-instruction packed form
__device__ __forceinline__ cpx cmul_s(cpx a, cpx b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ v2f cmul_a(v2f a, v2f b)
{
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}
__global__ void __launch_bounds__(256, 4) c0(cpx *p) { cpx v[16], w[16]; for (int i = 0; i < 16; ++i) { v[i] = p[threadIdx.x + 256 * i]; w[i] = p[threadIdx.x + 256 * ((i + 5) & 15)]; } for (int it = 0; it < ITER; ++it) for (int r = 0; r < 4; ++r) for (int i = 0; i < 16; ++i) v[i] = cmul_s(v[i], w[(i + r) & 15]); for (int i = 0; i < 16; ++i) p[threadIdx.x + 256 * i] = v[i]; }
__global__ void __launch_bounds__(256, 4) c1(v2f *p) { v2f v[16], w[16]; for (int i = 0; i < 16; ++i) { v[i] = p[threadIdx.x + 256 * i]; w[i] = p[threadIdx.x + 256 * ((i + 5) & 15)]; } for (int it = 0; it < ITER; ++it) for (int r = 0; r < 4; ++r) for (int i = 0; i < 16; ++i) v[i] = cmul_a(v[i], w[(i + r) & 15]); for (int i = 0; i < 16; ++i) p[threadIdx.x + 256 * i] = v[i]; }
int main()
{
    void *p; CK(hipMalloc(&p, 256 * 16 * 8)); CK(hipMemset(p, 0, 256 * 16 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * 4;      // 4 workgroups of 4 waves per CU = 4 waves per SIMD
    auto run = [&](auto k, const char *name, auto ptr) {
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, ptr);
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) { CK(hipEventRecord(e0)); hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, ptr); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
        // per SIMD: 4 waves x ITER x 2 butterflies
        printf("%-8s %8.3f ms  -> %.1f clocks per butterfly-wave at 2.4 GHz (4 waves per SIMD share it)\n", name, best, best * 1e-3 * 2.4e9 / (4.0 * ITER * 2));
    };
    run(b0, "struct", (cpx *)p); run(b1, "vector", (v2f *)p); run(b2, "asm", (v2f *)p);
    printf("64 complex multiplies per iteration (same normalisation: / (4 ITER 2)):\n");
    run(c0, "cmul struct", (cpx *)p); run(c1, "cmul asm", (v2f *)p);
    return 0;
}
