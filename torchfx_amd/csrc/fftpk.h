// fftpk.h -- float32 FFT butterflies in packed arithmetic for gfx950, shared by the one-launch overlap-save kernel
// (olslds.hip) and the three-pass pipeline (olsnative.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace tfx {
namespace pk {

#define PK_DFT16_AT(k) (4 * ((k) & 3) + ((k) >> 2))

// A complex number is one aligned VGPR pair, and gfx950's v_pk_{add,mul,fma}_f32 take per-source half selectors
// (op_sel / op_sel_hi) and per-half negation: a +- i b is ONE instruction, a complex product two -- the compiler never
// emits those forms (it builds swapped pairs with v_mov / v_xor first: PMC round 4, 1206 vector instructions per wave
// and pair, the kernel 69 % VALU-bound), so the butterflies are written with single-instruction asm statements the
// scheduler is free to place.  A 16-point DFT is 80 instructions instead of ~160.
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v2f pk_add_ib(v2f a, v2f b)            // a + i b
{
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ v2f pk_sub_ib(v2f a, v2f b)            // a - i b
{
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// a * w (CONJ = false) or a * conj(w)
template <bool CONJ> __device__ __forceinline__ v2f pk_cmul(v2f a, v2f w)
{
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));          // (a.x w.x, a.y w.x)
    if (CONJ) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    else      asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// x * (c - i s) forward, x * (c + i s) inverse, with c = (-1)^NEGC W[SELC], s = (-1)^NEGS W[SELS] picked from one constant pair
template <int SELC, int NEGC, int SELS, int NEGS, bool INV>
__device__ __forceinline__ v2f pk_twc(v2f x, v2f W)
{
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,%3] op_sel_hi:[1,%3] neg_lo:[0,%4] neg_hi:[0,%4]" : "=v"(t) : "v"(x), "v"(W), "n"(SELC), "n"(NEGC));
    if (!INV) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,%4,0] op_sel_hi:[0,%4,1] neg_lo:[0,%5,0] neg_hi:[1,%5,0]"
                  : "=v"(r) : "v"(x), "v"(W), "v"(t), "n"(SELS), "n"(NEGS));
    else      asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,%4,0] op_sel_hi:[0,%4,1] neg_lo:[1,%5,0] neg_hi:[0,%5,0]"
                  : "=v"(r) : "v"(x), "v"(W), "v"(t), "n"(SELS), "n"(NEGS));
    return r;
}
// 4-point DFT; ROT2: a2 stands for (-i) a2 forward / (+i) a2 inverse (a twiddle of the 16-point DFT folded in)
template <bool INV, bool ROT2>
__device__ __forceinline__ void pk_dft4(v2f &a0, v2f &a1, v2f &a2, v2f &a3)
{
    v2f s02, d02;
    if (!ROT2) { s02 = a0 + a2; d02 = a0 - a2; }
    else if (!INV) { s02 = pk_sub_ib(a0, a2); d02 = pk_add_ib(a0, a2); }
    else { s02 = pk_add_ib(a0, a2); d02 = pk_sub_ib(a0, a2); }
    const v2f s13 = a1 + a3, d13 = a1 - a3;
    a0 = s02 + s13;
    a2 = s02 - s13;
    a1 = INV ? pk_add_ib(d02, d13) : pk_sub_ib(d02, d13);
    a3 = INV ? pk_sub_ib(d02, d13) : pk_add_ib(d02, d13);
}
// 16-point DFT, same index conventions as dft16<>: Wc = (cos pi/8, sin pi/8), Wr = (sqrt 1/2, sqrt 1/2)
template <bool INV>
__device__ __forceinline__ void pk_dft16(v2f (&v)[16], v2f Wc, v2f Wr)
{
#pragma unroll
    for (int t1 = 0; t1 < 4; ++t1) pk_dft4<INV, false>(v[t1], v[t1 + 4], v[t1 + 8], v[t1 + 12]);
    v[1 + 4] = pk_twc<0, 0, 1, 0, INV>(v[1 + 4], Wc);      // (C1, S1)
    v[1 + 8] = pk_twc<0, 0, 0, 0, INV>(v[1 + 8], Wr);      // (R2, R2)
    v[1 + 12] = pk_twc<1, 0, 0, 0, INV>(v[1 + 12], Wc);    // (S1, C1)
    v[2 + 4] = pk_twc<0, 0, 0, 0, INV>(v[2 + 4], Wr);      // (R2, R2)
    /* v[2 + 8]: (0, 1) = -i / +i, folded into the second pass (ROT2) */
    v[2 + 12] = pk_twc<0, 1, 0, 0, INV>(v[2 + 12], Wr);    // (-R2, R2)
    v[3 + 4] = pk_twc<1, 0, 0, 0, INV>(v[3 + 4], Wc);      // (S1, C1)
    v[3 + 8] = pk_twc<0, 1, 0, 0, INV>(v[3 + 8], Wr);      // (-R2, R2)
    v[3 + 12] = pk_twc<0, 1, 1, 1, INV>(v[3 + 12], Wc);    // (-C1, -S1)
    pk_dft4<INV, false>(v[0], v[1], v[2], v[3]);
    pk_dft4<INV, false>(v[4], v[5], v[6], v[7]);
    pk_dft4<INV, true>(v[8], v[9], v[10], v[11]);
    pk_dft4<INV, false>(v[12], v[13], v[14], v[15]);
}

// N ds_read_b64 at base + t * STRIDE_B (t = T0 ... T0 + N - 1), issued from one asm statement WITHOUT waiting; the caller
// passes the results through lds_wait(), which ties them to the s_waitcnt
template <int STRIDE_B, int T0>
__device__ __forceinline__ void lds_issue8_b64(v2f (&v)[8], unsigned a)
{
    asm volatile(
        "ds_read_b64 %0, %8 offset:%9\n\tds_read_b64 %1, %8 offset:%10\n\tds_read_b64 %2, %8 offset:%11\n\t"
        "ds_read_b64 %3, %8 offset:%12\n\tds_read_b64 %4, %8 offset:%13\n\tds_read_b64 %5, %8 offset:%14\n\t"
        "ds_read_b64 %6, %8 offset:%15\n\tds_read_b64 %7, %8 offset:%16"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
        : "v"(a), "n"((T0 + 0) * STRIDE_B), "n"((T0 + 1) * STRIDE_B), "n"((T0 + 2) * STRIDE_B), "n"((T0 + 3) * STRIDE_B),
          "n"((T0 + 4) * STRIDE_B), "n"((T0 + 5) * STRIDE_B), "n"((T0 + 6) * STRIDE_B), "n"((T0 + 7) * STRIDE_B)
        : "memory");
}
__device__ __forceinline__ void lds_wait8(v2f (&a)[8])
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void *p)
{
    typedef const char __attribute__((address_space(3))) *lds_ptr;
    return (unsigned)(uintptr_t)(lds_ptr)(const char *)p;
}

template <bool INV>
__device__ __forceinline__ void fft4096_pk(v2f (&v)[16], v2f *l2, const v2f *twB, const v2f *twA, int j, v2f Wc, v2f Wr)
{
    const int kb = j & 15, jh = j >> 4;
    pk_dft16<INV>(v, Wc, Wr);
#pragma unroll
    for (int k = 0; k < 16; ++k) l2[j + jh + 272 * k] = v[PK_DFT16_AT(k)];
    __syncthreads();
    {   // stage 2: data (n0 + 256 k0) + 16 n1 and twiddles W256^(n1 k0) = twB[16 n1 + k0]
        v2f a[8], b[8], wa[8], wb[8];
        const unsigned da = lds_addr(l2 + kb + 272 * jh), ta = lds_addr(twB + jh);
        lds_issue8_b64<17 * 8, 0>(a, da);
        lds_issue8_b64<17 * 8, 8>(b, da);
        lds_issue8_b64<16 * 8, 0>(wa, ta);
        lds_issue8_b64<16 * 8, 8>(wb, ta);
        lds_wait8(a); lds_wait8(b); lds_wait8(wa); lds_wait8(wb);
        v[0] = a[0];
#pragma unroll
        for (int t = 1; t < 8; ++t) v[t] = pk_cmul<INV>(a[t], wa[t]);
#pragma unroll
        for (int t = 0; t < 8; ++t) v[8 + t] = pk_cmul<INV>(b[t], wb[t]);
    }
    __syncthreads();
    pk_dft16<INV>(v, Wc, Wr);
#pragma unroll
    for (int k = 0; k < 16; ++k) l2[j + jh + 272 * k] = v[PK_DFT16_AT(k)];
    __syncthreads();
    {   // stage 3: data 16 j + n0, twiddles W4096^(n0 j) = twA[16 n0 + (j & 15)] * twB[16 n0 + (j >> 4)]
        v2f a[8], wa[8], wb[8];
        const unsigned da = lds_addr(l2 + 17 * j), t1 = lds_addr(twA + kb), t2 = lds_addr(twB + jh);
        lds_issue8_b64<8, 0>(a, da);
        lds_issue8_b64<16 * 8, 0>(wa, t1);
        lds_issue8_b64<16 * 8, 0>(wb, t2);
        lds_wait8(a); lds_wait8(wa); lds_wait8(wb);
        v[0] = a[0];
#pragma unroll
        for (int t = 1; t < 8; ++t) v[t] = pk_cmul<INV>(a[t], pk_cmul<false>(wa[t], wb[t]));
        lds_issue8_b64<8, 8>(a, da);
        lds_issue8_b64<16 * 8, 8>(wa, t1);
        lds_issue8_b64<16 * 8, 8>(wb, t2);
        lds_wait8(a); lds_wait8(wa); lds_wait8(wb);
#pragma unroll
        for (int t = 0; t < 8; ++t) v[8 + t] = pk_cmul<INV>(a[t], pk_cmul<false>(wa[t], wb[t]));
    }
    __syncthreads();
    pk_dft16<INV>(v, Wc, Wr);
    v2f o[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) o[k] = v[PK_DFT16_AT(k)];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = o[k];
}


}  // namespace pk
}  // namespace tfx
