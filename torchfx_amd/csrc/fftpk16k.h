// fftpk16k.h -- 16 384-point complex transform of ONE 1024-thread workgroup in packed float32 arithmetic (fftpk.h), for the
// one-launch overlap-save kernel's second block size (olslds.hip: 2048 < K <= 8192 taps).
//
// Thread j holds sixteen values.  N = 4 * 16 * 16 * 16, time index n = n0 + 4 n1 + 64 n2 + 1024 n3 (n0 < 4), frequency
// index k = k0 + 16 k1 + 256 k2 + 4096 k3 (k3 < 4).  Forward (decimation in time, most significant digit first):
//     S1  thread (n0, n1, n2) = j            : DFT16 over n3                              -> k0
//     X1  -> thread (n0, n1, k0), values over n2   * W256^(n2 k0)
//     S2                                       DFT16 over n2                              -> k1
//     X2  -> thread (n0, k0, k1), values over n1   * W4096^(n1 (k0 + 16 k1))
//     S3                                       DFT16 over n1                              -> k2
//     X3  -> thread (k0, k1, q = k2 / 4), values over (n0, k2 % 4)   * W16384^(n0 (k0 + 16 k1 + 256 k2))
//     S4                                       four DFT4 over n0                          -> k3
// and leaves X[k] in the "spectral ownership": thread r = k0 + 16 k1 + 256 q, register 4 (k2 % 4) + k3.  The inverse is the
// mirror image (S4' first, the exchanges in reverse order, conjugated twiddles) and ends in the natural ownership
// x[j + 1024 t] in register t -- what coalesced stores want.  Three exchanges per direction through one LDS buffer of
// X16K_SLOTS complex slots; every exchange writes AND reads with the lane index on the fastest-moving address digit (strides
// 68 / 17 / 260 / 1040 between the other digits), so all of them are free of bank conflicts for 8-byte elements.
#pragma once
#include "fftpk.h"

namespace tfx {
namespace pk {

constexpr int X16K_SLOTS = 17408;                 // complex slots of the exchange buffer (16 * 1088)
constexpr int X16K_TABLES = 256 + 256 + 64 + 64 + 64;

struct Tab16k {                                   // twiddle tables in LDS (filled by fill_tab16k)
    const v2f *twA;                               // [16 a + b] = W4096^(a b)
    const v2f *twB;                               // [16 a + b] = W256^(a b)
    const v2f *tA;                                // [16 n0 + k0] = W16384^(n0 k0)      (n0 < 4)
    const v2f *tB;                                // [16 n0 + k1] = W1024^(n0 k1)
    const v2f *tC;                                // [16 n0 + k2] = W64^(n0 k2)
};

// v[t] = src[t * DS] * tw(t), t < 16, in two batches of eight (registers): tw(t) = wa[t * 16] (ONE = true) or wa[t * 16] * wb[t * 16];
// t = 0 has a unit twiddle.  CONJ multiplies by the conjugate.
template <bool CONJ, bool TWO, int DS>
__device__ __forceinline__ void gather_tw16(v2f (&v)[16], const v2f *src, const v2f *wa, const v2f *wb)
{
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        v2f d[8], a[8], b[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) d[t] = src[(8 * h + t) * DS];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (8 * h + t == 0) continue;
            a[t] = wa[(8 * h + t) * 16];
            if (TWO) b[t] = wb[(8 * h + t) * 16];
        }
        __builtin_amdgcn_sched_barrier(0);           // all reads of the batch are issued before the first product
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (8 * h + t == 0) { v[0] = d[0]; continue; }
            v[8 * h + t] = pk_cmul<CONJ>(d[t], TWO ? pk_cmul<false>(a[t], b[t]) : a[t]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// W16384^(d K2), K2 = k01 + 256 k2, for d = 1 ... 3 and the four k2 = 4 q + e of a thread: w[4 e + d] (d = 0: one)
__device__ __forceinline__ void tw16k_last(v2f (&w)[16], const Tab16k &tb, int k0, int k1, int q)
{
#pragma unroll
    for (int d = 1; d < 4; ++d) {
        const v2f p = pk_cmul<false>(tb.tA[16 * d + k0], tb.tB[16 * d + k1]);
#pragma unroll
        for (int e = 0; e < 4; ++e) w[4 * e + d] = pk_cmul<false>(p, tb.tC[16 * d + 4 * q + e]);
    }
}

// in: v[t] = z[j + 1024 t];  out: v[4 e + k3] = Z[(j & 255) + 256 (4 (j >> 8) + e) + 4096 k3]
__device__ __forceinline__ void fft16384_fwd(v2f (&v)[16], v2f *L, const Tab16k &tb, int j, v2f Wc, v2f Wr)
{
    const int lo6 = j & 63, hi4 = j >> 6;
    pk_dft16<false>(v, Wc, Wr);                                        // S1: over n3 -> k0
#pragma unroll
    for (int k = 0; k < 16; ++k) L[j + 1024 * k] = v[PK_DFT16_AT(k)];  // X1: slot (n0 + 4 n1) + 64 n2 + 1024 k0
    __syncthreads();
    gather_tw16<false, false, 64>(v, L + lo6 + 1024 * hi4, tb.twB + hi4, nullptr);     // thread (n0, n1, k0 = j >> 6): values over n2, * W256^(n2 k0)
    __syncthreads();
    pk_dft16<false>(v, Wc, Wr);                                        // S2: over n2 -> k1
    {
        const int n0 = j & 3, n1 = (j >> 2) & 15;                      // writer (n0, n1, k0): slot (n0 + 4 k0) + 68 n1 + 1088 k1
        v2f *dst = L + n0 + 4 * hi4 + 68 * n1;
#pragma unroll
        for (int k = 0; k < 16; ++k) dst[1088 * k] = v[PK_DFT16_AT(k)];
    }
    __syncthreads();
    gather_tw16<false, true, 68>(v, L + lo6 + 1088 * hi4, tb.twA + ((j >> 2) & 15), tb.twB + hi4);   // reader (n0, k0, k1 = j >> 6): over n1, * W4096^(n1 (k0 + 16 k1))
    __syncthreads();
    pk_dft16<false>(v, Wc, Wr);                                        // S3: over n1 -> k2
    {
        const int n0 = j & 3, k0 = (j >> 2) & 15;                      // writer (n0, k0, k1): slot (k0 + 16 k1) + 260 n0 + 1040 k2
        v2f *dst = L + k0 + 16 * hi4 + 260 * n0;
#pragma unroll
        for (int k = 0; k < 16; ++k) dst[1040 * k] = v[PK_DFT16_AT(k)];
    }
    __syncthreads();
    {
        v2f d[16], w[16];
        const int q = j >> 8;                                          // reader (k0, k1, q): values (n0, e), k2 = 4 q + e
        const v2f *src = L + (j & 255) + 1040 * 4 * q;
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int n0 = 0; n0 < 4; ++n0) d[4 * e + n0] = src[260 * n0 + 1040 * e];
        tw16k_last(w, tb, j & 15, (j >> 4) & 15, q);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[4 * e] = d[4 * e];
#pragma unroll
            for (int n0 = 1; n0 < 4; ++n0) v[4 * e + n0] = pk_cmul<false>(d[4 * e + n0], w[4 * e + n0]);
        }
    }
    __syncthreads();                                                   // the buffer is free again
#pragma unroll
    for (int e = 0; e < 4; ++e) pk_dft4<false, false>(v[4 * e], v[4 * e + 1], v[4 * e + 2], v[4 * e + 3]);   // S4: over n0 -> k3
}

// in: v[4 e + k3] = Z[(j & 255) + 256 (4 (j >> 8) + e) + 4096 k3];  out: v[t] = N * z[j + 1024 t]  (unscaled inverse)
__device__ __forceinline__ void fft16384_inv(v2f (&v)[16], v2f *L, const Tab16k &tb, int j, v2f Wc, v2f Wr)
{
    const int lo6 = j & 63, hi4 = j >> 6;
#pragma unroll
    for (int e = 0; e < 4; ++e) pk_dft4<true, false>(v[4 * e], v[4 * e + 1], v[4 * e + 2], v[4 * e + 3]);    // over k3 -> m0
    {
        v2f w[16];
        const int q = j >> 8;
        tw16k_last(w, tb, j & 15, (j >> 4) & 15, q);                   // conj W16384^(m0 K2)
        v2f *dst = L + (j & 255) + 1040 * 4 * q;                       // writer (k0, k1, q): slot (k0 + 16 k1) + 260 m0 + 1040 k2
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dst[1040 * e] = v[4 * e];
#pragma unroll
            for (int m0 = 1; m0 < 4; ++m0) dst[260 * m0 + 1040 * e] = pk_cmul<true>(v[4 * e + m0], w[4 * e + m0]);
        }
    }
    __syncthreads();
    {
        const int m0 = j >> 8;                                         // reader (k0, k1, m0): values over k2
        const v2f *src = L + (j & 255) + 260 * m0;
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = src[1040 * t];
    }
    __syncthreads();
    pk_dft16<true>(v, Wc, Wr);                                         // over k2 -> m1
    {
        const int k0 = j & 15, k1 = (j >> 4) & 15, m0 = j >> 8;        // writer (k0, k1, m0): slot k0 + 17 (m0 + 4 m1) + 1088 k1
        v2f *dst = L + k0 + 17 * m0 + 1088 * k1;
#pragma unroll
        for (int k = 0; k < 16; ++k) dst[68 * k] = v[PK_DFT16_AT(k)];
    }
    __syncthreads();
    gather_tw16<true, false, 1088>(v, L + hi4 + 17 * lo6, tb.twB + ((j >> 2) & 15), nullptr);   // reader (m0, m1, k0 = j >> 6): over k1, * conj W256^(k1 m1)
    __syncthreads();
    pk_dft16<true>(v, Wc, Wr);                                         // over k1 -> m2
    {
        v2f *dst = L + lo6 + 1024 * hi4;                               // writer (m0, m1, k0): slot (m0 + 4 m1) + 64 m2 + 1024 k0
#pragma unroll
        for (int k = 0; k < 16; ++k) dst[64 * k] = v[PK_DFT16_AT(k)];
    }
    __syncthreads();
    gather_tw16<true, true, 1024>(v, L + j, tb.twA + ((j >> 2) & 15), tb.twB + hi4);   // reader (m0, m1, m2 = j >> 6) = j: over k0, * conj W4096^(k0 (m1 + 16 m2))
    __syncthreads();                                                   // the buffer is free again
    pk_dft16<true>(v, Wc, Wr);                                         // over k0 -> m3
    v2f o[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) o[k] = v[PK_DFT16_AT(k)];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = o[k];
}

// tables -> LDS (1024 threads), from a global copy laid out [twA 256 | twB 256 | tA 64 | tB 64 | tC 64]
__device__ __forceinline__ Tab16k fill_tab16k(v2f *lds_tab, const v2f *__restrict__ g, int j)
{
    if (j < X16K_TABLES) lds_tab[j] = g[j];
    Tab16k tb;
    tb.twA = lds_tab;
    tb.twB = lds_tab + 256;
    tb.tA = lds_tab + 512;
    tb.tB = lds_tab + 576;
    tb.tC = lds_tab + 640;
    return tb;
}

}  // namespace pk
}  // namespace tfx
