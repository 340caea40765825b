// ldsfft.h -- the templated (float32 / float64) complex arithmetic, 16-point butterfly and 4096-point workgroup transform shared
// by the one-launch overlap-save kernels (olslds.hip) and the float64 three-pass pipeline (olsnative64.hip).
#pragma once
#include "common.h"

namespace tfx {
namespace ldsfft {

template <typename R> struct alignas(2 * sizeof(R)) cx {
    R x, y;
};
template <typename R> __device__ __forceinline__ cx<R> mk(R a, R b)
{
    cx<R> r;
    r.x = a;
    r.y = b;
    return r;
}
template <typename R> __device__ __forceinline__ cx<R> cmul(cx<R> a, cx<R> b) { return mk<R>(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
template <typename R> __device__ __forceinline__ cx<R> cmulc(cx<R> a, cx<R> b) { return mk<R>(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }  // a conj(b)
template <typename R> __device__ __forceinline__ cx<R> cadd(cx<R> a, cx<R> b) { return mk<R>(a.x + b.x, a.y + b.y); }
template <typename R> __device__ __forceinline__ cx<R> csub(cx<R> a, cx<R> b) { return mk<R>(a.x - b.x, a.y - b.y); }

template <typename R, bool INV>
__device__ __forceinline__ void dft4(cx<R> &a0, cx<R> &a1, cx<R> &a2, cx<R> &a3)
{
    const cx<R> s02 = cadd(a0, a2), d02 = csub(a0, a2), s13 = cadd(a1, a3), d13 = csub(a1, a3);
    const cx<R> id = INV ? mk<R>(-d13.y, d13.x) : mk<R>(d13.y, -d13.x);      // (+i or -i) d13
    a0 = cadd(s02, s13);
    a2 = csub(s02, s13);
    a1 = cadd(d02, id);
    a3 = csub(d02, id);
}

// 16-point DFT in registers, t = t1 + 4 t2, k = 4 k1 + k2; X[k] ends up at v[4 (k % 4) + k / 4]
template <typename R, bool INV>
__device__ __forceinline__ void dft16(cx<R> (&v)[16])
{
    constexpr R C1 = (R)0.92387953251128675613L, S1 = (R)0.38268343236508977173L, R2 = (R)0.70710678118654752440L;
#pragma unroll
    for (int t1 = 0; t1 < 4; ++t1) dft4<R, INV>(v[t1], v[t1 + 4], v[t1 + 8], v[t1 + 12]);
    auto tw = [&](cx<R> &x, R c, R sn) {                   // times (c - i sn) forward, (c + i sn) inverse
        const R s_ = INV ? -sn : sn;
        x = mk<R>(x.x * c + x.y * s_, x.y * c - x.x * s_);
    };
    tw(v[1 + 4], C1, S1);  tw(v[1 + 8], R2, R2);      tw(v[1 + 12], S1, C1);
    tw(v[2 + 4], R2, R2);  tw(v[2 + 8], (R)0, (R)1);  tw(v[2 + 12], -R2, R2);
    tw(v[3 + 4], S1, C1);  tw(v[3 + 8], -R2, R2);     tw(v[3 + 12], -C1, -S1);
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) dft4<R, INV>(v[4 * k2], v[4 * k2 + 1], v[4 * k2 + 2], v[4 * k2 + 3]);
}
#define LDS_DFT16_AT(k) (4 * ((k) & 3) + ((k) >> 2))

// sixteen ds_read_b64 at base + t * STRIDE_B from one asm statement: the load/store optimiser would pair them into
// ds_read2_b64, which the LDS serves at half the bytes per clock (MI355X_MICROARCH.md, LDS table)
template <int STRIDE_B>
__device__ __forceinline__ void lds_read16_b64(cx<float> (&v)[16], const cx<float> *p)
{
    typedef const char __attribute__((address_space(3))) *lds_ptr;
    const unsigned a = (unsigned)(uintptr_t)(lds_ptr)(const char *)p;
    double d[16];
    asm volatile(
        "ds_read_b64 %0, %16 offset:%17\n\tds_read_b64 %1, %16 offset:%18\n\tds_read_b64 %2, %16 offset:%19\n\t"
        "ds_read_b64 %3, %16 offset:%20\n\tds_read_b64 %4, %16 offset:%21\n\tds_read_b64 %5, %16 offset:%22\n\t"
        "ds_read_b64 %6, %16 offset:%23\n\tds_read_b64 %7, %16 offset:%24\n\tds_read_b64 %8, %16 offset:%25\n\t"
        "ds_read_b64 %9, %16 offset:%26\n\tds_read_b64 %10, %16 offset:%27\n\tds_read_b64 %11, %16 offset:%28\n\t"
        "ds_read_b64 %12, %16 offset:%29\n\tds_read_b64 %13, %16 offset:%30\n\tds_read_b64 %14, %16 offset:%31\n\t"
        "ds_read_b64 %15, %16 offset:%32\n\ts_waitcnt lgkmcnt(0)"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]),
          "=&v"(d[8]), "=&v"(d[9]), "=&v"(d[10]), "=&v"(d[11]), "=&v"(d[12]), "=&v"(d[13]), "=&v"(d[14]), "=&v"(d[15])
        : "v"(a), "n"(0 * STRIDE_B), "n"(1 * STRIDE_B), "n"(2 * STRIDE_B), "n"(3 * STRIDE_B), "n"(4 * STRIDE_B),
          "n"(5 * STRIDE_B), "n"(6 * STRIDE_B), "n"(7 * STRIDE_B), "n"(8 * STRIDE_B), "n"(9 * STRIDE_B), "n"(10 * STRIDE_B),
          "n"(11 * STRIDE_B), "n"(12 * STRIDE_B), "n"(13 * STRIDE_B), "n"(14 * STRIDE_B), "n"(15 * STRIDE_B)
        : "memory");
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = __builtin_bit_cast(cx<float>, d[t]);
}
template <int STRIDE> __device__ __forceinline__ void lds_read16(cx<float> (&v)[16], const cx<float> *p) { lds_read16_b64<STRIDE * 8>(v, p); }
template <int STRIDE> __device__ __forceinline__ void lds_read16(cx<double> (&v)[16], const cx<double> *p)
{
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = p[t * STRIDE];
}

// 4096-point transform of one workgroup (256 threads), thread j owns elements j + 256 t in natural order on entry and
// on exit.  Radix (16, 16, 16) Stockham, "write contiguous / read strided" exchanges (olsnative.hip, layout 1/2): with
// n = n0 + 16 n1 + 256 n2, k = k0 + 16 k1 + 256 k2
//   stage 1  thread j = n0 + 16 n1 : DFT over n2 -> A[k0] stored at  j + 256 k0
//   stage 2  thread j = n0 + 16 k0 : reads (n0 + 256 k0) + 16 n1, * W256^(n1 k0), DFT over n1 -> B[k1] at j + 256 k1
//   stage 3  thread j = k0 + 16 k1 : reads 16 j + n0, * W4096^(n0 j), DFT over n0 -> X[j + 256 k2]
// physical position of logical p is p + p / 16: addresses stay base + immediate and every exchange is conflict-free
// for 8-byte elements.  The last barrier leaves the buffer free for the next transform.
template <typename R, bool INV>
__device__ __forceinline__ void fft4096(cx<R> (&v)[16], cx<R> *lds, const cx<R> *twB, const cx<R> *twA, int j)
{
    dft16<R, INV>(v);
    const int kb = j & 15, jh = j >> 4;
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[j + jh + 272 * k] = v[LDS_DFT16_AT(k)];
    __syncthreads();
    lds_read16<17>(v, lds + kb + 272 * jh);
#pragma unroll
    for (int t = 1; t < 16; ++t) {
        const cx<R> w = twB[16 * t + jh];                      // [t][k0]: broadcast within a 16-lane group
        v[t] = INV ? cmulc(v[t], w) : cmul(v[t], w);
    }
    __syncthreads();
    dft16<R, INV>(v);
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[j + jh + 272 * k] = v[LDS_DFT16_AT(k)];
    __syncthreads();
    lds_read16<1>(v, lds + 17 * j);
#pragma unroll
    for (int t = 1; t < 16; ++t) {
        const cx<R> w = cmul(twA[16 * t + kb], twB[16 * t + jh]);   // W4096^(t j) = W4096^(t (j & 15)) W256^(t (j >> 4))
        v[t] = INV ? cmulc(v[t], w) : cmul(v[t], w);
    }
    __syncthreads();
    dft16<R, INV>(v);
    cx<R> o[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) o[k] = v[LDS_DFT16_AT(k)];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = o[k];
}

}  // namespace ldsfft
}  // namespace tfx
