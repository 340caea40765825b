// olslds.hip -- overlap-save convolution in ONE launch: the whole transform lives in LDS.
//
// For kernels that fit on chip (K <= 8192 taps in float32, 4096 in float64: the reference's default FIR mode --
// `FIR.forward` with conv_mode "fft", src/torchfx/filter/fir.py:552-579 -> fft_conv1d, _fftconv.py:70-141 -- is
// exercised at K = 5 ... 1024 by its own tests and benchmarks) the three-pass pipeline of olsnative.hip moves 24-31 B per
// output sample through a workspace although a block of N = 4096 points is 32 KB.  Here one workgroup owns one
// PAIR of real frames (N = 4096 below; 8192 and 16 384 points are a radix-2 / radix-4 step in registers around two /
// four of the same 4096-point transforms, further down):
//
//     z[n] = frame_a[n] + i frame_b[n]                 gathered straight from the signal (zero fill / history
//                                                      outside the row: causal padding, ragged tail)
//     Z = FFT_4096(z) ; Z *= Hs ; o = IFFT_4096(Z)     radix 16 x 16 x 16 Stockham, registers + LDS,
//                                                      Hs = conj(FFT(taps, zero padded)) / N, cached per filter
//     y_a[f_a S + n] = Re o[n], y_b[f_b S + n] = Im o[n],  n < S = N - K + 1 (the valid part of the block)
//
// (the taps are real, so the two frames separate as real and imaginary part -- no untangling pass).  No workspace:
// HBM traffic is 4 N / S + 4 bytes per output sample (9.3 at K = 1024) instead of 20 N / S + 4, one launch
// instead of three per slab, and rows of any length qualify (the reference's own test shape [2, 44100] no longer
// falls to rocFFT).  float32 and float64 signals (the reference keeps float64 signals float64 through FIR.forward,
// fir.py:526-579); the spectrum and all twiddles are computed on the host in float64.
//
// Semantics = fft_conv1d (src/torchfx/filter/_fftconv.py:70-141): causal correlation with the stored flipped kernel,
// output length T + l + r - K + 1; the block size is a power of two chosen for the device, not int(5 K).
#include "common.h"
#include "epilogue.h"
#include "fftpk.h"
#include "fftpk16k.h"
#include "ldsfft.h"
#include "../../include/torchfx_hip.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace tfx {

void host_fft_f64(std::vector<double> &re, std::vector<double> &im);      // olsnative.hip

namespace ldsfft {

using pk::v2f;
using pk::v4f;

template <typename R> struct Geom {
    int64_t Tn, Tout;     // row lengths in / out
    int64_t F, S;         // frames per row, hop
    int64_t pad_left;     // left padding of the framed signal (the caller's plus `lead`)
    int64_t nframes;      // C * F
    const R *hist;        // streaming: [C, H] samples preceding each row instead of zero padding, or null
    int64_t H;
    R ep_gain;
    int ep_scale, ep_clamp, ep_stat;
    double *ep_partial;   // [nframes]: one partial per frame
    int nt;               // TFX_OLS_LDS_NT (default 2): 2 = the output is stored with the nontemporal hint (written once: 2-3 % on every
                          // block size; the same hint on the signal loads, whose overlap the XCD's L2 serves, changes nothing)
};

constexpr int LDS_N = 4096;
template <typename R> constexpr size_t lds_bytes() { return (size_t)(LDS_N + LDS_N / 16 + 512) * sizeof(cx<R>); }

// Where a pair of frames lives: rows and offsets of frame a (real part) and frame b (imaginary part)
// (the 64-bit divisions run on the vector unit: their wave-uniform results go back to scalar registers, or row bases and
// frame offsets sit in VGPR pairs for the whole transform -- three spilled pairs in the 8192-point kernel, 138 MB of scratch
// traffic per call beside 1.47 GB of signal)
__device__ __forceinline__ int64_t uniform64(int64_t v)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uint64_t)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
template <typename R> struct PairAt {
    int64_t fa, ca, cb, ra, rb;
    bool has_b;
    __device__ __forceinline__ PairAt(int64_t pair, const Geom<R> &g)
    {
        fa = 2 * pair;
        has_b = fa + 1 < g.nframes;
        ca = uniform64(fa / g.F); ra = fa - ca * g.F;
        cb = has_b ? uniform64((fa + 1) / g.F) : ca; rb = has_b ? (fa + 1) - cb * g.F : ra;
    }
};

// thread j's sixteen elements j + 256 t of z = frame_a + i frame_b, gathered from the signal (zero fill / history outside the row)
template <typename R, int NP = 4096, int VPT = 16>
__device__ __forceinline__ void fetch_pair(cx<R> (&v)[VPT], const R *__restrict__ x, const Geom<R> &g, const PairAt<R> &p, int j)
{
    constexpr int TPB = NP / VPT;                 // threads per block: thread j holds elements j + TPB t
    const int64_t ia0 = p.ra * g.S - g.pad_left, ib0 = p.rb * g.S - g.pad_left;
    const R *xa = x + p.ca * g.Tn, *xb = x + p.cb * g.Tn;
    if (ia0 >= 0 && ia0 + NP <= g.Tn && p.has_b && ib0 >= 0 && ib0 + NP <= g.Tn) {     // interior pair: no checks
        // wave-uniform bases + a 32-bit lane offset: the loads take the scalar-base form (one VGPR of address for all of
        // them instead of a 64-bit pointer per 4 KB of reach -- those cost the 8192-point kernel three spilled pairs)
        const R *pa = xa + ia0, *pb = xb + ib0;
#pragma unroll
        for (int t = 0; t < VPT; ++t) v[t] = mk<R>(pa[(unsigned)(j + TPB * t)], pb[(unsigned)(j + TPB * t)]);
    } else {
#pragma unroll
        for (int t = 0; t < VPT; ++t) {
            const int64_t ia = ia0 + j + TPB * t, ib = ib0 + j + TPB * t;
            R re = (ia >= 0 && ia < g.Tn) ? xa[ia] : (R)0;
            R im = (p.has_b && ib >= 0 && ib < g.Tn) ? xb[ib] : (R)0;
            if (g.hist) {
                if (ia < 0 && ia >= -g.H) re = g.hist[p.ca * g.H + g.H + ia];
                if (p.has_b && ib < 0 && ib >= -g.H) im = g.hist[p.cb * g.H + g.H + ib];
            }
            v[t] = mk<R>(re, im);
        }
    }
}

// Two frames share one complex transform, so a non-finite sample of one comes out in both.  Inside a signal row that only widens
// the non-finite stretch by a block (the reference's blocks are other sizes anyway, _fftconv.py:119-122) -- but a pair that
// STRADDLES two rows would carry row c's NaN into the tail of row c - 1, and rows are independent signals.  Such pairs (at most
// one per row, only when a row has an odd number of frames) check frame b on the way in: non-finite samples enter the transform
// as zeros, and the workgroup -- it owns the pair from load to store -- writes NaN as frame b's output (poison_partner), which is
// what it would have been.  Returns whether frame b has to be poisoned (workgroup-uniform).
template <typename R, int VPT>
__device__ __forceinline__ bool sanitize_partner(cx<R> (&v)[VPT], const PairAt<R> &p)
{
    if (!(p.has_b && p.ca != p.cb)) return false;              // workgroup-uniform: the barrier below is reached by all or none
    int bad = 0;
#pragma unroll
    for (int t = 0; t < VPT; ++t)
        if (!(__builtin_fabs((double)v[t].y) <= 1.7976931348623157e308)) { v[t].y = (R)0; bad = 1; }
    return __syncthreads_or(bad) != 0;
}
template <typename R, int VPT>
__device__ __forceinline__ void poison_partner(cx<R> (&v)[VPT], bool nan_b)
{
    if (!nan_b) return;
#pragma unroll
    for (int t = 0; t < VPT; ++t) v[t].y = (R)__builtin_nan("");
}

// the valid part of the block, n < S: real part -> frame a's hop, imaginary part -> frame b's
template <typename R, int NP = 4096, int VPT = 16>
__device__ __forceinline__ void store_pair(const cx<R> (&v)[VPT], R *__restrict__ y, const Geom<R> &g, const PairAt<R> &p, int j, char *smem)
{
    constexpr int TPB = NP / VPT, NW = TPB / 64;  // threads and wavefronts per block
    const int64_t oa0 = p.ra * g.S, ob0 = p.rb * g.S;
    R *ya = y + p.ca * g.Tout + oa0, *yb = y + p.cb * g.Tout + ob0;
    const bool epi = g.ep_scale | g.ep_clamp | (g.ep_stat >= 0);
    if (!epi && p.has_b && oa0 + g.S <= g.Tout && ob0 + g.S <= g.Tout) {     // whole hops inside their rows
        if (g.nt & 2) {                                        // the output is written once: streaming stores
#pragma unroll
            for (int k = 0; k < VPT; ++k) {
                const unsigned n = (unsigned)(j + TPB * k);
                if ((int64_t)n < g.S) { __builtin_nontemporal_store(v[k].x, ya + n); __builtin_nontemporal_store(v[k].y, yb + n); }
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < VPT; ++k) {
            const unsigned n = (unsigned)(j + TPB * k);
            if ((int64_t)n < g.S) { ya[n] = v[k].x; yb[n] = v[k].y; }
        }
        return;
    }
    double acc_a = 0.0, acc_b = 0.0;
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
        const int n = j + TPB * k;
        if (n < g.S) {
            cx<R> o = v[k];
            const bool wa = oa0 + n < g.Tout, wb = p.has_b && ob0 + n < g.Tout;
            if (epi) {                                       // Gain / clamp / statistic on the stored values (epilogue.h)
                if (g.ep_scale) { o.x *= g.ep_gain; o.y *= g.ep_gain; }
                if (g.ep_clamp) { o.x = clamp_unit(o.x); o.y = clamp_unit(o.y); }
                if (g.ep_stat >= 0) {
                    if (wa) acc_a = red_comb_rt(g.ep_stat, acc_a, red_elem_rt(g.ep_stat, (double)o.x));
                    if (wb) acc_b = red_comb_rt(g.ep_stat, acc_b, red_elem_rt(g.ep_stat, (double)o.y));
                }
            }
            if (wa) ya[n] = o.x;
            if (wb) yb[n] = o.y;
        }
    }
    if (g.ep_stat >= 0) {                      // one partial per frame, threads combined in a fixed order
        double *red = (double *)smem;          // the transform buffer is free (fft4096 ends with a barrier)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            acc_a = red_comb_rt(g.ep_stat, acc_a, __shfl_xor(acc_a, off));
            acc_b = red_comb_rt(g.ep_stat, acc_b, __shfl_xor(acc_b, off));
        }
        const int w = j >> 6;
        if ((j & 63) == 0) { red[w] = acc_a; red[NW + w] = acc_b; }
        __syncthreads();
        if (j == 0) {
            double sa = red[0], sb = red[NW];
            for (int u = 1; u < NW; ++u) { sa = red_comb_rt(g.ep_stat, sa, red[u]); sb = red_comb_rt(g.ep_stat, sb, red[NW + u]); }
            g.ep_partial[p.fa] = sa;
            if (p.has_b) g.ep_partial[p.fa + 1] = sb;
        }
    }
}

// forward transform, spectrum multiply, inverse transform of the sixteen elements a thread holds
template <typename R>
__device__ __forceinline__ void transform_pair(cx<R> (&v)[16], cx<R> *lds, const cx<R> *twB, const cx<R> *twA,
                                               const cx<R> *__restrict__ Hs, int j)
{
    fft4096<R, false>(v, lds, twB, twA, j);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = cmul(v[t], Hs[j + 256 * t]);
    __builtin_amdgcn_sched_barrier(0);
    fft4096<R, true>(v, lds, twB, twA, j);
}
#ifndef TFX_LDS_NO_PK
template <>
__device__ __forceinline__ void transform_pair<float>(cx<float> (&v)[16], cx<float> *lds, const cx<float> *twB, const cx<float> *twA,
                                                      const cx<float> *__restrict__ Hs, int j)
{
    const v2f Wc = {0.92387953251128675613f, 0.38268343236508977173f}, Wr = {0.70710678118654752440f, 0.70710678118654752440f};
    v2f u[16], h[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) u[t] = __builtin_bit_cast(v2f, v[t]);
    pk::fft4096_pk<false>(u, (v2f *)lds, (const v2f *)twB, (const v2f *)twA, j, Wc, Wr);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 16; t += 2) {                      // float32 spectra are stored pair-interleaved: one 16-byte load = H[j + 256 t], H[j + 256 (t + 1)]
        const v4f q = ((const v4f *)Hs)[(t >> 1) * 256 + j];
        h[t] = v2f{q.x, q.y};
        h[t + 1] = v2f{q.z, q.w};
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 16; ++t) u[t] = pk::pk_cmul<false>(u[t], h[t]);
    __builtin_amdgcn_sched_barrier(0);
    pk::fft4096_pk<true>(u, (v2f *)lds, (const v2f *)twB, (const v2f *)twA, j, Wc, Wr);
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = __builtin_bit_cast(cx<float>, u[t]);
}
#endif

// Pair -> workgroup: workgroups are dealt to the eight XCDs round robin (observed, MI355X_MICROARCH.md; only speed depends on
// it).  XCD b % 8 owns the pairs [xcd * per_xcd, (xcd + 1) * per_xcd) and its workgroups (local index m = b / 8) take them in
// order: the pairs one XCD works on at a time are neighbours in memory, so the K - 1 samples two consecutive pairs share are
// found in that XCD's L2 (PMC: 8.0 B of HBM traffic per output sample at K = 1024, i.e. 1.00 x algorithmic).
// (Persistent workgroups that fetch pair i + 1 into registers while they transform pair i were built and measured: 3 waves per
// SIMD instead of 4 -- or spills at 4 -- cost more than the overlap gains, 0.41 vs 0.375 ms; profiles/r04_experiments.txt.)
template <typename R>
__global__ void __launch_bounds__(256, sizeof(R) == 4 ? 4 : 2)
ols_lds4096_kernel(const R *__restrict__ x, R *__restrict__ y, const cx<R> *__restrict__ Hs,
                   const cx<R> *__restrict__ tw256g, const cx<R> *__restrict__ t4log, Geom<R> g, int64_t npairs, int64_t per_xcd)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cx<R> *lds = (cx<R> *)smem;                       // [4096 + 256]
    cx<R> *twB = lds + LDS_N + LDS_N / 16;            // [16][16]  W256^(t k)
    cx<R> *twA = twB + 256;                           // [16][16]  W4096^(t a)
    const int j = threadIdx.x;
    twB[j] = tw256g[((j >> 4) * (j & 15)) & 255];
    twA[j] = t4log[j];
    const int64_t pair = (int64_t)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if ((int64_t)(blockIdx.x >> 3) >= per_xcd || pair >= npairs) return;
    __syncthreads();
    cx<R> v[16];
    const PairAt<R> p(pair, g);
    fetch_pair<R>(v, x, g, p, j);
    const bool nan_b = sanitize_partner<R, 16>(v, p);
    transform_pair<R>(v, lds, twB, twA, Hs, j);
    poison_partner<R, 16>(v, nan_b);
    store_pair<R>(v, y, g, p, j, smem);
}

// ---- 8192 points in the same 256-thread workgroup (float32; K <= 4096) ------------------------------------------------------
// One radix-2 step in registers around TWO 4096-point transforms that run one after the other through the same 34 KB exchange
// buffer: thread j holds z[j + 256 t], t < 32;  a = z_lo + z_hi,  b = (z_lo - z_hi) W8192^n  (n = j + 256 t < 4096, decimation in
// frequency), FFT_4096(a) = the even bins, FFT_4096(b) = the odd bins, and the mirror image on the way back.  Same LDS and
// four workgroups per CU as the 4096-point kernel, 13 butterfly levels instead of 12 for twice the block: the valid part of a
// block grows from (4096 - K) / 4096 to (8192 - K) / 8192 -- 0.75 -> 0.875 at 1024 taps, fewer points transformed and fewer
// input samples read per output sample -- and 2048 < K <= 4096 stays in one launch.  W8192^n = W8192^j W32^t: the first factor is
// one table entry per thread, the second is W256^(8 t) = twB[16 t + 8], already in LDS.  The spectrum is stored as
// [even bins | odd bins], each half in the pair-interleaved order of the 4096-point kernel.
constexpr int LDS8K = 8192;

__device__ __forceinline__ void transform_pair8k(cx<float> (&v)[32], cx<float> *lds, const cx<float> *twB_, const cx<float> *twA_,
                                                 const v4f *__restrict__ Hq, v2f wj, int j)
{
    const v2f Wc = {0.92387953251128675613f, 0.38268343236508977173f}, Wr = {0.70710678118654752440f, 0.70710678118654752440f};
    const v2f *twB = (const v2f *)twB_, *twA = (const v2f *)twA_;
    v2f a[16], b[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const v2f lo = __builtin_bit_cast(v2f, v[t]), hi = __builtin_bit_cast(v2f, v[t + 16]);
        const v2f w = t ? pk::pk_cmul<false>(wj, twB[16 * t + 8]) : wj;
        a[t] = lo + hi;
        b[t] = pk::pk_cmul<false>(lo - hi, w);
    }
    pk::fft4096_pk<false>(a, (v2f *)lds, twB, twA, j, Wc, Wr);
    {
        v4f q[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) q[m] = Hq[m * 256 + j];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            a[2 * m] = pk::pk_cmul<false>(a[2 * m], v2f{q[m].x, q[m].y});
            a[2 * m + 1] = pk::pk_cmul<false>(a[2 * m + 1], v2f{q[m].z, q[m].w});
        }
    }
    pk::fft4096_pk<false>(b, (v2f *)lds, twB, twA, j, Wc, Wr);
    {
        v4f q[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) q[m] = Hq[(8 + m) * 256 + j];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            b[2 * m] = pk::pk_cmul<false>(b[2 * m], v2f{q[m].x, q[m].y});
            b[2 * m + 1] = pk::pk_cmul<false>(b[2 * m + 1], v2f{q[m].z, q[m].w});
        }
    }
    pk::fft4096_pk<true>(a, (v2f *)lds, twB, twA, j, Wc, Wr);
    pk::fft4096_pk<true>(b, (v2f *)lds, twB, twA, j, Wc, Wr);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const v2f w = t ? pk::pk_cmul<false>(wj, twB[16 * t + 8]) : wj;
        const v2f bw = pk::pk_cmul<true>(b[t], w);
        v[t] = __builtin_bit_cast(cx<float>, a[t] + bw);
        v[t + 16] = __builtin_bit_cast(cx<float>, a[t] - bw);
    }
}

// float64 (and the unpacked cross-check): the same radix-2 step with the plain transform; spectrum [even | odd] in natural order
template <typename R>
__device__ __forceinline__ void transform_pair8k_plain(cx<R> (&v)[32], cx<R> *lds, const cx<R> *twB, const cx<R> *twA,
                                                       const cx<R> *__restrict__ Hs, cx<R> wj, int j)
{
    cx<R> a[16], b[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const cx<R> w = t ? cmul(wj, twB[16 * t + 8]) : wj;
        a[t] = cadd(v[t], v[t + 16]);
        b[t] = cmul(csub(v[t], v[t + 16]), w);
    }
    fft4096<R, false>(a, lds, twB, twA, j);
#pragma unroll
    for (int t = 0; t < 16; ++t) a[t] = cmul(a[t], Hs[j + 256 * t]);
    fft4096<R, false>(b, lds, twB, twA, j);
#pragma unroll
    for (int t = 0; t < 16; ++t) b[t] = cmul(b[t], Hs[4096 + j + 256 * t]);
    fft4096<R, true>(a, lds, twB, twA, j);
    fft4096<R, true>(b, lds, twB, twA, j);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const cx<R> w = t ? cmul(wj, twB[16 * t + 8]) : wj;
        const cx<R> bw = cmulc(b[t], w);
        v[t] = cadd(a[t], bw);
        v[t + 16] = csub(a[t], bw);
    }
}

template <typename R>
__global__ void __launch_bounds__(256, sizeof(R) == 4 ? 4 : 2)
ols_lds8192_kernel(const R *__restrict__ x, R *__restrict__ y, const cx<R> *__restrict__ Hs, const cx<R> *__restrict__ tw256g,
                   const cx<R> *__restrict__ t4log, const cx<R> *__restrict__ w8kg, Geom<R> g, int64_t npairs, int64_t per_xcd)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cx<R> *lds = (cx<R> *)smem;
    cx<R> *twB = lds + LDS_N + LDS_N / 16;
    cx<R> *twA = twB + 256;
    const int j = threadIdx.x;
    twB[j] = tw256g[((j >> 4) * (j & 15)) & 255];
    twA[j] = t4log[j];
    const int64_t pair = (int64_t)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if ((int64_t)(blockIdx.x >> 3) >= per_xcd || pair >= npairs) return;
    const cx<R> wj = w8kg[j];
    __syncthreads();
    cx<R> v[32];
    const PairAt<R> p(pair, g);
    fetch_pair<R, LDS8K, 32>(v, x, g, p, j);
    const bool nan_b = sanitize_partner<R, 32>(v, p);
    if constexpr (sizeof(R) == 4)
        transform_pair8k(v, lds, twB, twA, (const v4f *)Hs, __builtin_bit_cast(v2f, wj), j);
    else
        transform_pair8k_plain<R>(v, lds, twB, twA, Hs, wj, j);
    poison_partner<R, 32>(v, nan_b);
    store_pair<R, LDS8K, 32>(v, y, g, p, j, smem);
}

// ---- 16 384 points in the same 256-thread workgroup (float32; 4096 < K <= 8192 on long rows) --------------------------------
// The same construction one level up: a radix-4 step in registers around FOUR 4096-point transforms (thread j holds
// z[j + 256 t], t < 64: u_r = sum_q z[n + 4096 q] W4^(q r), times W16384^(n r), FFT_4096(u_r) = the bins 4 m + r).
// W16384^((j + 256 t) r) = W16384^(j r) W64^(t r): three table entries per thread times W256^(4 t r) = twB[16 t + 4 r] from LDS.
// 128 VGPRs of data, 198 in all: two workgroups per CU instead of four, still well ahead of the three-pass pipeline
// (64 x 2.88 M: 4097 / 5000 / 6000 / 8192 taps 0.50 / 0.52 / 0.56 / 0.67 ms against 0.84 / 0.85 / 0.89 / 0.91).
// Spectrum: [r][the pair-interleaved order of the 4096-point kernel over m].
__device__ __forceinline__ void transform_pair16k_r4(cx<float> (&v)[64], cx<float> *lds, const cx<float> *twB_, const cx<float> *twA_,
                                                     const v4f *__restrict__ Hq, const v2f *__restrict__ w16kg, int j)
{
    const v2f Wc = {0.92387953251128675613f, 0.38268343236508977173f}, Wr = {0.70710678118654752440f, 0.70710678118654752440f};
    const v2f *twB = (const v2f *)twB_, *twA = (const v2f *)twA_;
    v2f u[4][16];
    {
        const v2f w1 = w16kg[j], w2 = w16kg[512 + j], w3 = w16kg[1024 + j];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            v2f a0 = __builtin_bit_cast(v2f, v[t]), a1 = __builtin_bit_cast(v2f, v[t + 16]);
            v2f a2 = __builtin_bit_cast(v2f, v[t + 32]), a3 = __builtin_bit_cast(v2f, v[t + 48]);
            pk::pk_dft4<false, false>(a0, a1, a2, a3);
            u[0][t] = a0;
            u[1][t] = pk::pk_cmul<false>(a1, t ? pk::pk_cmul<false>(w1, twB[16 * t + 4]) : w1);
            u[2][t] = pk::pk_cmul<false>(a2, t ? pk::pk_cmul<false>(w2, twB[16 * t + 8]) : w2);
            u[3][t] = pk::pk_cmul<false>(a3, t ? pk::pk_cmul<false>(w3, twB[16 * t + 12]) : w3);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        pk::fft4096_pk<false>(u[r], (v2f *)lds, twB, twA, j, Wc, Wr);
        v4f q[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) q[m] = Hq[(8 * r + m) * 256 + j];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            u[r][2 * m] = pk::pk_cmul<false>(u[r][2 * m], v2f{q[m].x, q[m].y});
            u[r][2 * m + 1] = pk::pk_cmul<false>(u[r][2 * m + 1], v2f{q[m].z, q[m].w});
        }
        pk::fft4096_pk<true>(u[r], (v2f *)lds, twB, twA, j, Wc, Wr);
    }
    {
        const v2f w1 = w16kg[j], w2 = w16kg[512 + j], w3 = w16kg[1024 + j];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            v2f a0 = u[0][t];
            v2f a1 = pk::pk_cmul<true>(u[1][t], t ? pk::pk_cmul<false>(w1, twB[16 * t + 4]) : w1);
            v2f a2 = pk::pk_cmul<true>(u[2][t], t ? pk::pk_cmul<false>(w2, twB[16 * t + 8]) : w2);
            v2f a3 = pk::pk_cmul<true>(u[3][t], t ? pk::pk_cmul<false>(w3, twB[16 * t + 12]) : w3);
            pk::pk_dft4<true, false>(a0, a1, a2, a3);
            v[t] = __builtin_bit_cast(cx<float>, a0);
            v[t + 16] = __builtin_bit_cast(cx<float>, a1);
            v[t + 32] = __builtin_bit_cast(cx<float>, a2);
            v[t + 48] = __builtin_bit_cast(cx<float>, a3);
        }
    }
}

__global__ void __launch_bounds__(256, 2)          // 198 VGPRs; at three workgroups per CU (168) it spills and runs 5-8 % slower
ols_lds16k_r4_kernel(const float *__restrict__ x, float *__restrict__ y, const v4f *__restrict__ Hq, const cx<float> *__restrict__ tw256g,
                     const cx<float> *__restrict__ t4log, const v2f *__restrict__ w16kg, Geom<float> g, int64_t npairs, int64_t per_xcd)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cx<float> *lds = (cx<float> *)smem;
    cx<float> *twB = lds + LDS_N + LDS_N / 16;
    cx<float> *twA = twB + 256;
    const int j = threadIdx.x;
    twB[j] = tw256g[((j >> 4) * (j & 15)) & 255];
    twA[j] = t4log[j];
    const int64_t pair = (int64_t)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if ((int64_t)(blockIdx.x >> 3) >= per_xcd || pair >= npairs) return;
    __syncthreads();
    cx<float> v[64];
    const PairAt<float> p(pair, g);
    fetch_pair<float, 16384, 64>(v, x, g, p, j);
    const bool nan_b = sanitize_partner<float, 64>(v, p);
    transform_pair16k_r4(v, lds, twB, twA, Hq, w16kg, j);
    poison_partner<float, 64>(v, nan_b);
    store_pair<float, 16384, 64>(v, y, g, p, j, smem);
}

// ---- 16 384 points in a 512-thread workgroup (round 6): the radix-4 construction with HALF the registers per thread --------------
// ols_lds16k_r4_kernel holds 64 complex values per thread (198 VGPRs: two workgroups = 8 wavefronts per CU).  Here thread j of 512
// holds z[j + 512 t], t < 32 -- the footprint of the 8192-point kernel, so 16 wavefronts per CU fit -- and the two halves of the
// workgroup run two of the four 4096-point transforms AT THE SAME TIME, each through its own exchange buffer:
//   radix-4 step in registers (n, n + 4096 q are t0 + 8 q of one thread), times W16384^(n r) = W16384^(j r) W256^(8 r t0)
//   -> u_r[j + 512 t0];  half h transforms r = 2 h and 2 h + 1 and needs u_r[jj + 256 t'] (jj = j & 255): the even t' of half 0 and
//   the odd t' of half 1 are its own, the others sit in thread j ^ 256 -- one exchange of 16 values per thread through the (still
//   free) transform buffers, and the mirror image on the way back.  Spectrum and tables: those of the 256-thread radix-4 kernel.
__device__ __forceinline__ void transform_pair16k_w8(cx<float> (&v)[32], v2f *lbuf, const cx<float> *twB_, const cx<float> *twA_,
                                                     const v4f *__restrict__ Hq, const v2f *__restrict__ w16kg, int j)
{
    const v2f Wc = {0.92387953251128675613f, 0.38268343236508977173f}, Wr = {0.70710678118654752440f, 0.70710678118654752440f};
    const v2f *twB = (const v2f *)twB_, *twA = (const v2f *)twA_;
    const int jj = j & 255;
    const int half = __builtin_amdgcn_readfirstlane(j >> 8);          // wave-uniform: wavefronts 0-3 / 4-7
    v2f *lds = lbuf + half * (LDS_N + LDS_N / 16);                     // this half's 4096-point exchange buffer
    v2f *X = lbuf;                                                     // both buffers as one [16][512] exchange (8192 of 8704 slots)
    v2f u[4][8];
    {
        const v2f w1 = w16kg[j], w2 = w16kg[512 + j], w3 = w16kg[1024 + j];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            v2f a0 = __builtin_bit_cast(v2f, v[t]), a1 = __builtin_bit_cast(v2f, v[t + 8]);
            v2f a2 = __builtin_bit_cast(v2f, v[t + 16]), a3 = __builtin_bit_cast(v2f, v[t + 24]);
            pk::pk_dft4<false, false>(a0, a1, a2, a3);
            u[0][t] = a0;                                              // W256^(8 r t) = twB[16 (4 r) + 2 t]
            u[1][t] = pk::pk_cmul<false>(a1, t ? pk::pk_cmul<false>(w1, twB[16 * 4 + 2 * t]) : w1);
            u[2][t] = pk::pk_cmul<false>(a2, t ? pk::pk_cmul<false>(w2, twB[16 * 8 + 2 * t]) : w2);
            u[3][t] = pk::pk_cmul<false>(a3, t ? pk::pk_cmul<false>(w3, twB[16 * 12 + 2 * t]) : w3);
        }
    }
    v2f A[16], B[16];
    if (half == 0) {
#pragma unroll
        for (int t = 0; t < 8; ++t) { X[t * 512 + j] = u[2][t]; X[(8 + t) * 512 + j] = u[3][t]; }
    } else {
#pragma unroll
        for (int t = 0; t < 8; ++t) { X[t * 512 + j] = u[0][t]; X[(8 + t) * 512 + j] = u[1][t]; }
    }
    __syncthreads();
    if (half == 0) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            A[2 * t] = u[0][t]; A[2 * t + 1] = X[t * 512 + (j ^ 256)];
            B[2 * t] = u[1][t]; B[2 * t + 1] = X[(8 + t) * 512 + (j ^ 256)];
        }
    } else {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            A[2 * t] = X[t * 512 + (j ^ 256)]; A[2 * t + 1] = u[2][t];
            B[2 * t] = X[(8 + t) * 512 + (j ^ 256)]; B[2 * t + 1] = u[3][t];
        }
    }
    __syncthreads();
    const v4f *Hh = Hq + (size_t)(16 * half) * 256;                    // spectrum quarters r = 2 h (A) and 2 h + 1 (B)
    pk::fft4096_pk<false>(A, lds, twB, twA, jj, Wc, Wr);
    {
        v4f q[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) q[m] = Hh[m * 256 + jj];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            A[2 * m] = pk::pk_cmul<false>(A[2 * m], v2f{q[m].x, q[m].y});
            A[2 * m + 1] = pk::pk_cmul<false>(A[2 * m + 1], v2f{q[m].z, q[m].w});
        }
    }
    pk::fft4096_pk<true>(A, lds, twB, twA, jj, Wc, Wr);
    pk::fft4096_pk<false>(B, lds, twB, twA, jj, Wc, Wr);
    {
        v4f q[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) q[m] = Hh[(8 + m) * 256 + jj];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            B[2 * m] = pk::pk_cmul<false>(B[2 * m], v2f{q[m].x, q[m].y});
            B[2 * m + 1] = pk::pk_cmul<false>(B[2 * m + 1], v2f{q[m].z, q[m].w});
        }
    }
    pk::fft4096_pk<true>(B, lds, twB, twA, jj, Wc, Wr);              // ends with a barrier: both buffers are free
    // back: half 0 keeps the even t' of its two sub-sequences and hands the odd ones to thread j + 256, half 1 the other way round
    if (half == 0) {
#pragma unroll
        for (int t = 0; t < 8; ++t) { X[t * 512 + j] = A[2 * t + 1]; X[(8 + t) * 512 + j] = B[2 * t + 1]; }
    } else {
#pragma unroll
        for (int t = 0; t < 8; ++t) { X[t * 512 + j] = A[2 * t]; X[(8 + t) * 512 + j] = B[2 * t]; }
    }
    __syncthreads();
    if (half == 0) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            u[0][t] = A[2 * t]; u[1][t] = B[2 * t];
            u[2][t] = X[t * 512 + (j ^ 256)]; u[3][t] = X[(8 + t) * 512 + (j ^ 256)];
        }
    } else {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            u[0][t] = X[t * 512 + (j ^ 256)]; u[1][t] = X[(8 + t) * 512 + (j ^ 256)];
            u[2][t] = A[2 * t + 1]; u[3][t] = B[2 * t + 1];
        }
    }
    {
        const v2f w1 = w16kg[j], w2 = w16kg[512 + j], w3 = w16kg[1024 + j];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            v2f a0 = u[0][t];
            v2f a1 = pk::pk_cmul<true>(u[1][t], t ? pk::pk_cmul<false>(w1, twB[16 * 4 + 2 * t]) : w1);
            v2f a2 = pk::pk_cmul<true>(u[2][t], t ? pk::pk_cmul<false>(w2, twB[16 * 8 + 2 * t]) : w2);
            v2f a3 = pk::pk_cmul<true>(u[3][t], t ? pk::pk_cmul<false>(w3, twB[16 * 12 + 2 * t]) : w3);
            pk::pk_dft4<true, false>(a0, a1, a2, a3);
            v[t] = __builtin_bit_cast(cx<float>, a0);
            v[t + 8] = __builtin_bit_cast(cx<float>, a1);
            v[t + 16] = __builtin_bit_cast(cx<float>, a2);
            v[t + 24] = __builtin_bit_cast(cx<float>, a3);
        }
    }
}

constexpr size_t lds16k_w8_bytes() { return (size_t)(2 * (LDS_N + LDS_N / 16) + 512) * sizeof(cx<float>); }

__global__ void __launch_bounds__(512, 4)         // 4 wavefronts per SIMD = two workgroups per CU (LDS: 2 x 73.7 KB)
ols_lds16k_w8_kernel(const float *__restrict__ x, float *__restrict__ y, const v4f *__restrict__ Hq, const cx<float> *__restrict__ tw256g,
                     const cx<float> *__restrict__ t4log, const v2f *__restrict__ w16kg, Geom<float> g, int64_t npairs, int64_t per_xcd)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cx<float> *lbuf = (cx<float> *)smem;                              // two [4096 + 256] exchange buffers
    cx<float> *twB = lbuf + 2 * (LDS_N + LDS_N / 16);
    cx<float> *twA = twB + 256;
    const int j = threadIdx.x;
    if (j < 256) { twB[j] = tw256g[((j >> 4) * (j & 15)) & 255]; twA[j] = t4log[j]; }
    const int64_t pair = (int64_t)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if ((int64_t)(blockIdx.x >> 3) >= per_xcd || pair >= npairs) return;
    __syncthreads();
    cx<float> v[32];
    const PairAt<float> p(pair, g);
    fetch_pair<float, 16384, 32>(v, x, g, p, j);
    const bool nan_b = sanitize_partner<float, 32>(v, p);
    transform_pair16k_w8(v, (v2f *)lbuf, twB, twA, Hq, w16kg, j);
    poison_partner<float, 32>(v, nan_b);
    store_pair<float, 16384, 32>(v, y, g, p, j, smem);
}

// ---- second block size: 16 384 points, one 1024-thread workgroup per pair of frames (float32; 2048 < K <= 8192) ----------
// fftpk16k.h: radix 16 x 16 x 16 x 4, three exchanges per direction through 136 KB of LDS -- ONE workgroup per CU, sixteen
// wavefronts that run their phases in lockstep.  Measured against the three-pass pipeline on rows long enough for it
// (64 x 2.88 M, 4096 taps): 0.97 ms against 0.79 -- nothing overlaps the load, exchange and store phases of the one
// workgroup, and the register prefetch of the next pair that would (a second set of 32 VGPRs on top of 108) spills at the
// 128 registers sixteen wavefronts leave (1.42 ms).  So this kernel serves the rows the three-pass pipeline does NOT take --
// shorter than its 65 536-sample block, where the alternative is the five-launch rocFFT path -- and TFX_OLS_LDS16K=2 forces it
// everywhere.  The spectrum is stored in the transform's "spectral ownership" (thread r, register s <-> bin
// (r & 255) + 256 (4 (r >> 8) + s / 4) + 4096 (s % 4)), register pairs interleaved for 16-byte loads.
constexpr int LDS16K = 16384;
constexpr size_t lds16k_bytes() { return (size_t)(pk::X16K_SLOTS + pk::X16K_TABLES) * sizeof(v2f); }

__device__ __forceinline__ void transform_pair16k(cx<float> (&v)[16], v2f *L, const pk::Tab16k &tb, const v4f *__restrict__ Hq, int j)
{
    const v2f Wc = {0.92387953251128675613f, 0.38268343236508977173f}, Wr = {0.70710678118654752440f, 0.70710678118654752440f};
    v2f u[16];
    v4f hq[8];
#pragma unroll
    for (int t = 0; t < 16; ++t) u[t] = __builtin_bit_cast(v2f, v[t]);
    pk::fft16384_fwd(u, L, tb, j, Wc, Wr);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 8; ++m) hq[m] = Hq[m * 1024 + j];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        u[2 * m] = pk::pk_cmul<false>(u[2 * m], v2f{hq[m].x, hq[m].y});
        u[2 * m + 1] = pk::pk_cmul<false>(u[2 * m + 1], v2f{hq[m].z, hq[m].w});
    }
    __builtin_amdgcn_sched_barrier(0);
    pk::fft16384_inv(u, L, tb, j, Wc, Wr);
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = __builtin_bit_cast(cx<float>, u[t]);
}

__global__ void __launch_bounds__(1024, 4)
ols_lds16k_kernel(const float *__restrict__ x, float *__restrict__ y, const v4f *__restrict__ Hq, const v2f *__restrict__ gtab,
                  Geom<float> g, int64_t npairs, int64_t per_xcd)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    v2f *L = (v2f *)smem;
    const int j = threadIdx.x;
    const pk::Tab16k tb = pk::fill_tab16k(L + pk::X16K_SLOTS, gtab, j);
    const int64_t wpx = gridDim.x >> 3, m = blockIdx.x >> 3;
    const int64_t lo = (int64_t)(blockIdx.x & 7u) * per_xcd;
    const int64_t hi = lo + per_xcd < npairs ? lo + per_xcd : npairs;
    __syncthreads();
    for (int64_t pair = lo + m; pair < hi; pair += wpx) {
        cx<float> v[16];
        const PairAt<float> p(pair, g);
        fetch_pair<float, LDS16K>(v, x, g, p, j);
        const bool nan_b = sanitize_partner<float, 16>(v, p);
        transform_pair16k(v, L, tb, Hq, j);
        poison_partner<float, 16>(v, nan_b);
        store_pair<float, LDS16K>(v, y, g, p, j, smem);
        __syncthreads();                         // the statistic's scratch is the transform buffer
    }
}

// ---- host: per-filter tables ---------------------------------------------------------------------
// A plan's device buffer (spectrum | tables, one allocation) is OWNED by the plan objects that point into it: get_plan hands
// out a copy, so a caller keeps the buffer alive until its launches are enqueued, whatever another host thread evicts in
// the meantime; the last owner's hipFree synchronises the device, i.e. runs behind those launches (round 4 returned a
// plain copy and freed every spectrum when the 66th filter arrived: a launch on a freed buffer).
struct Plan {
    void *Hs = nullptr, *tw256 = nullptr, *t4lo = nullptr, *w8k = nullptr;
    std::shared_ptr<void> owner;
};
static std::mutex g_mu;
static std::map<std::vector<char>, Plan> g_plans;
static std::map<std::vector<char>, uint64_t> g_used;                     // last use of a key (eviction: least recently used first)
static uint64_t g_tick = 0;
static const std::vector<char> *g_last_key[TFX_MAX_DEVICES] = {};       // std::map nodes are stable
static const Plan *g_last[TFX_MAX_DEVICES] = {};
constexpr size_t LDS_PLAN_CAP = 64;

template <typename R> static void *upload(const std::vector<cx<R>> &h)
{
    void *d = nullptr;
    TFX_HIP(hipMalloc(&d, h.size() * sizeof(cx<R>)));
    TFX_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(cx<R>), hipMemcpyHostToDevice));
    return d;
}

// kind: 0 = 4096 points, 1 = 8192 (radix 2 around 4096), 2 = 16 384 in a 1024-thread workgroup, 3 = 16 384 as radix 4 around 4096
template <typename R> static Plan get_plan(const R *kf, int64_t K, int64_t lead, int N, int kind, hipStream_t stream)
{
    std::lock_guard<std::mutex> lk(g_mu);
    const int dev = current_device();
    const size_t nb = (size_t)K * sizeof(R);
    const char tail[3] = {(char)(sizeof(R) + 16 * kind), (char)lead, (char)dev};
    if (const std::vector<char> *lk_ = g_last_key[dev]) {       // steady state: one memcmp, no key construction
        if (lk_->size() == nb + 3 && memcmp(lk_->data(), kf, nb) == 0 && memcmp(lk_->data() + nb, tail, 3) == 0) {
            g_used[*lk_] = ++g_tick;
            return *g_last[dev];
        }
    }
    std::vector<char> key((const char *)kf, (const char *)kf + nb);
    key.insert(key.end(), tail, tail + 3);
    auto it = g_plans.find(key);
    if (it == g_plans.end()) {
        // a new filter costs a device allocation and a blocking upload: not inside a stream capture
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        TFX_CHECK(!(hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone),
                  "overlap-save: first use of this filter (%lld taps) inside a stream capture -- run it once before capturing "
                  "(its spectrum is uploaded with a blocking copy)", (long long)K);
        while (g_plans.size() >= LDS_PLAN_CAP) {              // least recently used entry goes; its buffer lives on with whoever still holds it
            auto victim = g_used.begin();
            for (auto u = g_used.begin(); u != g_used.end(); ++u)
                if (u->second < victim->second) victim = u;
            for (int d = 0; d < TFX_MAX_DEVICES; ++d)
                if (g_last_key[d] && *g_last_key[d] == victim->first) { g_last_key[d] = nullptr; g_last[d] = nullptr; }
            g_plans.erase(victim->first);
            g_used.erase(victim);
        }
        // conj(FFT(taps behind `lead` zeros, zero padded to N)) / N in float64  (_fftconv.py:123-124,131 + irfft scaling)
        std::vector<double> re((size_t)N, 0.0), im((size_t)N, 0.0);
        for (int64_t i = 0; i < K; ++i) re[(size_t)(lead + i)] = (double)kf[i];
        host_fft_f64(re, im);
        std::vector<cx<R>> hs((size_t)N);
        auto W = [](double num, double den) {
            const double a = -2.0 * M_PI * num / den;
            cx<R> w; w.x = (R)cos(a); w.y = (R)sin(a);
            return w;
        };
        Plan p;
        if (kind == 2) {
            // spectral ownership of fftpk16k.h: thread r, register s <-> bin (r & 255) + 256 (4 (r >> 8) + s / 4) + 4096 (s % 4);
            // registers (2 m, 2 m + 1) of a thread sit next to each other: one 16-byte load
            for (int r = 0; r < 1024; ++r)
                for (int sl = 0; sl < 16; ++sl) {
                    const int k = (r & 255) + 256 * (4 * (r >> 8) + (sl >> 2)) + 4096 * (sl & 3);
                    const int at = ((sl >> 1) * 1024 + r) * 2 + (sl & 1);
                    hs[at].x = (R)(re[k] / N); hs[at].y = (R)(-im[k] / N);
                }
            std::vector<cx<R>> tab(pk::X16K_TABLES);          // [twA 256 | twB 256 | tA 64 | tB 64 | tC 64]
            for (int a = 0; a < 16; ++a)
                for (int b = 0; b < 16; ++b) { tab[16 * a + b] = W(a * b, 4096); tab[256 + 16 * a + b] = W(a * b, 256); }
            for (int d = 0; d < 4; ++d)
                for (int b = 0; b < 16; ++b) {
                    tab[512 + 16 * d + b] = W(d * b, 16384); tab[576 + 16 * d + b] = W(d * b, 1024); tab[640 + 16 * d + b] = W(d * b, 64);
                }
            hs.insert(hs.end(), tab.begin(), tab.end());
            p.Hs = upload<R>(hs);                             // one allocation, one copy: spectrum | tables
            p.owner = std::shared_ptr<void>(p.Hs, [](void *q) { (void)hipFree(q); });
            p.tw256 = (char *)p.Hs + (size_t)N * sizeof(cx<R>);
            p.t4lo = nullptr;
        } else {
            std::vector<cx<R>> t256(256), t4(256);
            if (kind == 3) {
                // [r][...]: bin 4 m + r, m = jj + 256 t, each quarter pair-interleaved like the 4096-point spectrum
                for (int k = 0; k < LDS16K; ++k) {
                    const int r = k & 3, m = k >> 2, t = m >> 8, jj = m & 255;
                    const int at = ((r * 8 + (t >> 1)) * 256 + jj) * 2 + (t & 1);
                    hs[at].x = (R)(re[k] / LDS16K); hs[at].y = (R)(-im[k] / LDS16K);
                }
            } else if (kind == 1) {
                // [even bins | odd bins]: bin 2 m + h, m = jj + 256 t, each half pair-interleaved like the 4096-point spectrum
                for (int k = 0; k < LDS8K; ++k) {
                    const int h = k & 1, m = k >> 1, t = m >> 8, jj = m & 255;
                    const int at = sizeof(R) == 4 ? ((h * 8 + (t >> 1)) * 256 + jj) * 2 + (t & 1) : h * 4096 + m;
                    hs[at].x = (R)(re[k] / LDS8K); hs[at].y = (R)(-im[k] / LDS8K);
                }
            } else
            for (int k = 0; k < LDS_N; ++k) {
                // float32: thread j multiplies elements j + 256 t; the pairs (t, t + 1) sit next to each other so it loads them 16 bytes at a time
                const int t = k >> 8, jj = k & 255;
                const int at = sizeof(R) == 4 ? ((t >> 1) * 256 + jj) * 2 + (t & 1) : k;
                hs[at].x = (R)(re[k] / LDS_N); hs[at].y = (R)(-im[k] / LDS_N);
            }
            for (int i = 0; i < 256; ++i) t256[i] = W(i, 256);
            for (int t = 0; t < 16; ++t)
                for (int a2 = 0; a2 < 16; ++a2) t4[16 * t + a2] = W(t * a2, 4096);
            hs.insert(hs.end(), t256.begin(), t256.end());    // one allocation, one copy: spectrum | W256 | W4096 table | W8192^j
            hs.insert(hs.end(), t4.begin(), t4.end());
            if (kind == 1)
                for (int i = 0; i < 256; ++i) hs.push_back(W(i, 8192));
            if (kind == 3)
                for (int r = 1; r < 4; ++r)
                    for (int i = 0; i < 512; ++i) hs.push_back(W(r * i, 16384));      // [r - 1][i]: the 256-thread kernel reads i < 256, the 512-thread one all
            p.Hs = upload<R>(hs);
            p.owner = std::shared_ptr<void>(p.Hs, [](void *q) { (void)hipFree(q); });
            p.tw256 = (char *)p.Hs + (size_t)N * sizeof(cx<R>);
            p.t4lo = (char *)p.tw256 + 256 * sizeof(cx<R>);
            p.w8k = (char *)p.t4lo + 256 * sizeof(cx<R>);
        }
        it = g_plans.emplace(std::move(key), p).first;
    }
    g_used[it->first] = ++g_tick;
    g_last_key[dev] = &it->first;
    g_last[dev] = &it->second;
    return it->second;
}

static int64_t envi(const char *name, int64_t dflt) { return env_i64(name, dflt); }      // read once per process (common.h)

}  // namespace ldsfft

void olslds_clear()
{
    using namespace ldsfft;
    std::lock_guard<std::mutex> lk(g_mu);
    g_plans.clear();                        // the owners free their buffers (hipFree waits for the device)
    g_used.clear();
    for (int d = 0; d < TFX_MAX_DEVICES; ++d) { g_last_key[d] = nullptr; g_last[d] = nullptr; }
}

// taps this path takes: at least half of every block must be valid output -- 4096 points for K < 640 (float32; measured
// equal to the 8192-point block at 512 taps, faster below) and K < 700 (float64), 8192 points above that up to 4096 taps
// (float32, 64 x 2.88 M: 1024 taps 0.36 -> 0.34 ms, 2048 taps 0.49 -> 0.39, 4096 taps 0.85 (three passes) -> 0.52; float64, 32 x 2.88 M: 2048 taps
// 0.60 -> 0.44, 4096 taps 3.0 (rocFFT) -> 0.62), 16 384 points for 4096 < K <= 8192 (float32; TFX_OLS_LDS16K=0 and
// TFX_OLS_LDS16K_R4=0 hand them back to the three-pass pipeline / rocFFT)
bool olslds_supported(int64_t K, int dtype, int64_t L, int64_t *N_out)
{
    if (ldsfft::envi("TFX_OLS_LDS", 1) == 0 || ldsfft::envi("TFX_OLS_NATIVE", 1) == 0) return false;
    const int64_t lg = ldsfft::envi("TFX_FFT_LOG2N", 0);
    if (lg != 0 && lg != 12 && lg != 13 && lg != 14) return false;         // a forced block size of another path
    int64_t N = 0;
    const int64_t use16k = ldsfft::envi("TFX_OLS_LDS16K", 1);                // 0 never, 1 where the three-pass pipeline does not reach, 2 always
    // taps from which the 8192-point block pays (0: never): measured equal at 512 taps in float32 and in float64
    const int64_t min8k = ldsfft::envi("TFX_OLS_LDS8K_MINK", dtype == TFX_F32 ? 640 : 700);
    const bool can8k = K >= 1 && K <= ldsfft::LDS8K / 2 && min8k > 0 && (lg == 0 || lg == 13);
    // rows shorter than 65 536 samples are a handful of workgroups that all run at once: the call takes as long as ONE workgroup,
    // so the smallest block that fits wins there ([2, 44100], 1500 taps: 8 us at 4096 points, 15 us at 8192)
    // from ~3300 taps the radix-4 kernel at 16 384 points overtakes it on long rows (4096 taps: 0.47 against 0.53 ms, 3000 taps:
    // 0.45 against 0.44) although it runs two workgroups per CU instead of four
    const int64_t min16k = ldsfft::envi("TFX_OLS_LDS16K_MINK", 3400);
    const bool r4_long = dtype == TFX_F32 && lg == 0 && L >= 65536 && use16k == 1 && ldsfft::envi("TFX_OLS_LDS16K_R4", 1) >= 1 &&
                         min16k > 0 && K >= min16k && K <= ldsfft::LDS16K / 2;
    if (r4_long) N = ldsfft::LDS16K;
    else if (can8k && (lg == 13 || (K >= min8k && (L >= 65536 || K > ldsfft::LDS_N / 2)))) N = ldsfft::LDS8K;
    else if (K >= 1 && K <= ldsfft::LDS_N / 2 && lg != 14 && lg != 13) N = ldsfft::LDS_N;
    else if (K >= 1 && K <= ldsfft::LDS16K / 2 && dtype == TFX_F32 && (lg == 0 || lg == 14) &&
             (use16k >= 2 || lg == 14 || (use16k == 1 && (L < 65536 || ldsfft::envi("TFX_OLS_LDS16K_R4", 1) >= 1)) ||
              ldsfft::envi("TFX_OLS_LDS16K_R4", 1) >= 2))
        N = ldsfft::LDS16K;
    if (!N) return false;
    if (N_out) *N_out = N;
    return true;
}

// frame geometry shared with tfx_ols_plan_info: `lead` zero taps in front of the flipped kernel move the frame starts
// onto 128-byte lines when the rows themselves are aligned, and the hop is rounded down to whole lines
void olslds_geometry(int64_t K, int64_t Tn, int64_t pl, int64_t pr, int elem_bytes, int64_t N, int64_t *lead_out, int64_t *S_out)
{
    const int64_t line = 128 / elem_bytes;
    const int64_t Tout = Tn + pl + pr - K + 1;
    const bool align = (Tn % line == 0) && (Tout % line == 0) && ldsfft::envi("TFX_OLS_ALIGN", 1) != 0;
    const int64_t lead = align ? (line - (pl % line)) % line : 0;
    int64_t S = N - (K + lead) + 1;
    if (align && S > 2 * line) S -= S % line;
    *lead_out = lead;
    *S_out = S;
}

template <typename R>
static void olslds_typed(const R *x, R *y, int64_t C, int64_t Tn, const R *kf_host, int64_t K, int64_t pl, int64_t pr,
                         hipStream_t stream, const R *hist, int64_t H, const Epilogue *ep)
{
    using namespace ldsfft;
    Geom<R> g;
    const int64_t L = Tn + pl + pr;
    g.Tn = Tn; g.Tout = L - K + 1;
    g.hist = hist; g.H = hist ? H : 0;
    g.ep_gain = ep ? (R)ep->gain : (R)1; g.ep_scale = ep ? ep->scale : 0; g.ep_clamp = ep ? ep->clamp : 0;
    g.ep_stat = ep ? ep->stat_mode : -1; g.ep_partial = nullptr;
    g.nt = (int)envi("TFX_OLS_LDS_NT", 2);
    int64_t lead = 0, N = 0;
    TFX_CHECK(olslds_supported(K, sizeof(R) == 4 ? TFX_F32 : TFX_F64, Tn + pl + pr, &N), "olslds_forward: %lld taps are not for this path", (long long)K);
    olslds_geometry(K, Tn, pl, pr, (int)sizeof(R), N, &lead, &g.S);
    g.pad_left = pl + lead;
    g.F = ceil_div(g.Tout, g.S);
    g.nframes = C * g.F;
    // 16 384 points: the 1024-thread workgroup on rows the three-pass pipeline does not reach (few pairs: the sixteen wavefronts
    // of one pair run side by side), four 4096-point transforms in a 256-thread workgroup on long rows (TFX_OLS_LDS16K_R4: 0 never, 2 always)
    const int64_t r4 = envi("TFX_OLS_LDS16K_R4", 1);
    const bool use_r4 = N == LDS16K && sizeof(R) == 4 && (r4 >= 2 || (r4 == 1 && L >= 65536 && envi("TFX_OLS_LDS16K", 1) < 2));
    const int kind = N == LDS_N ? 0 : N == LDS8K ? 1 : use_r4 ? 3 : 2;
    const Plan plan = get_plan<R>(kf_host, K, lead, (int)N, kind, stream);      // holds its buffer until this function has enqueued its launch
    const int64_t npairs = ceil_div(g.nframes, 2);
    if (g.ep_stat >= 0) g.ep_partial = (double *)scratch("olslds_ep_partial", (size_t)g.nframes * sizeof(double), stream);
    const int dev = current_device();
    const int64_t per_xcd = ceil_div(npairs, 8);
    TFX_CHECK(per_xcd * 8 < ((int64_t)1 << 31), "fft_conv_forward: too many frames for one launch");
    // raise the kernel's dynamic-LDS limit once per device, then launch `grid` workgroups of `threads`
    auto launch = [&](auto kernel, bool &ready, const char *name, size_t lds, int64_t grid, int threads, auto... args) {
        if (!ready) {
            TFX_HIP(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            ready = true;
        }
        ProfScope ps(name, stream);
        hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(threads), lds, stream, args...);
        TFX_HIP(hipGetLastError());
    };
    static bool ready[4][TFX_MAX_DEVICES] = {};            // per kernel kind of this instantiation
    bool done = false;
    if constexpr (sizeof(R) == 4) {
        if (kind == 3 && envi("TFX_OLS_LDS16K_W8", 1) != 0) {
            static bool ready_w8[TFX_MAX_DEVICES] = {};
            launch(ols_lds16k_w8_kernel, ready_w8[dev], "ols_lds16k_w8_kernel", lds16k_w8_bytes(), per_xcd * 8, 512,
                   (const float *)x, (float *)y, (const v4f *)plan.Hs, (const cx<float> *)plan.tw256, (const cx<float> *)plan.t4lo,
                   (const v2f *)plan.w8k, g, npairs, per_xcd);
            done = true;
        } else if (kind == 3) {
            launch(ols_lds16k_r4_kernel, ready[3][dev], "ols_lds16k_r4_kernel", lds_bytes<float>(), per_xcd * 8, 256,
                   (const float *)x, (float *)y, (const v4f *)plan.Hs, (const cx<float> *)plan.tw256, (const cx<float> *)plan.t4lo,
                   (const v2f *)plan.w8k, g, npairs, per_xcd);
            done = true;
        } else if (kind == 2) {
            static int cus_tab[TFX_MAX_DEVICES] = {};
            if (!cus_tab[dev]) {
                int cus = 0;
                TFX_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
                cus_tab[dev] = std::max(8, cus / 8 * 8);      // one 1024-thread workgroup per CU
            }
            const int64_t grid = std::max<int64_t>(8, std::min<int64_t>(envi("TFX_OLS_LDS16K_GRID", cus_tab[dev]) / 8 * 8, per_xcd * 8));
            launch(ols_lds16k_kernel, ready[2][dev], "ols_lds16k_kernel", lds16k_bytes(), grid, 1024,
                   (const float *)x, (float *)y, (const v4f *)plan.Hs, (const v2f *)plan.tw256, g, npairs, per_xcd);
            done = true;
        }
    }
    if (!done && kind == 1)
        launch(ols_lds8192_kernel<R>, ready[1][dev], "ols_lds8192_kernel", lds_bytes<R>(), per_xcd * 8, 256,
               x, y, (const cx<R> *)plan.Hs, (const cx<R> *)plan.tw256, (const cx<R> *)plan.t4lo, (const cx<R> *)plan.w8k, g, npairs, per_xcd);
    else if (!done)
        launch(ols_lds4096_kernel<R>, ready[0][dev], "ols_lds4096_kernel", lds_bytes<R>(), per_xcd * 8, 256,
               x, y, (const cx<R> *)plan.Hs, (const cx<R> *)plan.tw256, (const cx<R> *)plan.t4lo, g, npairs, per_xcd);
    if (g.ep_stat >= 0)
        stat_finish(g.ep_partial, ep->per_row ? C : 1, ep->per_row ? g.F : g.nframes, g.ep_stat, ep->stat_out, stream);
}

void olslds_forward(const void *x, void *y, int dtype, int64_t C, int64_t Tn, const void *kf_host, int64_t K,
                    int64_t pl, int64_t pr, hipStream_t stream, const void *hist, int64_t H, const Epilogue *ep)
{
    if (dtype == TFX_F32)
        olslds_typed<float>((const float *)x, (float *)y, C, Tn, (const float *)kf_host, K, pl, pr, stream, (const float *)hist, H, ep);
    else
        olslds_typed<double>((const double *)x, (double *)y, C, Tn, (const double *)kf_host, K, pl, pr, stream, (const double *)hist, H, ep);
}

}  // namespace tfx
