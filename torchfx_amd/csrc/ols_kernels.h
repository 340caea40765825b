// ols_kernels.h -- device code of the three-pass overlap-save pipeline (olsnative.hip holds the plans, lanes and the dispatch;
// its header comment describes the decomposition): geometry, radix-16 column passes A / C, pass A with the SOS cascade in front
// (ols_col_fwd16_sos_kernel) and its non-finite fix-up, the row passes B for 256 / 1024 / 4096 / 8192-point rows and the
// kernels that compute a filter's spectrum on the device.  Included by olsnative.hip only.
#pragma once
#include "common.h"
#include "epilogue.h"
#include "fftpk.h"

namespace tfx {

typedef float2 cpx;

__device__ __forceinline__ cpx cmul(cpx a, cpx b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cpx cmulc(cpx a, cpx b) { return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }  // a*conj(b)
__device__ __forceinline__ cpx cadd(cpx a, cpx b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cpx csub(cpx a, cpx b) { return make_float2(a.x - b.x, a.y - b.y); }

template <bool INV>
__device__ __forceinline__ void dft4(cpx &a0, cpx &a1, cpx &a2, cpx &a3)
{
    const cpx s02 = cadd(a0, a2), d02 = csub(a0, a2), s13 = cadd(a1, a3), d13 = csub(a1, a3);
    const cpx id = INV ? make_float2(-d13.y, d13.x) : make_float2(d13.y, -d13.x);   // (+i or -i) * d13
    a0 = cadd(s02, s13);
    a2 = csub(s02, s13);
    a1 = cadd(d02, id);
    a3 = csub(d02, id);
}

__device__ __forceinline__ void wave_sync2()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct OlsGeom {
    int64_t Tn;        // input row length
    int64_t Tout;      // output row length
    int64_t F;         // frames per channel
    int64_t S;         // hop = valid outputs per frame
    int64_t pad_left;  // left zero padding of the framed signal (>= the caller's; rounded up for alignment)
    int64_t out_shift; // = pad_left - caller's pad_left: block output i is y[i - out_shift]
    int64_t nframes;   // C * F
    const float *hist; // streaming: [C, H] samples preceding each row (x[-H .. -1]) instead of zero padding, or null
    int64_t H;
    // epilogue of the inverse column pass on the stored samples (epilogue.h)
    float ep_gain;
    int ep_scale, ep_clamp, ep_stat;
    double *ep_partial;   // [nframes][N2 / 32]: one partial per (frame, column block); row c owns F * N2/32 consecutive ones
    int N2;            // row length (N = 256 * N2)
    int P2;            // row pitch of the workspace T in elements (N2 + pad: breaks the power-of-two stride)
    int sh_on, sh_base; // rows that are not whole 128-byte lines (T % 32 != 0, or a base pointer inside a line): row c's frame grid
                       // moves left by sh(c) = (sh_base + c * Tn) % 32 samples, so every frame still starts on a 128-byte line of
                       // memory; block output i of frame f is then y[f * S + i - sh(c)]  (sh_base = element offset of x in its line)
    int *nf_flag;      // cascade in pass A: [nframes], 1 = the recursion of this frame met a non-finite value (every slot written)
    int *nf_pair;      // plain pass A: [C] zeroed per call, or null: row c's first frame shares its transform with row c - 1's last and
                       // held a non-finite sample (only when a row has an odd number of frames)
    int nt;            // nontemporal hints (TFX_OLS_NT, default 3): 1 = signal loads of pass A, 2 = signal stores of pass C -- the signal
                       // is read once and written once; chain step 7.98 -> 7.87 ms.  (The same hint on the workspace loads of
                       // passes B and C, their last use, changes nothing.)
};

constexpr int OLS_N1 = 256;
constexpr int OLS_CB = 32;      // columns per workgroup in the column passes

// LDS positions of the row passes are padded by one element per 16: the stride-16 writes of a
// radix-16 Stockham stage become conflict-free and every address stays base + immediate.  (An XOR
// swizzle removes the remaining 2-way read conflict but costs 32 computed addresses per stage and
// measured slower.)
__device__ __forceinline__ int pad16(int p) { return p + (p >> 4); }
__device__ __forceinline__ int row_shift(const OlsGeom &g, int64_t c) { return g.sh_on ? (int)(((int64_t)g.sh_base + c * g.Tn) & 31) : 0; }

template <bool INV>
__device__ __forceinline__ void dft16(cpx (&v)[16])
{
    // t = t1 + 4 t2, k = 4 k1 + k2:  W16^(tk) = W4^(t1 k1) W16^(t1 k2) W4^(t2 k2)
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
#pragma unroll
    for (int t1 = 0; t1 < 4; ++t1) dft4<INV>(v[t1], v[t1 + 4], v[t1 + 8], v[t1 + 12]);
    // v[t1 + 4 k2] *= W16^(t1 k2)   (forward: exp(-i pi n/8); inverse: conjugate)
    auto tw = [&](cpx &x, float c, float sn) {           // multiply by (c - i sn) forward, (c + i sn) inverse
        const float s_ = INV ? -sn : sn;
        x = make_float2(x.x * c + x.y * s_, x.y * c - x.x * s_);
    };
    tw(v[1 + 4], C1, S1);  tw(v[1 + 8], R2, R2);  tw(v[1 + 12], S1, C1);      // n = 1, 2, 3
    tw(v[2 + 4], R2, R2);  tw(v[2 + 8], 0.f, 1.f); tw(v[2 + 12], -R2, R2);    // n = 2, 4, 6
    tw(v[3 + 4], S1, C1);  tw(v[3 + 8], -R2, R2); tw(v[3 + 12], -C1, -S1);    // n = 3, 6, 9
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) dft4<INV>(v[4 * k2], v[4 * k2 + 1], v[4 * k2 + 2], v[4 * k2 + 3]);
    // X[k] now sits at v[4 (k % 4) + k / 4]
}
#define DFT16_AT(k) (4 * ((k) & 3) + ((k) >> 2))


// ---------------------------------------------------------------------------------------------
// Column pass (A: forward from the signal, C: inverse to the output), radix (16, 16).  256 threads:
// thread (col = tid & 31, q = tid >> 5) owns butterflies j = q + 8 i (i < 2) of its column, 16 rows
// each (rows j + 16 t).  LDS: one [256][32] complex buffer (64 KB), one exchange per direction.
// ---------------------------------------------------------------------------------------------
// PK: the butterflies in packed arithmetic (fftpk.h: a 16-point DFT in 80 vector instructions instead of ~160, a twiddle
// product in 2 instead of 4); the exchange and its addresses are the same.  TFX_OLS_PK=0 selects the compiler-scheduled form.
template <bool INV, int NBF, bool PK, bool LEAN = false>
__device__ __forceinline__ void col_stages16(cpx (&v)[NBF][16], cpx *lds, const cpx *tw256, int col, int q)
{
    constexpr int QS = 16 / NBF;               // butterfly j = q + QS * i
    if (PK) {
        using pk::v2f;
        const v2f Wc = {0.92387953251128675613f, 0.38268343236508977173f}, Wr = {0.70710678118654752440f, 0.70710678118654752440f};
        v2f *L = (v2f *)lds;
        const v2f *TW = (const v2f *)tw256;
#pragma unroll
        for (int i = 0; i < NBF; ++i) {
            v2f u[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) u[t] = __builtin_bit_cast(v2f, v[i][t]);
            pk::pk_dft16<INV>(u, Wc, Wr);
            const int j = q + QS * i;
#pragma unroll
            for (int k = 0; k < 16; ++k) L[(16 * j + k) * OLS_CB + col] = u[PK_DFT16_AT(k)];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NBF; ++i) {
            const int j = q + QS * i;
            v2f d[16], w[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) d[t] = L[(j + 16 * t) * OLS_CB + col];
            if (LEAN) {                                  // twiddles in four batches: 24 registers less at the peak
#pragma unroll
                for (int b = 0; b < 4; ++b) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) if (4 * b + t > 0) w[t] = TW[((4 * b + t) * j) & 255];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < 4; ++t) if (4 * b + t > 0) d[4 * b + t] = pk::pk_cmul<INV>(d[4 * b + t], w[t]);
                }
            } else {
#pragma unroll
            for (int t = 1; t < 16; ++t) w[t] = TW[(t * j) & 255];
            __builtin_amdgcn_sched_barrier(0);       // all reads are issued before the first product (asm consumers: the scheduler would sink them)
#pragma unroll
            for (int t = 1; t < 16; ++t) d[t] = pk::pk_cmul<INV>(d[t], w[t]);
            }
            pk::pk_dft16<INV>(d, Wc, Wr);          // natural-order output row j + 16 k sits at d[PK_DFT16_AT(k)]
#pragma unroll
            for (int t = 0; t < 16; ++t) v[i][t] = __builtin_bit_cast(cpx, d[t]);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < NBF; ++i) {
        dft16<INV>(v[i]);
        const int j = q + QS * i;
#pragma unroll
        for (int k = 0; k < 16; ++k) lds[(16 * j + k) * OLS_CB + col] = v[i][DFT16_AT(k)];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NBF; ++i) {
        const int j = q + QS * i;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            cpx x = lds[(j + 16 * t) * OLS_CB + col];
            if (t > 0) {
                const cpx w = tw256[(t * j) & 255];
                x = INV ? cmulc(x, w) : cmul(x, w);
            }
            v[i][t] = x;
        }
        dft16<INV>(v[i]);          // natural-order output row j + 16 k sits at v[i][DFT16_AT(k)]
    }
}

// NBF = butterflies per thread: 2 -> 256 threads (8 waves per CU at 2 workgroups), 1 -> 512 threads
// (16 waves per CU, half the registers per thread).  PROBE (development, tools/archive/ols_knobs.py):
// 1 = no FFT (load -> store), 2 = loads only, 3 = stores only.
template <int NBF, int PROBE>
__global__ void __launch_bounds__(512 / NBF, 2)
ols_col_fwd16_kernel(const float *__restrict__ x, cpx *__restrict__ T, const cpx *__restrict__ tw256g,
                     OlsGeom g, int64_t frame0)
{
    constexpr int QS = 16 / NBF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cpx *lds = (cpx *)smem;                    // [256][32]
    cpx *tw256 = lds + OLS_N1 * OLS_CB;        // [256]
    const int tid = threadIdx.x, col = tid & 31, q = tid >> 5;
    if (tid < 256) tw256[tid] = tw256g[tid];
    const int ncb = g.N2 / OLS_CB;
    const int64_t pair = blockIdx.x / ncb;
    const int cb = blockIdx.x % ncb;
    const int n2 = cb * OLS_CB + col;
    const int64_t fa = frame0 + 2 * pair, fb = fa + 1;
    const int64_t ca = fa / g.F, ia0 = (fa % g.F) * g.S - g.pad_left - row_shift(g, ca);
    const bool has_b = fb < g.nframes;
    const int64_t cb_ = has_b ? fb / g.F : 0, ib0 = has_b ? (fb % g.F) * g.S - g.pad_left - row_shift(g, cb_) : 0;
    const float *xa = x + ca * g.Tn, *xb = x + cb_ * g.Tn;
    cpx v[NBF][16];
    // interior frames (the common case) need no bounds checks
    const int64_t span = (int64_t)OLS_N1 * g.N2;
    const bool inner = ia0 >= 0 && ia0 + span <= g.Tn && has_b && ib0 >= 0 && ib0 + span <= g.Tn;
    if (PROBE == 3) {
#pragma unroll
        for (int i = 0; i < NBF; ++i)
#pragma unroll
            for (int t = 0; t < 16; ++t) v[i][t] = make_float2((float)t, (float)col);
    } else if (inner && (g.nt & 1)) {          // the signal is read once: streaming loads
#pragma unroll
        for (int i = 0; i < NBF; ++i)
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int64_t n = (int64_t)(q + QS * i + 16 * t) * g.N2 + n2;
                v[i][t] = make_float2(__builtin_nontemporal_load(xa + ia0 + n), __builtin_nontemporal_load(xb + ib0 + n));
            }
    } else if (inner) {
#pragma unroll
        for (int i = 0; i < NBF; ++i)
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int64_t n = (int64_t)(q + QS * i + 16 * t) * g.N2 + n2;
                v[i][t] = make_float2(xa[ia0 + n], xb[ib0 + n]);
            }
    } else {
#pragma unroll
        for (int i = 0; i < NBF; ++i)
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int64_t n = (int64_t)(q + QS * i + 16 * t) * g.N2 + n2;
                const int64_t ia = ia0 + n, ib = ib0 + n;
                float re = (ia >= 0 && ia < g.Tn) ? xa[ia] : 0.0f;
                float im = (has_b && ib >= 0 && ib < g.Tn) ? xb[ib] : 0.0f;
                if (g.hist) {
                    if (ia < 0 && ia >= -g.H) re = g.hist[ca * g.H + g.H + ia];
                    if (has_b && ib < 0 && ib >= -g.H) im = g.hist[cb_ * g.H + g.H + ib];
                }
                v[i][t] = make_float2(re, im);
            }
    }
    if (has_b && ca != cb_ && g.nf_pair) {
        // A pair that straddles two signal rows (the last frame of row ca, the first of row cb): frame b's non-finite samples
        // enter as zeros and row cb is flagged -- ols_straddle_fix_kernel makes that frame's output NaN, which it would have
        // been, while row ca keeps its own (rows are independent signals; inside a row a shared transform only widens the
        // non-finite stretch by a block).  Wave-uniform, at most one pair per row: the other workgroups never get here.
        bool bad = false;
#pragma unroll
        for (int i = 0; i < NBF; ++i)
#pragma unroll
            for (int t = 0; t < 16; ++t)
                if (!(__builtin_fabsf(v[i][t].y) <= 3.4028234663852886e38f)) { v[i][t].y = 0.0f; bad = true; }
        if (bad) g.nf_pair[cb_] = 1;
    }
    __syncthreads();
    if (PROBE == 0 || PROBE == 4) col_stages16<false, NBF, PROBE == 0>(v, lds, tw256, col, q);
    cpx *Tp = T + pair * ((int64_t)OLS_N1 * g.P2);
    if (PROBE == 2) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < NBF; ++i)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc += v[i][k].x + v[i][k].y;
        if (acc == 1.2345e30f) Tp[n2] = make_float2(acc, acc);
        return;
    }
#pragma unroll
    for (int i = 0; i < NBF; ++i)
#pragma unroll
        for (int k = 0; k < 16; ++k)
            Tp[(int64_t)(q + QS * i + 16 * k) * g.P2 + n2] = v[i][DFT16_AT(k)];
}

template <int NBF, int PROBE>
__global__ void __launch_bounds__(512 / NBF, 2)
ols_col_inv16_kernel(const cpx *__restrict__ T, float *__restrict__ y, const cpx *__restrict__ tw256g,
                     OlsGeom g, int64_t frame0)
{
    constexpr int QS = 16 / NBF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cpx *lds = (cpx *)smem;
    cpx *tw256 = lds + OLS_N1 * OLS_CB;
    const int tid = threadIdx.x, col = tid & 31, q = tid >> 5;
    if (tid < 256) tw256[tid] = tw256g[tid];
    const int ncb = g.N2 / OLS_CB;
    const int64_t pair = blockIdx.x / ncb;
    const int cb = blockIdx.x % ncb;
    const int n2 = cb * OLS_CB + col;
    const cpx *Tp = T + pair * ((int64_t)OLS_N1 * g.P2);
    cpx v[NBF][16];
    if (PROBE == 3) {
#pragma unroll
        for (int i = 0; i < NBF; ++i)
#pragma unroll
            for (int t = 0; t < 16; ++t) v[i][t] = make_float2((float)t, (float)col);
    } else {
#pragma unroll
        for (int i = 0; i < NBF; ++i)
#pragma unroll
            for (int t = 0; t < 16; ++t) v[i][t] = Tp[(int64_t)(q + QS * i + 16 * t) * g.P2 + n2];
    }
    __syncthreads();
    if (PROBE == 0 || PROBE == 4) col_stages16<true, NBF, PROBE == 0>(v, lds, tw256, col, q);

    const int64_t fa = frame0 + 2 * pair, fb = fa + 1;
    const int64_t ca = fa / g.F, oa0 = (fa % g.F) * g.S;
    const bool has_b = fb < g.nframes;
    const int64_t cb_ = has_b ? fb / g.F : 0, ob0 = has_b ? (fb % g.F) * g.S : 0;
    float *ya = y + ca * g.Tout, *yb = y + cb_ * g.Tout;
    if (PROBE == 2) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < NBF; ++i)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc += v[i][k].x + v[i][k].y;
        if (acc == 1.2345e30f) ya[oa0] = acc;
        return;
    }
    const int64_t sha = g.out_shift + row_shift(g, ca), shb = g.out_shift + row_shift(g, cb_);
    const bool epi = g.ep_scale | g.ep_clamp | (g.ep_stat >= 0);
    double acc_a = 0.0, acc_b = 0.0;
#pragma unroll
    for (int i = 0; i < NBF; ++i)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int64_t n = (int64_t)(q + QS * i + 16 * k) * g.N2 + n2;
            if (n < g.S) {                                   // valid part of the block
                cpx o = v[i][DFT16_AT(k)];
                const int64_t oa = oa0 + n - sha, ob = ob0 + n - shb;
                const bool wa = oa >= 0 && oa < g.Tout, wb = has_b && ob >= 0 && ob < g.Tout;
                if (epi) {                                   // Gain / clamp / statistic on the stored values
                    if (g.ep_scale) { o.x *= g.ep_gain; o.y *= g.ep_gain; }
                    if (g.ep_clamp) { o.x = clamp_unit(o.x); o.y = clamp_unit(o.y); }
                    if (g.ep_stat >= 0) {
                        if (wa) acc_a = red_comb_rt(g.ep_stat, acc_a, red_elem_rt(g.ep_stat, (double)o.x));
                        if (wb) acc_b = red_comb_rt(g.ep_stat, acc_b, red_elem_rt(g.ep_stat, (double)o.y));
                    }
                }
                if (g.nt & 2) {            // the output is written once and not read back here: streaming stores
                    if (wa) __builtin_nontemporal_store(o.x, ya + oa);
                    if (wb) __builtin_nontemporal_store(o.y, yb + ob);
                } else {
                    if (wa) ya[oa] = o.x;
                    if (wb) yb[ob] = o.y;
                }
            }
        }
    if (g.ep_stat >= 0) {                  // one partial per (frame, column block), threads combined in a fixed order
        double *red = (double *)smem;                        // the FFT buffer is free again
        __syncthreads();
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            acc_a = red_comb_rt(g.ep_stat, acc_a, __shfl_xor(acc_a, off));
            acc_b = red_comb_rt(g.ep_stat, acc_b, __shfl_xor(acc_b, off));
        }
        const int nw = (int)(blockDim.x >> 6), w = tid >> 6;
        if ((tid & 63) == 0) { red[w] = acc_a; red[nw + w] = acc_b; }
        __syncthreads();
        if (tid == 0) {
            double ra = red[0], rb = red[nw];
            for (int u = 1; u < nw; ++u) { ra = red_comb_rt(g.ep_stat, ra, red[u]); rb = red_comb_rt(g.ep_stat, rb, red[nw + u]); }
            g.ep_partial[fa * ncb + cb] = ra;
            if (has_b) g.ep_partial[fb * ncb + cb] = rb;
        }
    }
}

// see the straddling-pair branch of ols_col_fwd16_kernel: frame 0 of a flagged row becomes NaN (grid: C x chunks -- rows on x, the
// dimension without a 65 535 limit)
__global__ void __launch_bounds__(256) ols_straddle_fix_kernel(float *__restrict__ y, OlsGeom g)
{
    const int64_t c = blockIdx.x;
    if (!g.nf_pair[c]) return;
    if (blockIdx.y == 0 && threadIdx.x == 0 && g.ep_stat >= 0) g.ep_partial[(c * g.F) * (g.N2 / OLS_CB)] = __builtin_nan("");
    const int64_t hi = min(g.Tout, g.S - g.out_shift - row_shift(g, c));
    for (int64_t t = (int64_t)blockIdx.y * 256 + threadIdx.x; t < hi; t += (int64_t)gridDim.y * 256) y[c * g.Tout + t] = __builtin_nanf("");
}

// ---------------------------------------------------------------------------------------------
// Pass A with the SOS cascade in front of it (N2 = 4096 or 8192): `iir-cascade | FIR...` in the reference's own arithmetic --
// float64 DF1 recursion (src/torchfx/_csrc/cpu/iir_cpu.cpp:132-147), one rounding to float32 (filter/iir.py: the downcast
// of _sos_cascade_forward), float32 overlap-save (filter/_fftconv.py:123-140) -- without the recursion's own 8 B/sample pass.
//
// A frame is 256 rows of N2 = 4096 (N = 2^20) or 8192 (N = 2^21) consecutive samples and the column transform wants 32
// adjacent columns of ALL rows at once, so no workgroup ever holds a long run of consecutive samples -- but it holds 512
// SHORT runs that each continue where the previous column block stopped.  (The order the round-4 review proposed -- the
// contiguous row transform first -- is not a factorisation of the DFT: Cooley-Tukey transforms the strided index first,
// profiles/r05_experiments.txt section 1.)  One workgroup therefore owns a frame PAIR and walks its N2 / 32 column blocks in time
// order; thread (frame, row) is the recursion of that row: it carries the 2K+2 float64 state values of its row in
// registers from block to block and runs the plain sequential DF1 recursion over its 32 samples -- no scan, no matrices,
// 5 operations per sample and section (4 in the unit-b0 form).  A row starts `warm` samples early from zero state (the
// warm-up analysis of sos.hip at max|A^W| < 2^-40, fftconv.hip: the state at the row's first sample is the true one to the
// round-off a float64 recursion gathers over a row anyway); those
// warm-up blocks are read and filtered but not transformed.  Per block: coalesced 16-byte loads of the 512 lines
// (prefetched one block ahead) -> LDS stage [512][36] -> each thread takes its line, filters it and puts the rounded
// float32 samples back IN PLACE -> the stage is the column transform's input z = a + i b -> radix-16 x 16 transform through
// the same LDS bytes -> workspace.  Samples outside [0, T) enter the recursion as zeros and leave it as zeros (the
// reference filters T samples and the FIR pads afterwards).
// ---------------------------------------------------------------------------------------------
constexpr int SOSF_MAXK = 8;
constexpr int SOSF_K4W = 4;                      // up to this many sections the kernel fits 128 registers: two workgroups per CU
constexpr int SOSF_CH = 16;                      // samples per unrolled stretch of the recursion (bounds the live ranges)
constexpr int SOSF_LS = 36;                      // floats per line of the stage: conflict-free ds_read_b128 / ds_write_b128 per lane
constexpr size_t OLS_SHM_SOSF = (size_t)512 * SOSF_LS * sizeof(float) + 256 * sizeof(cpx);
struct SosFuse {                                 // by value in the kernel arguments: the coefficients are scalar operands
    double co[SOSF_MAXK][5];                     // b0, b1, b2, -a1, -a2 of each section; unit-b0 form: b0_0 ... b0_s, b1 / b0, b2 / b0, -a1, -a2
    double *sections;                            // optional [K, C, T] float64: every section's output (parity tests), or null
    int warm_blocks;                             // warm-up of a row in 32-sample blocks
    int prio;                                    // the transform / memory phases issue ahead of the recursion (TFX_OLS_SOS_PRIO)
};

template <int KS, bool TAPS, bool UNIT>
__global__ void __launch_bounds__(512, ((KS <= SOSF_K4W && !TAPS) ? 4 : 2))
ols_col_fwd16_sos_kernel(const float *__restrict__ x, cpx *__restrict__ T, const cpx *__restrict__ tw256g,
                         OlsGeom g, int64_t frame0, SosFuse sf)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *S = (float *)smem;                          // [512][SOSF_LS]: line (frame, row)
    cpx *lds = (cpx *)smem;                            // the transform's [256][32] exchange lives in the same bytes
    cpx *tw256 = (cpx *)(smem + (size_t)512 * SOSF_LS * sizeof(float));
    const int tid = threadIdx.x;
    if (tid < 256) tw256[tid] = tw256g[tid];
    const int64_t pair = blockIdx.x;
    const int64_t fa = frame0 + 2 * pair, fb = fa + 1;
    const int64_t ca = fa / g.F, ia0 = (fa % g.F) * g.S - g.pad_left - row_shift(g, ca);
    const bool has_b = fb < g.nframes;
    const int64_t cb_ = has_b ? fb / g.F : 0, ib0 = has_b ? (fb % g.F) * g.S - g.pad_left - row_shift(g, cb_) : 0;
    const float *xa = x + ca * g.Tn, *xb = x + cb_ * g.Tn;
    const int nblk = g.N2 / OLS_CB;
    // Wave-uniform clock of the walk: `ta` / `tb` = time (sample index in its row) of row 0's first sample of the block being
    // FETCHED; a line is row r of a frame, r * N2 samples later.  Frames start on 128-byte lines of MEMORY (row_shift), so a
    // line of the stage is one line of the signal; where the row begins or ends inside a line (T % 32 != 0) the loader takes
    // the samples one by one, everywhere else validity is one comparison of the 16-byte part's offset with two scalars.
    int64_t ta = ia0 - (int64_t)OLS_CB * sf.warm_blocks, tb = ib0 - (int64_t)OLS_CB * sf.warm_blocks;
    auto bound = [](int64_t v) { return (int)(v < -(int64_t)0x40000000 ? -(int64_t)0x40000000 : (v > (int64_t)0x40000000 ? (int64_t)0x40000000 : v)); };

    // recursion side: this thread is line `tid` = (frame tid >> 8, row tid & 255)
    const bool mine_b = tid >= 256;
    const int rel_l = (tid & 255) * g.N2;
    // section taps: a sample lies in the windows of two consecutive frames (they overlap by K - 1 samples); the frame whose LAST
    // S window samples hold it stores it (frame 0 of a row: its whole window) -- one writer per address
    const int own_lo = TAPS ? (((mine_b ? fb : fa) % g.F) == 0 ? 0 : OLS_N1 * g.N2 - (int)g.S) : 0;
    double h1[KS + 1], h2[KS + 1];             // h[0]: input history; h[s + 1]: output history of section s (iir_cpu.cpp:125-130)
#pragma unroll
    for (int s = 0; s <= KS; ++s) { h1[s] = 0.0; h2[s] = 0.0; }

    // loader side: line (tid >> 3) + 64 i, 16-byte part tid & 7;  i < 4: frame a, i >= 4: frame b
    const int lrow = tid >> 3, lpart = tid & 7;
    const int rel_f = lrow * g.N2 + 4 * lpart;
    const int rstep = 64 * g.N2;
    float4 P[8];
    auto fetch = [&]() {
        const float *pa = xa + ta, *pb = xb + tb;            // may point outside the row: dereferenced only where valid
        const int lo_a = bound(-ta), hi_a = bound(g.Tn - ta), lo_b = bound(-tb), hi_b = has_b ? bound(g.Tn - tb) : lo_b;
        int rel = rel_f;
        asm volatile("" : "+v"(rel));                        // offsets are recomputed per block, not kept in sixteen registers
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool isb = i >= 4;
            const int r = rel + (i & 3) * rstep;
            const int lo = isb ? lo_b : lo_a, hi = isb ? hi_b : hi_a;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r >= lo && r + 4 <= hi) {
                const float *src = (isb ? pb : pa) + r;
                v = (g.nt & 1) ? ldg16_stream<float4>(src) : *(const float4 *)src;
            } else if (r + 4 > lo && r < hi) {               // the row begins or ends inside this 16-byte part
                const float *src = (isb ? pb : pa) + r;
                if (r >= lo) v.x = src[0];
                if (r + 1 >= lo && r + 1 < hi) v.y = src[1];
                if (r + 2 >= lo && r + 2 < hi) v.z = src[2];
                if (r + 3 < hi) v.w = src[3];
            }
            P[i] = v;
        }
        ta += OLS_CB; tb += OLS_CB;
    };
    fetch();
    cpx *Tp = T + pair * ((int64_t)OLS_N1 * g.P2);
    const int col = tid & 31, q = tid >> 5;
    for (int blk = -sf.warm_blocks; blk < nblk; ++blk) {
#pragma unroll
        for (int i = 0; i < 8; ++i) *(float4 *)&S[(lrow + 64 * i) * SOSF_LS + 4 * lpart] = P[i];
        // the block being filtered is the one fetched last: its clock is one block behind ta / tb
        int64_t tcur = (mine_b ? tb : ta) - OLS_CB;
        {   // wave-uniform (a wavefront lies in one frame): keep it in scalar registers
            const unsigned lo32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uint64_t)tcur);
            const unsigned hi32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((uint64_t)tcur >> 32));
            tcur = (int64_t)(((uint64_t)hi32 << 32) | lo32);
        }
        // samples [first, last) of this thread's line lie inside the row.  In front of the row the recursion sees zeros from
        // zero state and returns zeros by itself; behind its end the filter's tail is cut (the reference filters T samples and
        // the FIR pads afterwards)
        const int last = (!mine_b || has_b) ? min(max(bound(g.Tn - tcur) - rel_l, 0), OLS_CB) : 0;
        int first = 0;
        if (TAPS) first = max(min(max(bound(-tcur) - rel_l, 0), OLS_CB), min(max(own_lo - (rel_l + OLS_CB * blk), 0), OLS_CB));
        if (blk + 1 < nblk) fetch();                       // in flight while this block is filtered and transformed
        __syncthreads();
        // The recursion is a long stream of independent float64 operations, the transform a short chain of LDS round trips and
        // barriers: with two workgroups per CU the arbiter (oldest first) lets one workgroup's recursion starve the other's
        // transform.  sf.prio: the transform and the memory phases issue ahead of the recursion.
        if (sf.prio) __builtin_amdgcn_s_setprio(0);
#pragma unroll 1
        for (int j = 0; j < OLS_CB / SOSF_CH; ++j) {
            float u[SOSF_CH];
#pragma unroll
            for (int i = 0; i < SOSF_CH / 4; ++i) {
                const float4 w4 = *(const float4 *)&S[tid * SOSF_LS + SOSF_CH * j + 4 * i];
                u[4 * i] = w4.x; u[4 * i + 1] = w4.y; u[4 * i + 2] = w4.z; u[4 * i + 3] = w4.w;
            }
            const int lastj = last - SOSF_CH * j, firstj = first - SOSF_CH * j;
#pragma unroll
            for (int n = 0; n < SOSF_CH; ++n) {
                double v = (double)u[n];
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    // UNIT: every b0 is pulled out of its section (the recursion is scale-invariant): 4 instead of 5 operations,
                    // the product of all b0 comes back in one multiply where the sample is rounded
                    double yn = UNIT ? __builtin_fma(sf.co[s][1], h1[s], v) : __builtin_fma(sf.co[s][1], h1[s], sf.co[s][0] * v);
                    yn = __builtin_fma(sf.co[s][2], h2[s], yn);
                    yn = __builtin_fma(sf.co[s][3], h1[s + 1], yn);
                    yn = __builtin_fma(sf.co[s][4], h2[s + 1], yn);
                    h2[s] = h1[s]; h1[s] = v;
                    v = yn;
                    if (TAPS) {
                        if (blk >= 0 && n >= firstj && n < lastj)
                            sf.sections[((int64_t)s * (g.nframes / g.F) + (mine_b ? cb_ : ca)) * g.Tn + tcur + rel_l + SOSF_CH * j + n] =
                                UNIT ? yn * sf.co[s][0] : yn;
                    }
                }
                h2[KS] = h1[KS]; h1[KS] = v;
                if (UNIT) v *= sf.co[KS - 1][0];
                u[n] = n < lastj ? (float)v : 0.0f;
            }
            // Two real frames ride one complex transform: a non-finite sample of one would come out in BOTH.  A state that
            // has gone non-finite stays so (flag + fix-up below): from here on this line enters the transform as zeros -- the
            // frame's own output is replaced by NaN afterwards, its partner's stays what it is.
            // (the stretch is overwritten in the stage, not in the registers: sixteen samples kept live to the end of the stretch
            // cost three spilled registers at the 128 this kernel has)
            if (blk >= 0) {
#pragma unroll
                for (int i = 0; i < SOSF_CH / 4; ++i)
                    *(float4 *)&S[tid * SOSF_LS + SOSF_CH * j + 4 * i] = make_float4(u[4 * i], u[4 * i + 1], u[4 * i + 2], u[4 * i + 3]);
                if (!(__builtin_fabs(h1[KS]) <= 1.7976931348623157e308)) {
#pragma unroll
                    for (int i = 0; i < SOSF_CH / 4; ++i)
                        *(float4 *)&S[tid * SOSF_LS + SOSF_CH * j + 4 * i] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
        if (sf.prio) __builtin_amdgcn_s_setprio(3);
        if (blk < 0) { __syncthreads(); continue; }           // warm-up block: every line was read, the stage may be refilled
        __syncthreads();
        cpx v[1][16];
#pragma unroll
        for (int t = 0; t < 16; ++t) v[0][t] = make_float2(S[(q + 16 * t) * SOSF_LS + col], S[(256 + q + 16 * t) * SOSF_LS + col]);
        __syncthreads();                                      // the exchange overwrites the stage
        col_stages16<false, 1, true, true>(v, lds, tw256, col, q);
        char *Tb = (char *)(Tp + blk * OLS_CB);               // wave-uniform base, 32-bit lane offsets (a pair's workspace is 8 MB)
        unsigned off = (unsigned)(q * g.P2 + col) * (unsigned)sizeof(cpx);
        asm volatile("" : "+v"(off));
        const unsigned ostep = 16u * (unsigned)g.P2 * (unsigned)sizeof(cpx);
#pragma unroll
        for (int k = 0; k < 16; ++k) { *(cpx *)(Tb + off) = v[0][DFT16_AT(k)]; off += ostep; }
        __syncthreads();                                      // exchange read: the stage may be refilled
    }
    // Non-finite values never leave a recursion (iir_cpu.cpp:132-147: once in the state, every later output of the row carries
    // them) -- but a row of this walk starts again from zero state, so a bad sample would "heal" one row later.  The state a
    // thread ends with tells whether its row (warm-up included) met one: one flag per frame, and olsnative_forward's fix-up pass
    // turns everything behind the first flagged frame of a signal row into NaN like the staged pair of launches does.
    {
        int *flag = (int *)smem;                              // the stage is free (last barrier above)
        if (tid < 2) flag[tid] = 0;
        __syncthreads();
        const double e1 = h1[KS], e2 = h2[KS];
        if (!(__builtin_fabs(e1) <= 1.7976931348623157e308 && __builtin_fabs(e2) <= 1.7976931348623157e308)) flag[mine_b ? 1 : 0] = 1;
        __syncthreads();
        if (tid == 0) g.nf_flag[fa] = flag[0];
        if (tid == 1 && has_b) g.nf_flag[fb] = flag[1];
    }
}

// Fix-up behind pass C of the cascade-in-pass-A pipeline (see the flags above).  Grid (C, chunks, 1 + sections): every
// workgroup first reads its signal row's F flags and leaves when none is set -- the common case, a few microseconds for the whole
// launch.  z = 0: the output row becomes NaN from the first sample of the first flagged frame: that frame's window holds the bad
// sample and every later frame's window lies behind it -- what the staged pair of launches returns (the cascade's output stays
// non-finite to the end of the row, and every block of the convolution that reaches it is non-finite as a whole).  The
// flagged frames entered the transform as zeros from the bad sample on, so the frame that shares a transform with one (even a
// frame of the neighbouring signal row) keeps its own finite output.  z = s + 1 (section taps, parity tests): section s is exact up to its first non-finite sample, which
// lies in the flagged frame's window or its warm-up; everything behind that sample becomes NaN (iir_cpu.cpp:132-147).
__global__ void __launch_bounds__(256)
ols_sos_nonfinite_fix_kernel(float *__restrict__ y, double *__restrict__ sections, OlsGeom g, int warm_blocks)
{
    __shared__ int s_first;
    __shared__ long long s_n;
    const int tid = threadIdx.x;
    const int64_t c = blockIdx.x;
    if (tid == 0) { s_first = 0x7fffffff; s_n = 0x7fffffffffffffffll; }
    __syncthreads();
    for (int64_t f = tid; f < g.F; f += 256)
        if (g.nf_flag[c * g.F + f]) atomicMin(&s_first, (int)f);
    __syncthreads();
    const int64_t first = s_first;
    if (first == 0x7fffffff) return;
    const int sh = row_shift(g, c);
    const float nanf_ = __builtin_nanf("");
    if (blockIdx.z == 0) {
        // the statistic of an epilogue saw the flagged frames' stand-in samples: a NaN partial wins both reductions (epilogue.h)
        if (blockIdx.y == 0 && tid == 0 && g.ep_stat >= 0) g.ep_partial[(c * g.F + first) * (g.N2 / OLS_CB)] = __builtin_nan("");
        const int64_t t0 = max((int64_t)0, first * g.S - g.out_shift - sh);
        const int64_t len = g.Tout - t0, per = (len + gridDim.y - 1) / gridDim.y;
        const int64_t lo = t0 + per * blockIdx.y, hi = min(g.Tout, lo + per);
        for (int64_t t = lo + tid; t < hi; t += 256) y[c * g.Tout + t] = nanf_;
        return;
    }
    if (!sections || blockIdx.y != 0) return;
    double *row = sections + ((int64_t)(blockIdx.z - 1) * (g.nframes / g.F) + c) * g.Tn;
    const int64_t start = max((int64_t)0, first * g.S - g.pad_left - sh - (int64_t)OLS_CB * warm_blocks);
    for (int64_t t0 = start; t0 < g.Tn; t0 += 4096) {         // first non-finite sample of this section, 4096 samples at a time
        for (int64_t t = t0 + tid; t < min(g.Tn, t0 + 4096); t += 256)
            if (!(__builtin_fabs(row[t]) <= 1.7976931348623157e308)) { atomicMin(&s_n, (long long)t); break; }
        __syncthreads();
        if (s_n != 0x7fffffffffffffffll) break;
    }
    const int64_t n = s_n;
    if (n == 0x7fffffffffffffffll) return;
    const double nan_ = __builtin_nan("");
    for (int64_t t = n + 1 + tid; t < g.Tn; t += 256) row[t] = nan_;
}

// ---------------------------------------------------------------------------------------------
// Row pass B: one wavefront per row of N2 = 4^L2 points; lane owns butterflies j = lane + 64 i.
// ---------------------------------------------------------------------------------------------
template <int L2, bool INV>
__device__ __forceinline__ void row_stages(cpx (&v)[(1 << (2 * L2)) / 256][4], cpx *lds, const cpx *twr, int lane)
{
    constexpr int N2 = 1 << (2 * L2), Q = N2 / 4, NB = Q / 64;     // NB butterflies per lane
#pragma unroll
    for (int i = 0; i < NB; ++i) dft4<INV>(v[i][0], v[i][1], v[i][2], v[i][3]);
#pragma unroll
    for (int s = 1; s < L2; ++s) {
        const int Ns_prev = 1 << (2 * (s - 1));
        const int Ns = Ns_prev * 4;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int j = lane + 64 * i;
            const int j0 = (j / Ns_prev) * (4 * Ns_prev) + (j % Ns_prev);
#pragma unroll
            for (int r = 0; r < 4; ++r) lds[j0 + r * Ns_prev] = v[i][r];
        }
        wave_sync2();
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int j = lane + 64 * i;
            const int k = j % Ns;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                cpx x = lds[j + Q * r];
                if (r > 0) {
                    const cpx w = twr[(r * k * (Q / Ns)) & (N2 - 1)];
                    x = INV ? cmulc(x, w) : cmul(x, w);
                }
                v[i][r] = x;
            }
            dft4<INV>(v[i][0], v[i][1], v[i][2], v[i][3]);
        }
        wave_sync2();
    }
}

template <int L2>
__global__ void __launch_bounds__(256)
ols_row_kernel(cpx *__restrict__ T, const cpx *__restrict__ Hp, const cpx *__restrict__ twrg,
               const cpx *__restrict__ tlo, const cpx *__restrict__ thi, int64_t nrows, int P2)
{
    constexpr int N2 = 1 << (2 * L2), Q = N2 / 4, NB = Q / 64;
    constexpr int64_t N = (int64_t)OLS_N1 * N2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cpx *twr = (cpx *)smem;                      // [N2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    cpx *lds = twr + N2 + wave * N2;             // per-wave [N2]
    for (int i = tid; i < N2; i += 256) twr[i] = twrg[i];
    __syncthreads();
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= nrows) return;
    const int k1 = (int)(row % OLS_N1);
    cpx *base = T + row * P2;
    const cpx *hrow = Hp + (int64_t)k1 * N2;

    cpx v[NB][4], w[NB][4];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n2 = lane + 64 * i + Q * r;
            const unsigned m = (unsigned)(((int64_t)k1 * n2) & (N - 1));
            w[i][r] = cmul(tlo[m & 511], thi[m >> 9]);        // W_N^(k1 n2)
            v[i][r] = cmul(base[n2], w[i][r]);
        }
    }
    row_stages<L2, false>(v, lds, twr, lane);
    // now v[i][r] = X[k2 = lane + 64 i + Q r]; multiply by the permuted, conjugated, scaled spectrum
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[i][r] = cmul(v[i][r], hrow[lane + 64 * i + Q * r]);
    row_stages<L2, true>(v, lds, twr, lane);
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) base[lane + 64 * i + Q * r] = cmulc(v[i][r], w[i][r]);
}

// ---------------------------------------------------------------------------------------------
// Row pass B, N2 = 256 (N = 2^16: 2048 < K <= 16384), packed arithmetic (fftpk.h): SIXTEEN LANES per row, four rows per
// wavefront; lane i of a row holds n2 = i + 16 t, so both radix-16 stages of a direction run in registers and a direction
// needs ONE exchange -- a 16 x 16 transposition inside the 16-lane group through a wave-local LDS tile (stride 17: conflict-free
// both ways), no workgroup barrier (the radix-4 kernel above: one row per wavefront, four exchanges per direction).
//   forward   lane i: DFT over t -> A[k0], * W256^(i k0), transpose -> lane k0 holds A_i[k0] over i, DFT over i -> X[k0 + 16 k1]
//   inverse   lane k0: DFT over k1 -> m0, * conj W256^(k0 m0), transpose -> lane m0, DFT over k0 -> x[m0 + 16 m1]
// The lane's fifteen W256^(i k) are registers for the whole launch (workgroups walk the rows with a grid stride); the four-step
// factors W_N^(k1 (i + 16 t)) = W_N^(k1 i) (W_N^(16 k1))^t come from two table look-ups and a depth-4 product tree.
__global__ void __launch_bounds__(256, 3)
ols_row256pk_kernel(cpx *__restrict__ T, const cpx *__restrict__ Hp, const cpx *__restrict__ tw256g,
                    const cpx *__restrict__ tlo, const cpx *__restrict__ thi, int64_t nrows, int P2)
{
    using pk::v2f;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, r = lane >> 4;
    v2f *L = (v2f *)smem + (wave * 4 + r) * 272;               // this row's 16 x 17 tile
    const v2f Wc = {0.92387953251128675613f, 0.38268343236508977173f}, Wr = {0.70710678118654752440f, 0.70710678118654752440f};
    v2f tw[16];
#pragma unroll
    for (int k = 1; k < 16; ++k) tw[k] = ((const v2f *)tw256g)[(i * k) & 255];
    for (int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * 4; row0 < nrows; row0 += (int64_t)gridDim.x * 16) {
        const int64_t row = row0 + r;
        const bool live = row < nrows;
        const int k1 = (int)((live ? row : 0) % OLS_N1);
        v2f *base = (v2f *)(T + (live ? row : 0) * P2);
        const v2f *hrow = (const v2f *)(Hp + (int64_t)k1 * 256);
        v2f w[16];
        {   // w[t] = W_N^(k1 i) * s^t, s = W_N^(16 k1), N = 65536
            const unsigned ma = (unsigned)(k1 * i), mb = (unsigned)(16 * k1);
            const v2f wl = pk::pk_cmul<false>(((const v2f *)tlo)[ma & 511], ((const v2f *)thi)[ma >> 9]);
            v2f u[16];
            u[1] = pk::pk_cmul<false>(((const v2f *)tlo)[mb & 511], ((const v2f *)thi)[mb >> 9]);
            u[2] = pk::pk_cmul<false>(u[1], u[1]);
            u[3] = pk::pk_cmul<false>(u[2], u[1]);
            u[4] = pk::pk_cmul<false>(u[2], u[2]);
            u[5] = pk::pk_cmul<false>(u[4], u[1]);
            u[6] = pk::pk_cmul<false>(u[4], u[2]);
            u[7] = pk::pk_cmul<false>(u[4], u[3]);
            u[8] = pk::pk_cmul<false>(u[4], u[4]);
#pragma unroll
            for (int t = 9; t < 16; ++t) u[t] = pk::pk_cmul<false>(u[8], u[t - 8]);
            w[0] = wl;
#pragma unroll
            for (int t = 1; t < 16; ++t) w[t] = pk::pk_cmul<false>(wl, u[t]);
        }
        v2f v[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = live ? base[i + 16 * t] : v2f{0.f, 0.f};
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = pk::pk_cmul<false>(v[t], w[t]);
        // ---- forward
        pk::pk_dft16<false>(v, Wc, Wr);
        L[17 * i] = v[PK_DFT16_AT(0)];
#pragma unroll
        for (int k = 1; k < 16; ++k) L[17 * i + k] = pk::pk_cmul<false>(v[PK_DFT16_AT(k)], tw[k]);
        wave_sync2();
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = L[17 * t + i];
        wave_sync2();
        pk::pk_dft16<false>(v, Wc, Wr);                        // X[i + 16 k] at v[PK_DFT16_AT(k)]
        v2f h[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) h[k] = hrow[i + 16 * k];
        __builtin_amdgcn_sched_barrier(0);
        v2f z[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) z[k] = pk::pk_cmul<false>(v[PK_DFT16_AT(k)], h[k]);
        // ---- inverse
        pk::pk_dft16<true>(z, Wc, Wr);
        L[17 * i] = z[PK_DFT16_AT(0)];
#pragma unroll
        for (int k = 1; k < 16; ++k) L[17 * i + k] = pk::pk_cmul<true>(z[PK_DFT16_AT(k)], tw[k]);
        wave_sync2();
#pragma unroll
        for (int t = 0; t < 16; ++t) z[t] = L[17 * t + i];
        wave_sync2();
        pk::pk_dft16<true>(z, Wc, Wr);                         // x[i + 16 m] at z[PK_DFT16_AT(m)]
        if (live) {
#pragma unroll
            for (int t = 0; t < 16; ++t) base[i + 16 * t] = pk::pk_cmul<true>(z[PK_DFT16_AT(t)], w[t]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Row pass B, N2 = 1024, radix (16, 16, 4) Stockham: the lane's 16 elements  n2 = lane + 64 t  are
// exactly the inputs of one radix-16 butterfly, so two of the three stages run in registers and a
// direction needs only two LDS exchanges (the radix-4 version above needs four).  LDS positions
// are padded by one element per 16 so the stride-16 writes of the first stage are conflict-free.
// ---------------------------------------------------------------------------------------------
// in: v[t] = element at position lane + 64 t (natural order).  out: same arrangement, transformed.
template <bool INV>
__device__ __forceinline__ void row_fft1024(cpx (&v)[16], cpx *lds, const cpx *twr, int lane)
{
    // stage A: radix 16, Ns = 1
    dft16<INV>(v);
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[pad16(16 * lane + k)] = v[DFT16_AT(k)];
    wave_sync2();
    // stage B: radix 16, Ns = 16: inputs lane + 64 t, twiddle W256^(t k), k = lane % 16
    const int kb = lane & 15;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        cpx x = lds[pad16(lane + 64 * t)];
        if (t > 0) {
            const cpx w = twr[(4 * t * kb) & 1023];
            x = INV ? cmulc(x, w) : cmul(x, w);
        }
        v[t] = x;
        if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);    // bound the number of loads in flight
    }
    wave_sync2();
    dft16<INV>(v);
    const int j0 = (lane >> 4) * 256 + kb;
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[pad16(j0 + 16 * k)] = v[DFT16_AT(k)];
    wave_sync2();
    // stage C: radix 4, Ns = 256: butterflies j = lane + 64 i, inputs j + 256 r, twiddle W1024^(r j)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = lane + 64 * i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            cpx x = lds[pad16(j + 256 * r)];
            if (r > 0) {
                const cpx w = twr[(r * j) & 1023];
                x = INV ? cmulc(x, w) : cmul(x, w);
            }
            v[i + 4 * r] = x;
        }
        dft4<INV>(v[i], v[i + 4], v[i + 8], v[i + 12]);     // outputs j + 256 r  ->  t = i + 4 r
        __builtin_amdgcn_sched_barrier(0);
    }
    wave_sync2();
}

// the same wavefront transform in packed arithmetic (fftpk.h): in: v[t] = element lane + 64 t, out: the same arrangement
template <bool INV>
__device__ __forceinline__ void row_fft1024_pk(pk::v2f (&v)[16], pk::v2f *lds, const pk::v2f *twr, int lane, pk::v2f Wc, pk::v2f Wr)
{
    using pk::v2f;
    pk::pk_dft16<INV>(v, Wc, Wr);                              // stage A: radix 16, Ns = 1
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[pad16(16 * lane + k)] = v[PK_DFT16_AT(k)];
    wave_sync2();
    const int kb = lane & 15;
    {                                                          // stage B: inputs lane + 64 t, twiddle W256^(t k) = W1024^(4 t k)
        v2f d[16], w[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) d[t] = lds[pad16(lane + 64 * t)];
#pragma unroll
        for (int t = 1; t < 16; ++t) w[t] = twr[(4 * t * kb) & 1023];
        __builtin_amdgcn_sched_barrier(0);
        v[0] = d[0];
#pragma unroll
        for (int t = 1; t < 16; ++t) v[t] = pk::pk_cmul<INV>(d[t], w[t]);
    }
    wave_sync2();
    pk::pk_dft16<INV>(v, Wc, Wr);
    const int j0 = (lane >> 4) * 256 + kb;
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[pad16(j0 + 16 * k)] = v[PK_DFT16_AT(k)];
    wave_sync2();
    {                                                          // stage C: radix 4, butterflies j = lane + 64 i, inputs j + 256 r
        v2f d[16], w[16];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                d[i + 4 * r] = lds[pad16(lane + 64 * i + 256 * r)];
                if (r > 0) w[i + 4 * r] = twr[(r * (lane + 64 * i)) & 1023];
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int r = 1; r < 4; ++r) d[i + 4 * r] = pk::pk_cmul<INV>(d[i + 4 * r], w[i + 4 * r]);
            pk::pk_dft4<INV, false>(d[i], d[i + 4], d[i + 8], d[i + 12]);      // outputs j + 256 r -> t = i + 4 r
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = d[t];
    }
    wave_sync2();
}

template <bool PK>
__global__ void __launch_bounds__(256, 3)
ols_row1024_kernel(cpx *__restrict__ T, const cpx *__restrict__ Hp, const cpx *__restrict__ twrg,
                   const cpx *__restrict__ tlo, const cpx *__restrict__ thi, const cpx *__restrict__ tu,
                   int64_t nrows, int64_t Nmask, int P2)
{
    constexpr int N2 = 1024, LDSROW = N2 + N2 / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cpx *twr = (cpx *)smem;                      // [1024]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform, and the compiler knows it
    cpx *lds = twr + N2 + wave * LDSROW;
    for (int i = tid; i < N2; i += 256) twr[i] = twrg[i];
    __syncthreads();
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= nrows) return;
    const int k1 = (int)(row % OLS_N1);
    cpx *base = T + row * P2;
    const cpx *hrow = Hp + (int64_t)k1 * N2;
    typedef const float __attribute__((address_space(4))) *cfp;
    const cfp tuc = (cfp)(uintptr_t)tu;          // uniform per row: W_N^(64 k1 t) -> scalar loads

    // W_N^(k1 n2), n2 = lane + 64 t  =  W_N^(k1 lane) * W_N^(64 k1 t)
    const unsigned ml = (unsigned)(k1 * lane);
    const cpx wl = cmul(tlo[ml & 511], thi[ml >> 9]);
    if (PK) {
        using pk::v2f;
        const v2f Wc = {0.92387953251128675613f, 0.38268343236508977173f}, Wr = {0.70710678118654752440f, 0.70710678118654752440f};
        v2f u[16], h[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) h[t] = ((const v2f *)base)[lane + 64 * t];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const unsigned iu = 2u * ((unsigned)(k1 * t) & (unsigned)(Nmask >> 6));
            const cpx w = cmul(wl, make_float2(tuc[iu], tuc[iu + 1]));
            u[t] = pk::pk_cmul<false>(h[t], __builtin_bit_cast(v2f, w));
        }
        row_fft1024_pk<false>(u, (v2f *)lds, (const v2f *)twr, lane, Wc, Wr);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 16; ++t) h[t] = ((const v2f *)hrow)[lane + 64 * t];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 16; ++t) u[t] = pk::pk_cmul<false>(u[t], h[t]);
        __builtin_amdgcn_sched_barrier(0);
        row_fft1024_pk<true>(u, (v2f *)lds, (const v2f *)twr, lane, Wc, Wr);
        float wlx = wl.x, wly = wl.y;
        asm volatile("" : "+v"(wlx), "+v"(wly));
        const cpx wl2 = make_float2(wlx, wly);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const unsigned iu = 2u * ((unsigned)(k1 * t) & (unsigned)(Nmask >> 6));
            const cpx w = cmul(wl2, make_float2(tuc[iu], tuc[iu + 1]));
            ((v2f *)base)[lane + 64 * t] = pk::pk_cmul<true>(u[t], __builtin_bit_cast(v2f, w));
        }
        return;
    }
    cpx v[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const unsigned iu = 2u * ((unsigned)(k1 * t) & (unsigned)(Nmask >> 6));
        const cpx ut = make_float2(tuc[iu], tuc[iu + 1]);
        v[t] = cmul(base[lane + 64 * t], cmul(wl, ut));
    }
    __builtin_amdgcn_sched_barrier(0);
    row_fft1024<false>(v, lds, twr, lane);
    __builtin_amdgcn_sched_barrier(0);      // keep the spectrum loads from being hoisted over the FFT
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = cmul(v[t], hrow[lane + 64 * t]);
    __builtin_amdgcn_sched_barrier(0);
    row_fft1024<true>(v, lds, twr, lane);
    __builtin_amdgcn_sched_barrier(0);
    // recompute the row twiddles instead of keeping 16 of them live across both FFTs: the empty
    // asm hides wl from common-subexpression elimination
    float wlx = wl.x, wly = wl.y;
    asm volatile("" : "+v"(wlx), "+v"(wly));
    const cpx wl2 = make_float2(wlx, wly);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const unsigned iu = 2u * ((unsigned)(k1 * t) & (unsigned)(Nmask >> 6));
        const cpx ut = make_float2(tuc[iu], tuc[iu + 1]);
        base[lane + 64 * t] = cmulc(v[t], cmul(wl2, ut));
    }
}

// ---------------------------------------------------------------------------------------------
// Row pass B, N2 = 4096 (N = 2^20: 93.6 % of every block is valid output at K = 65536 instead of
// 74.6 % at N = 2^18).  One workgroup per row, thread j owns elements n2 = j + 256 t: radix
// (16, 16, 16) Stockham, three register stages and two LDS exchanges per direction.
// ---------------------------------------------------------------------------------------------
// Exchange layout XCH (physical LDS position of logical p is p + p/16 in both):
//   0  "write strided, read contiguous" (textbook Stockham): stage-1 butterfly j scatters output k to 16 j + k,
//      stage 2 gathers j + 256 t.  The ds_write_b64 groups (16 contiguous lanes, banks mod 32 dwords) are
//      conflict-free, but a ds_read_b64 group is 32 lanes and the 32 positions j + j/16 straddle one pad slot:
//      lanes 0 and 31 of every group meet on one bank -> every read of stages 2 and 3 takes two LDS cycles per
//      group instead of one (PMC round 2: SQ_LDS_BANK_CONFLICT = 54 % of the LDS-active cycles).
//   1  "write contiguous, read strided": with digits n = n0 + 16 n1 + 256 n2, k = k0 + 16 k1 + 256 k2
//        stage 1  thread j = n0 + 16 n1 : DFT over n2 -> A[k0] stored at  j + 256 k0
//        stage 2  thread j = n0 + 16 k0 : reads (n0 + 256 k0) + 16 n1, * W256^(n1 k0), DFT over n1 -> B[k1] at j + 256 k1
//        stage 3  thread j = k0 + 16 k1 : reads 16 j + n0 (its own 16 consecutive slots), * W4096^(n0 j), DFT over n0
//      -> X[j + 256 k2], the same ownership as the input.  Physical addresses stay base + immediate:
//      j + j/16 + 272 k (stores: 16 contiguous lanes -> 16 contiguous slots), (j & 15) + 272 (j >> 4) + 17 t
//      (stage-2 loads: two runs of 16 slots 272 = 16 (mod 32) apart) and 17 j + t (stage-3 loads: 17 is odd, so 32
//      consecutive j hit 32 different slots mod 32) -- no conflicts on either side.
//   2  layout 1 with the sixteen loads of a stage issued as single ds_read_b64 from one asm statement (default): the
//      compiler pairs them into ds_read2_b64, which the LDS serves at half the bytes per clock.
// Measured on cfg 4 (one stream, same box, profiles/r03_experiments.txt): 4.01 / 3.73 / 3.69 ms for XCH 0 / 1 / 2.
// Sixteen ds_read_b64 at base + t * STRIDE_B that the load/store optimiser cannot pair into ds_read2_b64 (two
// 16-lane-group accesses with 32-dword banking at half the bytes per clock, MI355X_MICROARCH.md LDS table): the
// reads are issued from one asm statement, which also waits for them (the compiler does not track asm loads).
template <int STRIDE_B>
__device__ __forceinline__ void lds_read16_b64(cpx (&v)[16], const cpx *p)
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
    for (int t = 0; t < 16; ++t) v[t] = __builtin_bit_cast(cpx, d[t]);
}

template <bool INV, int XCH>
__device__ __forceinline__ void row_fft4096(cpx (&v)[16], cpx *lds, const cpx *twB, const cpx *twA, int j)
{
    dft16<INV>(v);                                             // stage 1
    const int kb = j & 15, jh = j >> 4;
    if (XCH == 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) lds[pad16(16 * j + k)] = v[DFT16_AT(k)];
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) lds[j + jh + 272 * k] = v[DFT16_AT(k)];
    }
    __syncthreads();
    if (XCH == 2) lds_read16_b64<17 * 8>(v, lds + kb + 272 * jh);
#pragma unroll
    for (int t = 0; t < 16; ++t) {                             // stage 2: twiddle W256^(t k0)
        cpx x = XCH == 0 ? lds[pad16(j + 256 * t)] : (XCH == 2 ? v[t] : lds[kb + 272 * jh + 17 * t]);
        if (t > 0) {
            const cpx w = twB[16 * t + (XCH == 0 ? kb : jh)];    // [t][k0]: broadcast within a group
            x = INV ? cmulc(x, w) : cmul(x, w);
        }
        v[t] = x;
    }
    __syncthreads();
    dft16<INV>(v);
    if (XCH == 0) {
        const int j0 = jh * 256 + kb;
#pragma unroll
        for (int k = 0; k < 16; ++k) lds[pad16(j0 + 16 * k)] = v[DFT16_AT(k)];
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) lds[j + jh + 272 * k] = v[DFT16_AT(k)];
    }
    __syncthreads();
    if (XCH == 2) lds_read16_b64<8>(v, lds + 17 * j);
#pragma unroll
    for (int t = 0; t < 16; ++t) {                             // stage 3: twiddle W4096^(t j)
        cpx x = XCH == 0 ? lds[pad16(j + 256 * t)] : (XCH == 2 ? v[t] : lds[17 * j + t]);
        if (t > 0) {
            // W4096^(t j) = W4096^(t (j & 15)) * W256^(t (j >> 4)): two [t][.] tables, conflict-free
            const cpx w = cmul(twA[16 * t + kb], twB[16 * t + jh]);
            x = INV ? cmulc(x, w) : cmul(x, w);
        }
        v[t] = x;
    }
    __syncthreads();
    dft16<INV>(v);
    // natural order: X[j + 256 k] = v[DFT16_AT(k)]
    cpx o[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) o[k] = v[DFT16_AT(k)];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = o[k];
}

// Row -> workgroup mapping (MAP):
//   0  row = blockIdx (pair-major, as stored)
//   1  XCD-aware: workgroup b runs on XCD b % 8 (observed dispatch order, MI355X_MICROARCH.md; only speed
//      depends on it).  Each XCD owns the spectrum rows k1 = xcd (mod 8) and walks them k1-major, so the
//      `np` frame pairs of the slab that share one spectrum row Hp[k1] are processed back to back by
//      neighbouring workgroups of ONE XCD: the 32 KB row is fetched from memory once per slab and
//      hit in that XCD's L2 by the other np - 1 rows (was: re-fetched for about every second pair,
//      +25 % read traffic of this pass).
// (Keeping Hp[k1] in registers and looping a workgroup over the pairs needs 168 VGPRs -> spills at 3 waves/SIMD.)
template <int MAP, int XCH>
__global__ void __launch_bounds__(256, 4)
ols_row4096_kernel(cpx *__restrict__ T, const cpx *__restrict__ Hp, const cpx *__restrict__ tw256g,
                   const cpx *__restrict__ t4log, const cpx *__restrict__ t4hig,
                   const cpx *__restrict__ tlo, const cpx *__restrict__ thi, const cpx *__restrict__ tu,
                   int64_t Nmask, int P2, int64_t npairs)
{
    constexpr int N2 = 4096;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cpx *lds = (cpx *)smem;                      // [4096 + 256]
    cpx *twB = lds + N2 + N2 / 16;               // [16][16]  W256^(t k)
    cpx *twA = twB + 256;                        // [16][16]  W4096^(t a)
    const int j = threadIdx.x;
    twB[j] = tw256g[((j >> 4) * (j & 15)) & 255];
    twA[j] = t4log[j];
    typedef const float __attribute__((address_space(4))) *cfp;
    const cfp tuc = (cfp)(uintptr_t)tu;          // uniform per row: W_N^(256 k1 t)
    const unsigned umask = (unsigned)(Nmask >> 8);
    int k1;
    int64_t p;
    if (MAP == 0) {
        k1 = (int)(blockIdx.x % OLS_N1);
        p = blockIdx.x / OLS_N1;
    } else {
        const unsigned xcd = blockIdx.x & 7u, m = blockIdx.x >> 3;
        k1 = (int)((m / (unsigned)npairs) * 8u + xcd);
        p = m % (unsigned)npairs;
    }
    __syncthreads();                             // tables visible
    const cpx *hrow = Hp + (int64_t)k1 * N2;
    const unsigned ml = (unsigned)(k1 * j);
    const cpx wl = cmul(tlo[ml & 511], thi[ml >> 9]);
    {
        cpx *base = T + (p * OLS_N1 + k1) * P2;
        if (XCH == 3) {
            // packed arithmetic (fftpk.h): same exchange layout as XCH 1 / 2, the butterflies and products as v_pk_* with operand
            // selectors: ~900 instead of 1420 vector instructions per wave and row
            using pk::v2f;
            const v2f Wc = {0.92387953251128675613f, 0.38268343236508977173f}, Wr = {0.70710678118654752440f, 0.70710678118654752440f};
            v2f u[16], h[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) h[t] = ((const v2f *)base)[j + 256 * t];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const unsigned iu = 2u * ((unsigned)(k1 * t) & umask);
                const cpx w = cmul(wl, make_float2(tuc[iu], tuc[iu + 1]));          // W_N^(k1 (j + 256 t))
                u[t] = pk::pk_cmul<false>(h[t], __builtin_bit_cast(v2f, w));
            }
            pk::fft4096_pk<false>(u, (v2f *)lds, (const v2f *)twB, (const v2f *)twA, j, Wc, Wr);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 16; ++t) h[t] = ((const v2f *)hrow)[j + 256 * t];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 16; ++t) u[t] = pk::pk_cmul<false>(u[t], h[t]);
            __builtin_amdgcn_sched_barrier(0);
            pk::fft4096_pk<true>(u, (v2f *)lds, (const v2f *)twB, (const v2f *)twA, j, Wc, Wr);
            float wlx = wl.x, wly = wl.y;
            asm volatile("" : "+v"(wlx), "+v"(wly));
            const cpx wl2 = make_float2(wlx, wly);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const unsigned iu = 2u * ((unsigned)(k1 * t) & umask);
                const cpx w = cmul(wl2, make_float2(tuc[iu], tuc[iu + 1]));
                ((v2f *)base)[j + 256 * t] = pk::pk_cmul<true>(u[t], __builtin_bit_cast(v2f, w));
            }
            return;
        }
        cpx v[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const unsigned iu = 2u * ((unsigned)(k1 * t) & umask);
            const cpx ut = make_float2(tuc[iu], tuc[iu + 1]);
            v[t] = cmul(base[j + 256 * t], cmul(wl, ut));
        }
        row_fft4096<false, XCH == 3 ? 2 : XCH>(v, lds, twB, twA, j);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = cmul(v[t], hrow[j + 256 * t]);
        __builtin_amdgcn_sched_barrier(0);
        row_fft4096<true, XCH == 3 ? 2 : XCH>(v, lds, twB, twA, j);
        float wlx = wl.x, wly = wl.y;
        asm volatile("" : "+v"(wlx), "+v"(wly));     // recompute, do not keep 16 twiddles live (see row1024)
        const cpx wl2 = make_float2(wlx, wly);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const unsigned iu = 2u * ((unsigned)(k1 * t) & umask);
            const cpx ut = make_float2(tuc[iu], tuc[iu + 1]);
            base[j + 256 * t] = cmulc(v[t], cmul(wl2, ut));
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Row pass B for 8192-point rows (N = 2^21 = 256 x 8192): the row transform is ONE radix-2 step in registers around two
// 4096-point transforms that run one after the other through the same exchange buffer (the construction of
// ols_lds8192_kernel, olslds.hip): thread j holds z[j + 256 t], t < 32;  a = z_lo + z_hi,  b = (z_lo - z_hi) W8192^n,
// FFT_4096(a) = even bins, FFT_4096(b) = odd bins, and the mirror image on the way back.  The spectrum row is stored in that
// order ([even | odd], each half pair-interleaved so that a thread reads two bins per 16-byte load): the order of the bins is
// irrelevant to a convolution.  Why 8192-sample rows: frames of 2^21 points waste 3.3 % of a block on the 66 559-tap overlap
// instead of 6.7 %, and a row of the recursion pass (ols_col_fwd16_sos_kernel) pays its warm-up once per 8192 samples.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int row8192_at(int k2)          // position of bin k2 inside a spectrum row
{
    const int h = k2 & 1, m = k2 >> 1, t = m >> 8, jj = m & 255;
    return ((h * 8 + (t >> 1)) * 256 + jj) * 2 + (t & 1);
}
inline int row8192_at_host(int k2)
{
    const int h = k2 & 1, m = k2 >> 1, t = m >> 8, jj = m & 255;
    return ((h * 8 + (t >> 1)) * 256 + jj) * 2 + (t & 1);
}

template <int MAP>                               // 0: row = blockIdx; 1: the XCD-aware map of ols_row4096_kernel (a spectrum row is an L2 hit for all pairs but one)
__global__ void __launch_bounds__(256, 4)
ols_row8192_kernel(cpx *__restrict__ T, const cpx *__restrict__ Hp, const cpx *__restrict__ tw256g,
                   const cpx *__restrict__ t4log, const cpx *__restrict__ tlo, const cpx *__restrict__ thi,
                   const cpx *__restrict__ tu, const cpx *__restrict__ w8kg, int64_t Nmask, int P2, int64_t npairs)
{
    using pk::v2f;
    using pk::v4f;
    constexpr int N2 = 8192;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cpx *lds = (cpx *)smem;                      // [4096 + 256]
    cpx *twBc = lds + 4096 + 256;                // [16][16]  W256^(t k)
    cpx *twAc = twBc + 256;                      // [16][16]  W4096^(t a)
    const int j = threadIdx.x;
    twBc[j] = tw256g[((j >> 4) * (j & 15)) & 255];
    twAc[j] = t4log[j];
    typedef const float __attribute__((address_space(4))) *cfp;
    const cfp tuc = (cfp)(uintptr_t)tu;          // uniform per row: W_N^(256 k1 t)
    const unsigned umask = (unsigned)(Nmask >> 8);
    int k1;
    int64_t p;
    if (MAP == 0) {
        k1 = (int)(blockIdx.x % OLS_N1);
        p = blockIdx.x / OLS_N1;
    } else {
        const unsigned xcd = blockIdx.x & 7u, m = blockIdx.x >> 3;
        k1 = (int)((m / (unsigned)npairs) * 8u + xcd);
        p = m % (unsigned)npairs;
    }
    const v2f wj = ((const v2f *)w8kg)[j];       // W8192^j
    __syncthreads();                             // tables visible
    const v2f *twB = (const v2f *)twBc, *twA = (const v2f *)twAc;
    const v4f *Hq = (const v4f *)(Hp + (int64_t)k1 * N2);
    const unsigned ml = (unsigned)(k1 * j);
    const cpx wl = cmul(tlo[ml & 511], thi[ml >> 9]);          // W_N^(k1 j)
    cpx *base = T + (p * OLS_N1 + k1) * P2;
    const v2f Wc = {0.92387953251128675613f, 0.38268343236508977173f}, Wr = {0.70710678118654752440f, 0.70710678118654752440f};
    // wave-uniform row base + one 32-bit lane offset: the 32 addresses of a row are immediates and adds, not 64-bit registers
    char *rowb = (char *)base;
    unsigned roff = (unsigned)j * (unsigned)sizeof(cpx);
    asm volatile("" : "+v"(roff));
    v2f a[16], b[16];
#pragma unroll
    for (int half = 0; half < 2; ++half) {         // two batches of loads: 32 registers in flight, not 64
        v2f lo[8], hi[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int t = 8 * half + u;
            lo[u] = *(const v2f *)(rowb + (roff + 2048u * (unsigned)t));
            hi[u] = *(const v2f *)(rowb + (roff + 2048u * (unsigned)(t + 16)));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int t = 8 * half + u;
            const unsigned i0 = 2u * ((unsigned)(k1 * t) & umask), i1 = 2u * ((unsigned)(k1 * (t + 16)) & umask);
            const cpx w0 = cmul(wl, make_float2(tuc[i0], tuc[i0 + 1])), w1 = cmul(wl, make_float2(tuc[i1], tuc[i1 + 1]));   // W_N^(k1 n2)
            const v2f l = pk::pk_cmul<false>(lo[u], __builtin_bit_cast(v2f, w0)), h = pk::pk_cmul<false>(hi[u], __builtin_bit_cast(v2f, w1));
            const v2f w = t ? pk::pk_cmul<false>(wj, twB[16 * t + 8]) : wj;                     // W8192^(j + 256 t) = W8192^j W32^t
            a[t] = l + h;
            b[t] = pk::pk_cmul<false>(l - h, w);
        }
        __builtin_amdgcn_sched_barrier(0);
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
    float wlx = wl.x, wly = wl.y;
    asm volatile("" : "+v"(wlx), "+v"(wly));     // recompute the row twiddles, do not keep 32 of them live
    const cpx wl2 = make_float2(wlx, wly);
    v2f wj2 = wj;
    asm volatile("" : "+v"(wj2));                // likewise the sixteen W8192^(j + 256 t): two instructions each, not 32 live registers
    unsigned roff2 = (unsigned)j * (unsigned)sizeof(cpx);
    asm volatile("" : "+v"(roff2));
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const v2f w = t ? pk::pk_cmul<false>(wj2, twB[16 * t + 8]) : wj2;
        const v2f bw = pk::pk_cmul<true>(b[t], w);
        const unsigned i0 = 2u * ((unsigned)(k1 * t) & umask), i1 = 2u * ((unsigned)(k1 * (t + 16)) & umask);
        const cpx w0 = cmul(wl2, make_float2(tuc[i0], tuc[i0 + 1])), w1 = cmul(wl2, make_float2(tuc[i1], tuc[i1 + 1]));
        *(v2f *)(rowb + (roff2 + 2048u * (unsigned)t)) = pk::pk_cmul<true>(a[t] + bw, __builtin_bit_cast(v2f, w0));
        *(v2f *)(rowb + (roff2 + 2048u * (unsigned)(t + 16))) = pk::pk_cmul<true>(a[t] - bw, __builtin_bit_cast(v2f, w1));
    }
}

// The spectrum of a long kernel on the device (N = 2^20): the taps go through the forward column pass as a one-frame
// "signal" (zero fill outside the taps) and then through this FORWARD-ONLY row pass, which leaves conj(X[k1 + 256 k2]) / N at
// [k1][k2] -- exactly the layout and scaling the row pass multiplies by.  Replaces a float64 host FFT of 2^20 points
// (4-40 ms of the first call with a new filter, depending on how many host cores the process gets) by two launches.
__global__ void __launch_bounds__(256, 4)
ols_rowspec4096_kernel(cpx *__restrict__ T, const cpx *__restrict__ tw256g, const cpx *__restrict__ t4log,
                       const cpx *__restrict__ tlo, const cpx *__restrict__ thi, const cpx *__restrict__ tu,
                       int64_t Nmask, int P2, float inv_n)
{
    using pk::v2f;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cpx *lds = (cpx *)smem;
    cpx *twB = lds + 4096 + 256;
    cpx *twA = twB + 256;
    const int j = threadIdx.x;
    twB[j] = tw256g[((j >> 4) * (j & 15)) & 255];
    twA[j] = t4log[j];
    typedef const float __attribute__((address_space(4))) *cfp;
    const cfp tuc = (cfp)(uintptr_t)tu;
    const unsigned umask = (unsigned)(Nmask >> 8);
    const int k1 = (int)blockIdx.x;
    __syncthreads();
    const unsigned ml = (unsigned)(k1 * j);
    const cpx wl = cmul(tlo[ml & 511], thi[ml >> 9]);
    cpx *base = T + (int64_t)k1 * P2;
    const v2f Wc = {0.92387953251128675613f, 0.38268343236508977173f}, Wr = {0.70710678118654752440f, 0.70710678118654752440f};
    v2f u[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const unsigned iu = 2u * ((unsigned)(k1 * t) & umask);
        const cpx w = cmul(wl, make_float2(tuc[iu], tuc[iu + 1]));
        u[t] = pk::pk_cmul<false>(((const v2f *)base)[j + 256 * t], __builtin_bit_cast(v2f, w));
    }
    pk::fft4096_pk<false>(u, (v2f *)lds, (const v2f *)twB, (const v2f *)twA, j, Wc, Wr);
#pragma unroll
    for (int t = 0; t < 16; ++t) base[j + 256 * t] = make_float2(u[t].x * inv_n, -u[t].y * inv_n);
}

// The same for 8192-point rows (N = 2^21): forward-only, conj(X[k1 + 256 k2]) / N written in the bin order
// ols_row8192_kernel multiplies by (row8192_at).  In place: a thread's 32 loads of the row precede its stores by four barriers.
__global__ void __launch_bounds__(256, 4)
ols_rowspec8192_kernel(cpx *__restrict__ T, const cpx *__restrict__ tw256g, const cpx *__restrict__ t4log,
                       const cpx *__restrict__ tlo, const cpx *__restrict__ thi, const cpx *__restrict__ tu,
                       const cpx *__restrict__ w8kg, int64_t Nmask, int P2, float inv_n)
{
    using pk::v2f;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cpx *lds = (cpx *)smem;
    cpx *twBc = lds + 4096 + 256;
    cpx *twAc = twBc + 256;
    const int j = threadIdx.x;
    twBc[j] = tw256g[((j >> 4) * (j & 15)) & 255];
    twAc[j] = t4log[j];
    typedef const float __attribute__((address_space(4))) *cfp;
    const cfp tuc = (cfp)(uintptr_t)tu;
    const unsigned umask = (unsigned)(Nmask >> 8);
    const int k1 = (int)blockIdx.x;
    const v2f wj = ((const v2f *)w8kg)[j];
    __syncthreads();
    const v2f *twB = (const v2f *)twBc, *twA = (const v2f *)twAc;
    const unsigned ml = (unsigned)(k1 * j);
    const cpx wl = cmul(tlo[ml & 511], thi[ml >> 9]);
    cpx *base = T + (int64_t)k1 * P2;
    const v2f Wc = {0.92387953251128675613f, 0.38268343236508977173f}, Wr = {0.70710678118654752440f, 0.70710678118654752440f};
    v2f a[16], b[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const unsigned i0 = 2u * ((unsigned)(k1 * t) & umask), i1 = 2u * ((unsigned)(k1 * (t + 16)) & umask);
        const cpx w0 = cmul(wl, make_float2(tuc[i0], tuc[i0 + 1])), w1 = cmul(wl, make_float2(tuc[i1], tuc[i1 + 1]));
        const v2f l = pk::pk_cmul<false>(((const v2f *)base)[j + 256 * t], __builtin_bit_cast(v2f, w0));
        const v2f h = pk::pk_cmul<false>(((const v2f *)base)[j + 256 * (t + 16)], __builtin_bit_cast(v2f, w1));
        const v2f w = t ? pk::pk_cmul<false>(wj, twB[16 * t + 8]) : wj;
        a[t] = l + h;
        b[t] = pk::pk_cmul<false>(l - h, w);
    }
    pk::fft4096_pk<false>(a, (v2f *)lds, twB, twA, j, Wc, Wr);
    pk::fft4096_pk<false>(b, (v2f *)lds, twB, twA, j, Wc, Wr);
#pragma unroll
    for (int t = 0; t < 16; ++t) {          // even bins 2 (j + 256 t) from a, odd bins from b
        base[((0 * 8 + (t >> 1)) * 256 + j) * 2 + (t & 1)] = make_float2(a[t].x * inv_n, -a[t].y * inv_n);
        base[((1 * 8 + (t >> 1)) * 256 + j) * 2 + (t & 1)] = make_float2(b[t].x * inv_n, -b[t].y * inv_n);
    }
}

}  // namespace tfx
