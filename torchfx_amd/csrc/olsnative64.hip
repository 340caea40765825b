// olsnative64.hip -- the three-pass overlap-save pipeline of olsnative.hip for float64 signals with more than 4096 taps.
//
// The reference keeps a float64 signal float64 through FIR.forward (src/torchfx/filter/fir.py:526-579 -> fft_conv1d,
// filter/_fftconv.py:70-141: rfft / irfft in the signal's dtype).  Up to 4096 taps the one-launch LDS kernels (olslds.hip) serve
// float64; above that rounds 1-5 fell to rocFFT (frame, r2c, multiply, c2r, un-frame: ~190 B of HBM traffic per output sample in
// float64).  Here: the same decomposition as the float32 pipeline -- N = 2^20 = 256 x 4096, two real frames per complex
// transform (z = a + i b; the taps are real, so conv(z) = conv(a) + i conv(b)), column pass A (256-point FFTs over n1 straight
// from the signal), row pass B (four-step twiddle, 4096-point FFT, x spectrum row, inverse, conjugate twiddle, in place), column
// pass C (inverse, only the S valid samples stored) -- in plain float64 arithmetic on the templated primitives of ldsfft.h
// (no packed math: there is no v_pk_*_f64), 16-byte complex elements, 16 columns per workgroup in the column passes (one
// 128-byte line of float64 signal, 64 KB of LDS).  20 N / S x 2 + 8 = ~51 B per output sample.  Spectrum and all tables are
// computed on the host in float64 (long double angles).  Frames start on 128-byte lines of memory for any row length
// (row_shift, as in the float32 pipeline).  Streaming history and blocks other than 2^20 stay with rocFFT (fftconv.hip).
#include "common.h"
#include "epilogue.h"
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

namespace ols64 {

using ldsfft::cx;
using ldsfft::cmul;
using ldsfft::cmulc;
using ldsfft::mk;
typedef cx<double> cpd;

constexpr int N1 = 256, N2 = 4096, CB = 16;          // N = N1 * N2; CB columns per workgroup in the column passes
constexpr int64_t NPTS = (int64_t)N1 * N2;

struct Geom {
    int64_t Tn, Tout, F, S, pad_left, nframes;
    int sh_on, sh_base;                              // rows that are not whole 128-byte lines: see row_shift (olsnative.hip)
    int *nf_pair;                                    // [C] zeroed per call, or null: see the straddling-pair branch of ols_col_fwd16_kernel (ols_kernels.h)
};
__device__ __forceinline__ int row_shift(const Geom &g, int64_t c) { return g.sh_on ? (int)(((int64_t)g.sh_base + c * g.Tn) & 15) : 0; }

// 256-point column transform, radix (16, 16): thread (col = tid & 15, j = tid >> 4) owns rows j + 16 t of its column.
// One exchange through lds[256][CB]; out: row j + 16 k at v[LDS_DFT16_AT(k)].
template <bool INV>
__device__ __forceinline__ void col_fft256(cpd (&v)[16], cpd *lds, const cpd *tw256, int col, int j)
{
    ldsfft::dft16<double, INV>(v);
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[(16 * j + k) * CB + col] = v[LDS_DFT16_AT(k)];
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        cpd x = lds[(j + 16 * t) * CB + col];
        if (t > 0) {
            const cpd w = tw256[(t * j) & 255];
            x = INV ? cmulc(x, w) : cmul(x, w);
        }
        v[t] = x;
    }
    ldsfft::dft16<double, INV>(v);
}

__global__ void __launch_bounds__(256, 2)
ols64_col_fwd_kernel(const double *__restrict__ x, cpd *__restrict__ T, const cpd *__restrict__ tw256g, Geom g, int64_t frame0)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cpd *lds = (cpd *)smem;                          // [256][16]
    cpd *tw256 = lds + N1 * CB;
    const int tid = threadIdx.x, col = tid & 15, j = tid >> 4;
    tw256[tid] = tw256g[tid];
    constexpr int ncb = N2 / CB;
    const int64_t pair = blockIdx.x / ncb;
    const int n2 = (int)(blockIdx.x % ncb) * CB + col;
    const int64_t fa = frame0 + 2 * pair, fb = fa + 1;
    const int64_t ca = fa / g.F, ia0 = (fa % g.F) * g.S - g.pad_left - row_shift(g, ca);
    const bool has_b = fb < g.nframes;
    const int64_t cb_ = has_b ? fb / g.F : 0, ib0 = has_b ? (fb % g.F) * g.S - g.pad_left - row_shift(g, cb_) : 0;
    const double *xa = x + ca * g.Tn, *xb = x + cb_ * g.Tn;
    cpd v[16];
    const bool inner = ia0 >= 0 && ia0 + NPTS <= g.Tn && has_b && ib0 >= 0 && ib0 + NPTS <= g.Tn;
    if (inner) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int64_t n = (int64_t)(j + 16 * t) * N2 + n2;
            v[t] = mk<double>(__builtin_nontemporal_load(xa + ia0 + n), __builtin_nontemporal_load(xb + ib0 + n));
        }
    } else {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int64_t n = (int64_t)(j + 16 * t) * N2 + n2;
            const int64_t ia = ia0 + n, ib = ib0 + n;
            v[t] = mk<double>((ia >= 0 && ia < g.Tn) ? xa[ia] : 0.0, (has_b && ib >= 0 && ib < g.Tn) ? xb[ib] : 0.0);
        }
    }
    if (has_b && ca != cb_ && g.nf_pair) {           // a pair that straddles two signal rows: row cb's non-finite samples must not reach row ca
        bool bad = false;
#pragma unroll
        for (int t = 0; t < 16; ++t)
            if (!(__builtin_fabs(v[t].y) <= 1.7976931348623157e308)) { v[t].y = 0.0; bad = true; }
        if (bad) g.nf_pair[cb_] = 1;
    }
    __syncthreads();
    col_fft256<false>(v, lds, tw256, col, j);
    cpd *Tp = T + pair * NPTS;
#pragma unroll
    for (int k = 0; k < 16; ++k) Tp[(int64_t)(j + 16 * k) * N2 + n2] = v[LDS_DFT16_AT(k)];
}

__global__ void __launch_bounds__(256, 2)
ols64_col_inv_kernel(const cpd *__restrict__ T, double *__restrict__ y, const cpd *__restrict__ tw256g, Geom g, int64_t frame0)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cpd *lds = (cpd *)smem;
    cpd *tw256 = lds + N1 * CB;
    const int tid = threadIdx.x, col = tid & 15, j = tid >> 4;
    tw256[tid] = tw256g[tid];
    constexpr int ncb = N2 / CB;
    const int64_t pair = blockIdx.x / ncb;
    const int n2 = (int)(blockIdx.x % ncb) * CB + col;
    const cpd *Tp = T + pair * NPTS;
    cpd v[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = Tp[(int64_t)(j + 16 * t) * N2 + n2];
    __syncthreads();
    col_fft256<true>(v, lds, tw256, col, j);
    const int64_t fa = frame0 + 2 * pair, fb = fa + 1;
    const int64_t ca = fa / g.F, oa0 = (fa % g.F) * g.S - row_shift(g, ca);
    const bool has_b = fb < g.nframes;
    const int64_t cb_ = has_b ? fb / g.F : 0, ob0 = has_b ? (fb % g.F) * g.S - row_shift(g, cb_) : 0;
    double *ya = y + ca * g.Tout, *yb = y + cb_ * g.Tout;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int64_t n = (int64_t)(j + 16 * k) * N2 + n2;
        if (n < g.S) {                                   // the valid part of the block
            const cpd o = v[LDS_DFT16_AT(k)];
            const int64_t oa = oa0 + n, ob = ob0 + n;
            if (oa >= 0 && oa < g.Tout) __builtin_nontemporal_store(o.x, ya + oa);
            if (has_b && ob >= 0 && ob < g.Tout) __builtin_nontemporal_store(o.y, yb + ob);
        }
    }
}

__global__ void __launch_bounds__(256) ols64_straddle_fix_kernel(double *__restrict__ y, Geom g)
{
    const int64_t c = blockIdx.x;                                    // rows on x: no 65 535 limit
    if (!g.nf_pair[c]) return;
    const int64_t hi = min(g.Tout, g.S - row_shift(g, c));          // frame 0 of row c
    for (int64_t t = (int64_t)blockIdx.y * 256 + threadIdx.x; t < hi; t += (int64_t)gridDim.y * 256) y[c * g.Tout + t] = __builtin_nan("");
}

// Row pass: one workgroup per row k1 of a frame pair; thread j holds T[k1][j + 256 t].
//   W_N^(k1 (j + 256 t)) = W_N^(k1 j) * W_4096^(k1 t):  tlo[m & 511] thi[m >> 9] (m = k1 j < 2^16) and tu[(k1 t) & 4095].
__global__ void __launch_bounds__(256, 2)
ols64_row_kernel(cpd *__restrict__ T, const cpd *__restrict__ Hp, const cpd *__restrict__ tw256g, const cpd *__restrict__ t4log,
                 const cpd *__restrict__ tlo, const cpd *__restrict__ thi, const cpd *__restrict__ tu)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cpd *lds = (cpd *)smem;                          // [4096 + 256]
    cpd *twB = lds + N2 + N2 / 16;                   // [16][16]  W256^(t k)
    cpd *twA = twB + 256;                            // [16][16]  W4096^(t a)
    const int j = threadIdx.x;
    twB[j] = tw256g[((j >> 4) * (j & 15)) & 255];
    twA[j] = t4log[j];
    const int k1 = (int)(blockIdx.x % N1);
    cpd *base = T + (int64_t)blockIdx.x * N2;        // rows are stored pair-major: row = pair * 256 + k1
    const cpd *hrow = Hp + (int64_t)k1 * N2;
    const unsigned ml = (unsigned)(k1 * j);
    const cpd wl = cmul(tlo[ml & 511], thi[ml >> 9]);
    __syncthreads();
    cpd v[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = cmul(base[j + 256 * t], cmul(wl, tu[(k1 * t) & 4095]));
    ldsfft::fft4096<double, false>(v, lds, twB, twA, j);
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = cmul(v[t], hrow[j + 256 * t]);
    ldsfft::fft4096<double, true>(v, lds, twB, twA, j);
#pragma unroll
    for (int t = 0; t < 16; ++t) base[j + 256 * t] = cmulc(v[t], cmul(wl, tu[(k1 * t) & 4095]));
}

// ---- host -----------------------------------------------------------------------------------------------------------------
struct Plan {
    cpd *Hp = nullptr, *tw256 = nullptr, *t4lo = nullptr, *tlo = nullptr, *thi = nullptr, *tu = nullptr;
    std::shared_ptr<void> owner;                     // one device allocation: spectrum | tables
};
static std::mutex g_mu;
static std::map<std::vector<char>, Plan> g_plans;
static bool g_attr[TFX_MAX_DEVICES] = {};

static cpd W(long double num, long double den)
{
    const long double a = -2.0L * 3.14159265358979323846264338327950288L * num / den;
    cpd w; w.x = (double)cosl(a); w.y = (double)sinl(a);
    return w;
}

static Plan get_plan(const double *kf, int64_t K, int64_t lead, hipStream_t stream)
{
    std::lock_guard<std::mutex> lk(g_mu);
    std::vector<char> key((const char *)kf, (const char *)kf + K * sizeof(double));
    key.push_back((char)lead);
    key.push_back((char)current_device());
    auto it = g_plans.find(key);
    if (it != g_plans.end()) return it->second;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    TFX_CHECK(!(hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone),
              "overlap-save: first use of this filter (%lld taps) inside a stream capture -- run it once before capturing "
              "(its spectrum and tables are uploaded with blocking copies)", (long long)K);
    if (g_plans.size() >= 8) g_plans.clear();        // 16 MB of spectrum each; a plan lives on with whoever still holds it
    // conj(FFT(taps behind `lead` zeros, zero padded to N)) / N   (_fftconv.py:123-124,131 + irfft scaling), row k1 = k % 256
    std::vector<double> re((size_t)NPTS, 0.0), im((size_t)NPTS, 0.0);
    for (int64_t i = 0; i < K; ++i) re[(size_t)(lead + i)] = kf[i];
    host_fft_f64(re, im);
    std::vector<cpd> all((size_t)NPTS + 256 + 256 + 512 + 128 + 4096);
    const double inv_n = 1.0 / (double)NPTS;
    for (int64_t k = 0; k < NPTS; ++k) {
        const int64_t k1 = k % N1, k2 = k / N1;      // k = k1 + 256 k2
        all[(size_t)(k1 * N2 + k2)].x = re[(size_t)k] * inv_n;
        all[(size_t)(k1 * N2 + k2)].y = -im[(size_t)k] * inv_n;
    }
    size_t o = (size_t)NPTS;
    const size_t o256 = o;
    for (int i = 0; i < 256; ++i) all[o++] = W(i, 256);
    const size_t o4 = o;
    for (int t = 0; t < 16; ++t)
        for (int a = 0; a < 16; ++a) all[o++] = W(t * a, 4096);
    const size_t olo = o;
    for (int i = 0; i < 512; ++i) all[o++] = W(i, (long double)NPTS);
    const size_t ohi = o;
    for (int i = 0; i < 128; ++i) all[o++] = W(512.0L * i, (long double)NPTS);
    const size_t ou = o;
    for (int i = 0; i < 4096; ++i) all[o++] = W(i, 4096);
    void *d = nullptr;
    TFX_HIP(hipMalloc(&d, all.size() * sizeof(cpd)));
    TFX_HIP(hipMemcpy(d, all.data(), all.size() * sizeof(cpd), hipMemcpyHostToDevice));
    Plan p;
    p.owner = std::shared_ptr<void>(d, [](void *q) { (void)hipFree(q); });
    p.Hp = (cpd *)d;
    p.tw256 = p.Hp + o256; p.t4lo = p.Hp + o4; p.tlo = p.Hp + olo; p.thi = p.Hp + ohi; p.tu = p.Hp + ou;
    g_plans[key] = p;
    return p;
}

}  // namespace ols64

void olsnative64_clear()
{
    std::lock_guard<std::mutex> lk(ols64::g_mu);
    ols64::g_plans.clear();
}

// float64 signals, taps beyond the one-launch kernels' 4096, a signal of at least one 2^20-point block, no streaming history
bool olsnative64_supported(int64_t K, int64_t L, bool has_hist)
{
    if (env_i64("TFX_OLS_NATIVE", 1) == 0 || env_i64("TFX_OLS_NATIVE64", 1) == 0 || env_i64("TFX_FFT_LOG2N", 0) != 0) return false;
    return !has_hist && K > 4096 && 2 * (K + 16) <= ols64::NPTS && L >= ols64::NPTS;
}

void olsnative64_geometry(int64_t K, int64_t Tn, int64_t pl, int64_t pr, int64_t *S_out, int64_t *F_out)
{
    const int64_t Tout = Tn + pl + pr - K + 1;
    const int64_t lead = (16 - (pl % 16)) % 16;
    int64_t S = ols64::NPTS - (K + lead) + 1;
    S -= S % 16;
    if (S_out) *S_out = S;
    if (F_out) *F_out = ceil_div(Tout + ((Tn % 16 != 0) ? 15 : 0), S);
}

void olsnative64_forward(const double *x, double *y, int64_t C, int64_t Tn, const double *kf_host, int64_t K, int64_t pl, int64_t pr,
                         hipStream_t stream)
{
    using namespace ols64;
    Geom g;
    const int64_t L = Tn + pl + pr;
    g.Tn = Tn; g.Tout = L - K + 1;
    const int64_t lead = (16 - (pl % 16)) % 16;      // frames start on 128-byte lines: `lead` zero taps in front of the flipped kernel
    g.pad_left = pl + lead;
    g.S = NPTS - (K + lead) + 1;
    g.S -= g.S % 16;
    g.sh_base = (int)(((uintptr_t)x >> 3) & 15);
    g.sh_on = (Tn % 16 != 0 || g.sh_base != 0) ? 1 : 0;
    g.F = ceil_div(g.Tout + (g.sh_on ? 15 : 0), g.S);
    g.nframes = C * g.F;
    g.nf_pair = nullptr;
    if (C > 1 && (g.F & 1)) {
        g.nf_pair = (int *)scratch("olsn64_nf_pair", (size_t)C * sizeof(int), stream);
        TFX_HIP(hipMemsetAsync(g.nf_pair, 0, (size_t)C * sizeof(int), stream));
    }
    const Plan plan = get_plan(kf_host, K, lead, stream);
    const int dev = current_device();
    constexpr size_t shm_col = (size_t)(N1 * CB + 256) * sizeof(cpd);
    constexpr size_t shm_row = (size_t)(N2 + N2 / 16 + 512) * sizeof(cpd);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_attr[dev]) {
            TFX_HIP(hipFuncSetAttribute((const void *)ols64_col_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_col));
            TFX_HIP(hipFuncSetAttribute((const void *)ols64_col_inv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_col));
            TFX_HIP(hipFuncSetAttribute((const void *)ols64_row_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_row));
            g_attr[dev] = true;
        }
    }
    const int64_t npairs = ceil_div(g.nframes, 2);
    // slabs of frame pairs (16 MB of workspace each): 8 pairs = 128 MB, the live workspace stays in the Infinity Cache
    int64_t slab = std::min<int64_t>(npairs, std::max<int64_t>(1, env_i64("TFX_OLS64_PAIRS_PER_SLAB", 8)));
    cpd *T = nullptr;
    for (;;) {
        T = (cpd *)scratch_try("olsn64_T", (size_t)slab * (size_t)NPTS * sizeof(cpd), stream);
        if (T || slab == 1) break;
        slab = std::max<int64_t>(1, slab / 2);
    }
    if (!T) T = (cpd *)scratch("olsn64_T", (size_t)NPTS * sizeof(cpd), stream);
    constexpr int ncb = N2 / CB;
    for (int64_t p0 = 0; p0 < npairs; p0 += slab) {
        const int64_t np = std::min(slab, npairs - p0);
        {
            ProfScope ps("ols64_col_fwd_kernel", stream);
            hipLaunchKernelGGL(ols64_col_fwd_kernel, dim3((unsigned)(np * ncb)), dim3(256), shm_col, stream, x, T, plan.tw256, g, 2 * p0);
            TFX_HIP(hipGetLastError());
        }
        {
            ProfScope ps("ols64_row_kernel", stream);
            hipLaunchKernelGGL(ols64_row_kernel, dim3((unsigned)(np * N1)), dim3(256), shm_row, stream, T, plan.Hp, plan.tw256, plan.t4lo,
                               plan.tlo, plan.thi, plan.tu);
            TFX_HIP(hipGetLastError());
        }
        {
            ProfScope ps("ols64_col_inv_kernel", stream);
            hipLaunchKernelGGL(ols64_col_inv_kernel, dim3((unsigned)(np * ncb)), dim3(256), shm_col, stream, T, y, plan.tw256, g, 2 * p0);
            TFX_HIP(hipGetLastError());
        }
    }
    if (g.nf_pair) {
        hipLaunchKernelGGL(ols64_straddle_fix_kernel, dim3((unsigned)C, 64), dim3(256), 0, stream, y, g);
        TFX_HIP(hipGetLastError());
    }
}

}  // namespace tfx
