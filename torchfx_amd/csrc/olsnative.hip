// olsnative.hip -- overlap-save convolution with hand-written LDS-resident FFT passes (float32).
//
// Why: with rocFFT the 65536-tap case costs 5 launches per slab (frame, r2c, cmul, c2r, un-frame)
// and ~95 B of HBM traffic per output sample.  Here the whole block pipeline is three kernels and
// 26-31 B/sample (N = 2^20 / 2^18), with the framing, zero padding, spectrum multiply, 1/N scaling and the
// "keep the first S samples" selection all fused into them:
//
//   * two real frames ride one complex transform:  z = frame_a + i frame_b.  The taps are real, so
//     conv(z) = conv(a) + i conv(b): no real-FFT untangling pass, the spectrum multiply stays local.
//   * N = N1 * N2 (N1 = 256, N2 = 256 / 1024 / 4096) four-step decomposition, n = n1*N2 + n2,
//     k = k1 + N1*k2:
//       A  column FFTs over n1 (256-point, 32 adjacent columns per workgroup, data gathered
//          straight from the signal with zero fill)                          -> T[k1][n2]
//       B  per row k1: * W_N^(n2 k1), N2-point FFT over n2, * Hp[k1][k2], inverse FFT,
//          * conj(W_N^(n2 k1)), in place; one wavefront per row (N2 <= 1024) or one workgroup per
//          row (N2 = 4096)                                                     -> T[k1][n2]
//       C  column inverse FFTs over k1; real part -> frame_a's output samples, imaginary part ->
//          frame_b's, only the first S = N-K+1 (valid) samples are stored      -> y
//     Hp[k1][k2] = conj(FFT(kf_pad))[k1 + N1 k2] / N is precomputed once per filter: on the host in float64 for
//     N = 2^16 / 2^18, ON THE DEVICE in float32 by this pipeline's own forward kernels for N = 2^20 / 2^21 (ols_rowspec4096_kernel, ols_rowspec8192_kernel;
//     the reference's rfft of the kernel is float32 too, _fftconv.py:123-124; TFX_OLS_GPU_SPECTRUM=0: host float64).
//   * every FFT is a Stockham autosort in registers + LDS -- radix 16 x 16 in the column passes (one
//     LDS exchange), radix 16 x 16 x 4 / 16 x 16 x 16 in the row passes (two exchanges per direction),
//     radix 4 for N2 = 256; layouts are chosen so all LDS accesses of the column passes are
//     conflict-free ([row][col] with the column on the lane index).
//   * frames start on 128-byte lines (zero taps prepended to the flipped kernel, hop rounded to 32),
//     slabs of frame pairs (64 MB of workspace each: the live workspace stays in the Infinity Cache) rotate over two internal streams.
//
// Semantics = fft_conv1d (src/torchfx/filter/_fftconv.py:70-141): causal correlation with the
// stored flipped kernel, output length T + l + r - K + 1.
#include "common.h"
#include "epilogue.h"
#include "fftpk.h"
#include "ols_kernels.h"
#include "../../include/torchfx_hip.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <future>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

namespace tfx {

// device code -- geometry (OlsGeom), the column and row passes, the cascade-in-pass-A kernel and its fix-up: ols_kernels.h

// ---------------------------------------------------------------------------------------------
// Host: plan (tables + permuted spectrum) cache and orchestration
// ---------------------------------------------------------------------------------------------
struct NativePlan {
    int64_t N = 0, K = 0;
    int N2 = 0;
    cpx *Hp = nullptr, *tw256 = nullptr, *twr = nullptr, *tlo = nullptr, *thi = nullptr, *tu = nullptr, *t4lo = nullptr, *t4hi = nullptr, *w8k = nullptr;
    float *taps_dev = nullptr;      // device copy of the taps while the spectrum kernels may still read it (N = 2^20)
    cpx *tables = nullptr;          // the one allocation tw256 ... t4hi point into
    hipEvent_t ready = nullptr;     // recorded behind the spectrum kernels: other streams wait for it before they read Hp
    hipStream_t ready_stream = nullptr;
    NativePlan() = default;
    NativePlan(const NativePlan &) = delete;
    NativePlan &operator=(const NativePlan &) = delete;
    ~NativePlan()                   // the last owner frees: hipFree waits for the device, i.e. runs behind every launch that used the plan
    {
        for (cpx *q : {Hp, tables}) if (q) (void)hipFree(q);
        if (taps_dev) (void)hipFree(taps_dev);
        if (ready) (void)hipEventDestroy(ready);
    }
};
// Plans are shared_ptr-owned: olsnative_forward keeps its plan alive until its launches are enqueued, whatever another host
// thread evicts meanwhile (round 4 handed out raw pointers and freed every plan when the 18th filter arrived).
typedef std::shared_ptr<NativePlan> NativePlanPtr;
static std::mutex g_np_mu;
static std::map<std::vector<char>, NativePlanPtr> g_nplans;
static int64_t g_free_mb[TFX_MAX_DEVICES] = {};                  // per device: free memory (MB) seen at first use, 0 = not asked yet
static NativePlanPtr g_last_plan[TFX_MAX_DEVICES];             // per device: the plan used last and its key
static std::vector<char> g_last_key[TFX_MAX_DEVICES];

// Forward FFT in float64 of a real sequence of L samples zero-padded to n = 2^m points (the spectrum of the taps, once per
// filter; it sits in the latency of the first call with a new filter).  One decimation-in-frequency step of radix R = 16
// turns the transform into 16 INDEPENDENT n/16-point transforms
//     X[16 m + r] = FFT_{n/16}( W_n^(n' r) * sum_q x[n' + (n/16) q] W_16^(q r) )[m],      q < ceil(L / (n/16)),
// each run by its own host thread on 1 MB of data (n = 2^20) with no synchronisation between them; the zero padding prunes
// the q sum to one or two terms.  2^20 points in ~4 ms instead of ~40 for the staged radix-2 with a thread fork per stage.
static void fft_radix2_inplace(double *re, double *im, size_t n, const double *wr, const double *wi, size_t wstride)
{
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const size_t half = len / 2, step = (n / len) * wstride;
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < half; ++k) {
                const double cr = wr[k * step], ci = wi[k * step];
                const size_t a = i + k, b = a + half;
                const double vr = re[b] * cr - im[b] * ci, vi = re[b] * ci + im[b] * cr;
                re[b] = re[a] - vr; im[b] = im[a] - vi;
                re[a] += vr; im[a] += vi;
            }
    }
}

static void host_fft(std::vector<double> &re, std::vector<double> &im)
{
    const size_t n = re.size();
    size_t L = n;                                  // support of the input: trailing zeros are pruned
    while (L > 0 && re[L - 1] == 0.0 && im[L - 1] == 0.0) --L;
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t R = n >= 65536 ? 16 : 1;                 // small transforms (the 4096-point spectra of olslds.hip): not worth the threads
    const size_t nthr = R > 1 ? std::min<size_t>(R, hw ? hw : 1) : 1;
    const size_t M = n / R, Q = (L + M - 1) / M;                 // sub-transform length, non-zero input blocks
    std::vector<double> wr(M / 2 ? M / 2 : 1), wi(M / 2 ? M / 2 : 1);      // W_M^k, k < M / 2: contiguous, 512 KB at n = 2^20
    for (size_t k = 0; k < M / 2; ++k) { const double a = -2.0 * M_PI * (double)k / (double)M; wr[k] = cos(a); wi[k] = sin(a); }
    if (R == 1) { fft_radix2_inplace(re.data(), im.data(), n, wr.data(), wi.data(), 1); return; }
    const std::vector<double> xr(re.begin(), re.begin() + (ptrdiff_t)std::min(n, Q * M)), xi(im.begin(), im.begin() + (ptrdiff_t)std::min(n, Q * M));
    std::vector<std::vector<double>> Yr(R), Yi(R);
    auto sub = [&](size_t r) {
        std::vector<double> yr(M), yi(M);
        double cq[16], sq[16];                                    // W_16^(q r)
        for (size_t q = 0; q < Q; ++q) { const double a = -2.0 * M_PI * (double)((q * r) % R) / (double)R; cq[q] = cos(a); sq[q] = sin(a); }
        std::vector<double> th_c((M + 255) / 256), th_s((M + 255) / 256), tl_c(256), tl_s(256);
        for (size_t i = 0; i < th_c.size(); ++i) { const double a = -2.0 * M_PI * (double)((256 * i * r) & (n - 1)) / (double)n; th_c[i] = cos(a); th_s[i] = sin(a); }
        for (size_t i = 0; i < 256; ++i) { const double a = -2.0 * M_PI * (double)((i * r) & (n - 1)) / (double)n; tl_c[i] = cos(a); tl_s[i] = sin(a); }
        for (size_t np = 0; np < M; ++np) {
            double ar = 0.0, ai = 0.0;
            for (size_t q = 0; q < Q; ++q) {                      // sum_q x[n' + M q] W_16^(q r)
                const double vr = xr[np + M * q], vi = xi[np + M * q];
                ar += vr * cq[q] - vi * sq[q]; ai += vr * sq[q] + vi * cq[q];
            }
            // W_n^(n' r) = W_n^(256 (n' >> 8) r) * W_n^((n' & 255) r): two small exact tables, one product
            const size_t hi_ = np >> 8, lo_ = np & 255;
            const double c = th_c[hi_] * tl_c[lo_] - th_s[hi_] * tl_s[lo_], s_ = th_c[hi_] * tl_s[lo_] + th_s[hi_] * tl_c[lo_];
            yr[np] = ar * c - ai * s_; yi[np] = ar * s_ + ai * c;
        }
        fft_radix2_inplace(yr.data(), yi.data(), M, wr.data(), wi.data(), 1);
        Yr[r].swap(yr); Yi[r].swap(yi);
    };
    // X[16 m + r] = Y_r[m]: interleaved by ranges of m (contiguous writes per thread -- writing with stride 16 from 16 threads
    // makes every cache line bounce between all of them)
    // real input (the taps): X[n - k] = conj(X[k]), i.e. sub-transform 16 - r is the mirrored conjugate of sub-transform r --
    // nine of the sixteen are computed
    bool real_in = true;
    for (size_t i = 0; i < xi.size() && real_in; ++i) real_in = xi[i] == 0.0;
    auto weave = [&](size_t part) {
        const size_t m0 = M * part / R, m1 = M * (part + 1) / R;
        for (size_t m = m0; m < m1; ++m)
            for (size_t r = 0; r < R; ++r) {
                if (real_in && r > R / 2) { re[R * m + r] = Yr[R - r][M - 1 - m]; im[R * m + r] = -Yi[R - r][M - 1 - m]; }
                else { re[R * m + r] = Yr[r][m]; im[R * m + r] = Yi[r][m]; }
            }
    };
    auto run = [&](auto &fn, size_t count) {
        const size_t nt = std::min(nthr, count);
        if (nt <= 1) { for (size_t r = 0; r < count; ++r) fn(r); return; }
        std::vector<std::thread> th;
        for (size_t t = 0; t < nt; ++t)
            th.emplace_back([=, &fn] { for (size_t r = t; r < count; r += nt) fn(r); });
        for (auto &t : th) t.join();
    };
    run(sub, real_in ? R / 2 + 1 : R);
    run(weave, R);
}

void host_fft_f64(std::vector<double> &re, std::vector<double> &im) { host_fft(re, im); }     // olslds.hip's spectra

static cpx *upload_cpx(const std::vector<cpx> &h)
{
    cpx *d = nullptr;
    TFX_HIP(hipMalloc((void **)&d, h.size() * sizeof(cpx)));
    TFX_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(cpx), hipMemcpyHostToDevice));
    return d;
}
static std::vector<cpx> twiddles(int64_t n, int64_t count, int64_t step)   // W_n^(step*i), i < count
{
    std::vector<cpx> t((size_t)count);
    for (int64_t i = 0; i < count; ++i) {
        const double a = -2.0 * M_PI * (double)((step * i) % n) / (double)n;
        t[(size_t)i] = make_float2((float)cos(a), (float)sin(a));
    }
    return t;
}

static int64_t envi(const char *name, int64_t dflt) { return env_i64(name, dflt); }      // read once per process (common.h)

// ---- per-device one-time set-up: kernel attributes (the first touch of a kernel loads the library's code object: 10-25 ms)
// and the internal streams with their fork / join events (the first stream of a process costs ~5 ms each).  The first call
// on a device runs this on a helper thread WHILE the calling thread computes the filter's spectrum on the host.
typedef void (*colf_t)(const float *, cpx *, const cpx *, OlsGeom, int64_t);
typedef void (*coli_t)(const cpx *, float *, const cpx *, OlsGeom, int64_t);
typedef void (*row_t)(cpx *, const cpx *, const cpx *, const cpx *, const cpx *, const cpx *, const cpx *, const cpx *,
                      int64_t, int, int64_t);
static const colf_t colf_tab[2][5] = {
    {ols_col_fwd16_kernel<2, 0>, ols_col_fwd16_kernel<2, 1>, ols_col_fwd16_kernel<2, 2>, ols_col_fwd16_kernel<2, 3>, ols_col_fwd16_kernel<2, 4>},
    {ols_col_fwd16_kernel<1, 0>, ols_col_fwd16_kernel<1, 1>, ols_col_fwd16_kernel<1, 2>, ols_col_fwd16_kernel<1, 3>, ols_col_fwd16_kernel<1, 4>}};
static const coli_t coli_tab[2][5] = {
    {ols_col_inv16_kernel<2, 0>, ols_col_inv16_kernel<2, 1>, ols_col_inv16_kernel<2, 2>, ols_col_inv16_kernel<2, 3>, ols_col_inv16_kernel<2, 4>},
    {ols_col_inv16_kernel<1, 0>, ols_col_inv16_kernel<1, 1>, ols_col_inv16_kernel<1, 2>, ols_col_inv16_kernel<1, 3>, ols_col_inv16_kernel<1, 4>}};
static const row_t row_tab[8] = {ols_row4096_kernel<0, 0>, ols_row4096_kernel<1, 0>, ols_row4096_kernel<0, 1>, ols_row4096_kernel<1, 1>,
                                 ols_row4096_kernel<0, 2>, ols_row4096_kernel<1, 2>, ols_row4096_kernel<0, 3>, ols_row4096_kernel<1, 3>};
typedef void (*colsos_t)(const float *, cpx *, const cpx *, OlsGeom, int64_t, SosFuse);
#define TFX_SOSF_ROW(TAPS_, UNIT_) {ols_col_fwd16_sos_kernel<1, TAPS_, UNIT_>, ols_col_fwd16_sos_kernel<2, TAPS_, UNIT_>, ols_col_fwd16_sos_kernel<3, TAPS_, UNIT_>, \
                                    ols_col_fwd16_sos_kernel<4, TAPS_, UNIT_>, ols_col_fwd16_sos_kernel<5, TAPS_, UNIT_>, ols_col_fwd16_sos_kernel<6, TAPS_, UNIT_>, \
                                    ols_col_fwd16_sos_kernel<7, TAPS_, UNIT_>, ols_col_fwd16_sos_kernel<8, TAPS_, UNIT_>}
static const colsos_t colsos_tab[4][SOSF_MAXK] = {TFX_SOSF_ROW(false, false), TFX_SOSF_ROW(true, false), TFX_SOSF_ROW(false, true), TFX_SOSF_ROW(true, true)};
constexpr int MAXL = 8;
struct Lanes {                      // internal streams and fork/join events of one device
    hipStream_t stream[MAXL] = {};
    hipEvent_t fork = nullptr, join[MAXL] = {};
};
static Lanes lanes_tab[TFX_MAX_DEVICES];
static bool attr_tab[TFX_MAX_DEVICES] = {};
// The lanes and their fork / join events are shared by every caller on the device.  A waiting stream takes the event's MOST
// RECENT record, so "record the fork event on my stream, make the lanes wait for it" must not interleave with another host
// thread's fork (its lanes would wait for the wrong stream's point); the join is safe either way (a later record on the same
// lane is a superset) but takes the same lock.  Only these two short sections are serialised, not the launches.
static std::mutex lane_mu;
constexpr size_t OLS_SHM_COL = (size_t)(OLS_N1 * OLS_CB + 256) * sizeof(cpx);

static std::mutex g_attr_mu;
static void ols_set_attributes(int dev)
{
    std::lock_guard<std::mutex> lk(g_attr_mu);            // not the plan lock: this runs beside the spectrum computation
    if (attr_tab[dev]) return;
    for (int a = 0; a < 8; ++a)
        TFX_HIP(hipFuncSetAttribute((const void *)row_tab[a], hipFuncAttributeMaxDynamicSharedMemorySize, (int)((4096 + 256 + 512) * sizeof(cpx))));
    TFX_HIP(hipFuncSetAttribute((const void *)ols_row8192_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((4096 + 256 + 512) * sizeof(cpx))));
    TFX_HIP(hipFuncSetAttribute((const void *)ols_row8192_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((4096 + 256 + 512) * sizeof(cpx))));
    TFX_HIP(hipFuncSetAttribute((const void *)ols_rowspec8192_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((4096 + 256 + 512) * sizeof(cpx))));
    TFX_HIP(hipFuncSetAttribute((const void *)ols_rowspec4096_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((4096 + 256 + 512) * sizeof(cpx))));
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 5; ++b) {
            TFX_HIP(hipFuncSetAttribute((const void *)colf_tab[a][b], hipFuncAttributeMaxDynamicSharedMemorySize, (int)OLS_SHM_COL));
            TFX_HIP(hipFuncSetAttribute((const void *)coli_tab[a][b], hipFuncAttributeMaxDynamicSharedMemorySize, (int)OLS_SHM_COL));
        }
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < SOSF_MAXK; ++b)
            TFX_HIP(hipFuncSetAttribute((const void *)colsos_tab[a][b], hipFuncAttributeMaxDynamicSharedMemorySize, (int)OLS_SHM_SOSF + 16384));
    attr_tab[dev] = true;
}
static void ols_make_lanes(int dev, int nlanes)           // lanes are created when first used
{
    std::lock_guard<std::mutex> fl(lane_mu);
    Lanes &ln = lanes_tab[dev];
    for (int i = 0; i < nlanes; ++i)
        if (!ln.stream[i]) {
            TFX_HIP(hipStreamCreateWithFlags(&ln.stream[i], hipStreamNonBlocking));
            TFX_HIP(hipEventCreateWithFlags(&ln.join[i], hipEventDisableTiming));
        }
    if (!ln.fork) TFX_HIP(hipEventCreateWithFlags(&ln.fork, hipEventDisableTiming));
}

static std::mutex g_warm_mu;
static std::shared_future<void> g_warm[TFX_MAX_DEVICES];
void olsnative_prewarm()
{
    const int dev = current_device();
    // the lanes of both pipelines (plain: 2; recursion in pass A: 3) -- a stream created later, behind gigabytes of workspace
    // allocations, cost 64 ms in the first call (profiles/r05_experiments.txt section 9)
    const int want_lanes = (int)std::min<int64_t>(MAXL, std::max<int64_t>(1, std::max(envi("TFX_OLS_STREAMS", 2), envi("TFX_OLS_SOS_STREAMS", 3))));
    std::lock_guard<std::mutex> lk(g_warm_mu);
    if (g_warm[dev].valid()) return;                         // started before (its result, or error, is kept)
    if (attr_tab[dev] && (want_lanes <= 1 || lanes_tab[dev].stream[want_lanes - 1])) return;
    g_warm[dev] = std::async(std::launch::async, [dev, want_lanes] {
        TFX_HIP(hipSetDevice(dev));
        ols_set_attributes(dev);
        if (want_lanes > 1) ols_make_lanes(dev, want_lanes);
        // the first raw hipMalloc + pageable host-to-device copy of a process set up the runtime's staging path (7-36 ms on the
        // boxes of round 4, TFX_OLS_TRACE): done here once so that the first plan's table upload does not pay for it
        void *p = nullptr;
        std::vector<char> h((size_t)1 << 20, 0);
        if (hipMalloc(&p, h.size()) == hipSuccess) {
            (void)hipMemcpy(p, h.data(), h.size(), hipMemcpyHostToDevice);
            (void)hipFree(p);
        }
    }).share();
}

// Every overlap-save entry waits for a running helper first: its hipMalloc / hipStreamCreate must not land in a stream capture
// the caller starts right after the call (advisor, round 4).  A failure of the helper is reported once, then forgotten.
void olsnative_wait_warm()
{
    const int dev = current_device();
    std::shared_future<void> w;
    {
        std::lock_guard<std::mutex> lk(g_warm_mu);
        w = g_warm[dev];
    }
    if (!w.valid()) return;
    try {
        w.get();
    } catch (...) {
        std::lock_guard<std::mutex> lk(g_warm_mu);
        g_warm[dev] = std::shared_future<void>();
        throw;
    }
}

// TFX_OLS_TRACE=1: host milliseconds of the set-up phases of a call on stderr (where the first call of a process goes)
struct HostTrace {
    bool on;
    std::chrono::steady_clock::time_point t0;
    HostTrace() : on(envi("TFX_OLS_TRACE", 0) != 0), t0(std::chrono::steady_clock::now()) {}
    void mark(const char *what)
    {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[tfx ols] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

static NativePlanPtr get_native_plan(const float *kf, int64_t K, int64_t N, int64_t lead, hipStream_t stream)
{
    // steady state (the same filter call after call, e.g. streaming chunks): one memcmp against the plan
    // used last, no key construction
    NativePlanPtr *last = g_last_plan;
    std::vector<char> *last_key = g_last_key;
    const int dev_ = current_device();
    {
        const std::vector<char> &lk = last_key[dev_];
        const size_t nb = (size_t)K * sizeof(float);
        if (last[dev_] && lk.size() == nb + 2 * sizeof(int64_t) + 1 && memcmp(lk.data(), kf, nb) == 0 &&
            memcmp(lk.data() + nb, &N, sizeof(N)) == 0 && memcmp(lk.data() + nb + sizeof(N), &lead, sizeof(lead)) == 0)
            return last[dev_];
    }
    std::vector<char> key((const char *)kf, (const char *)kf + K * sizeof(float));
    key.insert(key.end(), (const char *)&N, (const char *)&N + sizeof(N));
    key.insert(key.end(), (const char *)&lead, (const char *)&lead + sizeof(lead));
    key.push_back((char)current_device());
    auto it = g_nplans.find(key);
    if (it != g_nplans.end()) { last[dev_] = it->second; last_key[dev_] = key; return it->second; }
    {
        // a new filter costs device allocations, blocking uploads (and, for N = 2^20, two launches): not inside a stream capture
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        TFX_CHECK(!(hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone),
                  "overlap-save: first use of this filter (%lld taps) inside a stream capture -- run it once before capturing "
                  "(its spectrum and tables are uploaded with blocking copies)", (long long)K);
    }
    if (g_nplans.size() > 16) {            // the map lets go; a plan lives on with whoever still holds it
        for (int d = 0; d < TFX_MAX_DEVICES; ++d) last[d].reset();
        g_nplans.clear();
    }
    NativePlanPtr pl = std::make_shared<NativePlan>();
    pl->N = N; pl->K = K; pl->N2 = (int)(N / OLS_N1);
    HostTrace tr;
    const int N2 = pl->N2;
    const bool dev_spectrum = (N2 == 4096 || N2 == 8192) && envi("TFX_OLS_GPU_SPECTRUM", 1) != 0;
    if (!dev_spectrum) {
        // spectrum in float64 on the host: conj(FFT(kf zero-padded)) / N   (_fftconv.py:123-124,131 + irfft scaling)
        std::vector<double> re((size_t)N, 0.0), im((size_t)N, 0.0);
        // `lead` zeros in front of the flipped taps (= trailing zeros of the true impulse response):
        // same convolution, but the causal left padding grows to K-1+lead, which lets the frames start
        // on 128-byte boundaries while the outputs stay unshifted
        for (int64_t i = 0; i < K; ++i) re[(size_t)(lead + i)] = (double)kf[i];
        host_fft(re, im);
        tr.mark("  spectrum: host FFT");
        std::vector<cpx> hp((size_t)N);
        {
            const double inv_n = 1.0 / (double)N;                  // exact: N is a power of two
            auto rows = [&](int lo, int hi) {
                for (int k1 = lo; k1 < hi; ++k1)
                    for (int k2 = 0; k2 < N2; ++k2) {
                        const size_t k = (size_t)k1 + (size_t)OLS_N1 * (size_t)k2;
                        // 8192-point rows keep their bins in the order of ols_row8192_kernel's radix-2 split
                        hp[(size_t)k1 * N2 + (N2 == 8192 ? row8192_at_host(k2) : k2)] = make_float2((float)(re[k] * inv_n), (float)(-im[k] * inv_n));
                    }
            };
            const unsigned hw = std::thread::hardware_concurrency();
            const int nt = N >= (1 << 18) ? (int)std::min<unsigned>(16, hw ? hw : 1) : 1;
            if (nt <= 1) rows(0, OLS_N1);
            else {
                std::vector<std::thread> th;
                for (int t = 0; t < nt; ++t) th.emplace_back(rows, OLS_N1 * t / nt, OLS_N1 * (t + 1) / nt);
                for (auto &t : th) t.join();
            }
        }
        tr.mark("  spectrum: permute");
        pl->Hp = upload_cpx(hp);
        tr.mark("  spectrum: upload");
    }
    {
        // all twiddle tables in ONE allocation and ONE copy (eight hipMalloc + hipMemcpy pairs cost 7-9 ms of the first call)
        std::vector<cpx> ta(256);                        // twA[16 t + a] = W4096^(t a)
        for (int t = 0; t < 16; ++t)
            for (int a2 = 0; a2 < 16; ++a2) {
                const double ang = -2.0 * M_PI * (double)(t * a2) / 4096.0;
                ta[16 * t + a2] = make_float2((float)cos(ang), (float)sin(ang));
            }
        // row-uniform factors: W_N^(64 i) for the 1024-point rows, W_N^(256 i) for the 4096-point rows
        const std::vector<cpx> parts[8] = {twiddles(256, 256, 1), twiddles(N2, N2, 1), twiddles(N, 512, 1), twiddles(N, N / 512, 512),
                                           (N2 >= 4096) ? twiddles(N, N / 256, 256) : twiddles(N, N / 64, 64), ta, twiddles(4096, 64, 64),
                                           twiddles(8192, 256, 1)};
        cpx **slots[8] = {&pl->tw256, &pl->twr, &pl->tlo, &pl->thi, &pl->tu, &pl->t4lo, &pl->t4hi, &pl->w8k};
        std::vector<cpx> all;
        size_t off[8];
        for (int i = 0; i < 8; ++i) {
            off[i] = all.size();
            all.insert(all.end(), parts[i].begin(), parts[i].end());
            all.resize((all.size() + 31) & ~(size_t)31);          // 256-byte aligned sub-tables
        }
        cpx *basep = upload_cpx(all);
        for (int i = 0; i < 8; ++i) *slots[i] = basep + off[i];
        pl->tables = basep;
    }
    tr.mark("  twiddle tables");
    if (dev_spectrum) {
        // N = 2^20: the spectrum is computed by the pipeline's own kernels on the caller's stream (see ols_rowspec4096_kernel)
        TFX_HIP(hipMalloc((void **)&pl->taps_dev, (size_t)K * sizeof(float)));
        TFX_HIP(hipMemcpy(pl->taps_dev, kf, (size_t)K * sizeof(float), hipMemcpyHostToDevice));
        TFX_HIP(hipMalloc((void **)&pl->Hp, (size_t)N * sizeof(cpx)));
        ols_set_attributes(current_device());                 // the column pass needs more than 64 KB of dynamic LDS
        OlsGeom g{};
        g.Tn = K; g.Tout = K; g.F = 1; g.S = N; g.pad_left = lead; g.out_shift = 0; g.nframes = 1;
        g.hist = nullptr; g.H = 0; g.ep_gain = 1.0f; g.ep_scale = 0; g.ep_clamp = 0; g.ep_stat = -1; g.ep_partial = nullptr;
        g.N2 = N2; g.P2 = N2; g.nt = 0;
        hipLaunchKernelGGL(colf_tab[1][0], dim3((unsigned)(N2 / OLS_CB)), dim3(512), OLS_SHM_COL, stream,
                           (const float *)pl->taps_dev, pl->Hp, pl->tw256, g, (int64_t)0);
        TFX_HIP(hipGetLastError());
        if (N2 == 4096)
            hipLaunchKernelGGL(ols_rowspec4096_kernel, dim3(OLS_N1), dim3(256), (size_t)(4096 + 256 + 512) * sizeof(cpx), stream,
                               pl->Hp, pl->tw256, pl->t4lo, pl->tlo, pl->thi, pl->tu, N - 1, N2, (float)(1.0 / (double)N));
        else
            hipLaunchKernelGGL(ols_rowspec8192_kernel, dim3(OLS_N1), dim3(256), (size_t)(4096 + 256 + 512) * sizeof(cpx), stream,
                               pl->Hp, pl->tw256, pl->t4lo, pl->tlo, pl->thi, pl->tu, pl->w8k, N - 1, N2, (float)(1.0 / (double)N));
        TFX_HIP(hipGetLastError());
        TFX_HIP(hipEventCreateWithFlags(&pl->ready, hipEventDisableTiming));
        TFX_HIP(hipEventRecord(pl->ready, stream));
        pl->ready_stream = stream;
        tr.mark("  spectrum: device (2 launches)");
    }
    g_nplans[key] = pl;
    last[dev_] = pl; last_key[dev_] = key;
    return pl;
}

void olsnative_clear()
{
    {
        std::lock_guard<std::mutex> lk(g_np_mu);
        g_nplans.clear();
        for (int d = 0; d < TFX_MAX_DEVICES; ++d) { g_last_plan[d].reset(); g_free_mb[d] = 0; }
    }
    std::lock_guard<std::mutex> lk(g_warm_mu);          // a failed helper is not remembered beyond a clear
    for (int d = 0; d < TFX_MAX_DEVICES; ++d) g_warm[d] = std::shared_future<void>();
}


// block sizes this path implements: N = 256 * N2, N2 in {256, 1024, 4096, 8192}
bool olsnative_supported(int64_t K, int64_t L, int64_t *N_out)
{
    if (envi("TFX_OLS_NATIVE", 1) == 0) return false;
    int64_t N = 0;
    const int64_t lg = envi("TFX_FFT_LOG2N", 0);
    if (lg == 16 || lg == 18 || lg == 20 || lg == 21) N = (int64_t)1 << lg;
    else if (lg != 0) return false;
    else if (K < envi("TFX_OLS_NATIVE_MIN_K", 16)) return false;   // a handful of taps: use the direct kernel / rocFFT
    else if (4 * K <= (1 << 16)) {
        // 8193 ... 16 384 taps (below that the one-launch kernels serve): 2^16 points waste 12-25 % of a block on the overlap.  Rows of
        // 4 M samples and more take the 2^20-point block (16 x 28.8 M: 16 384 taps 2.20 -> 1.93 ms, 12 288: 2.07 -> 1.93, 8193: 1.98
        // -> 1.91), rows of 2 M and more the 2^18-point one from 9000 taps (64 x 2.88 M: 16 384 taps 0.936 -> 0.832, 12 288: 0.880 ->
        // 0.831, 9000: 0.850 -> 0.827); shorter rows are a handful of workgroups either way (profiles/r06_experiments.txt section 9)
        N = 1 << 16;
        if (K > 8192 && L >= 4 * ((int64_t)1 << 20)) N = (int64_t)1 << 20;
        else if (K >= envi("TFX_OLS_N18_MINK", 9000) && L >= ((int64_t)1 << 21)) N = 1 << 18;
    }
    // rows shorter than 4 M samples: 2^18 points -- but from ~22 K taps the 2^20-point block wins there too once a row holds two
    // of them (64 x 2.88 M: 65 536 taps 1.01-1.07 -> 0.86 ms, 32 768 taps 0.92 -> 0.85, 23 000 taps 0.897 -> 0.852; at 22 000 taps
    // 0.820 against 0.855; r05 experiments section 11, r06 section 9)
    else if (2 * K <= (1 << 18) && L < 4 * ((int64_t)1 << 20)) N = (K > envi("TFX_OLS_N20_MINK", 22528) && L >= ((int64_t)1 << 21)) ? ((int64_t)1 << 20) : (1 << 18);
    else if (2 * K <= (1 << 20)) {
        N = (int64_t)1 << 20;                            // long signals: 4x fewer blocks, less overlap
        // 2^21 = 256 x 8192 on signals of at least four such blocks: half the overlap again (TFX_OLS_N21: 0 never)
        if (envi("TFX_OLS_N21", 0) != 0 && K > 16384 && L >= ((int64_t)1 << 23)) N = (int64_t)1 << 21;
    }
    else return false;
    if (N < 2 * K) return false;
    if (L < N) {                                        // signal shorter than one block
        if (lg != 0) return false;
        while (N > (1 << 16) && L < N) N >>= 2;
        if (L < N || N < 2 * K) return false;
    }
    *N_out = N;
    return true;
}

// The cascade-in-pass-A form (ols_col_fwd16_sos_kernel): 4096-point rows, aligned frames, a cascade of at most SOSF_MAXK
// sections whose warm-up fits a row.  `force`: take the 2^20-point block even for rows shorter than one block (tests at
// fixture size; a one-frame launch).
bool sos_unit_rows(const double *sos_host, int64_t K, double (*rows)[5]);     // sos.hip

bool olsnative_sos_supported(int64_t Ksos, int64_t warm, int64_t K, int64_t Tn, int64_t pl, int64_t pr, int force, int64_t *N_out)
{
    if (envi("TFX_OLS_SOS", 1) == 0) return false;
    if (Ksos < 1 || Ksos > SOSF_MAXK || warm < 0 || warm > envi("TFX_OLS_SOS_MAXWARM", 4096)) return false;
    const int64_t L = Tn + pl + pr;
    if (L < K || envi("TFX_OLS_ALIGN", 1) == 0) return false;
    int64_t N = (int64_t)1 << (force == 2 ? 21 : 20);       // force: 1 = the 2^20-point block, 2 = the 2^21-point block, whatever the row length
    if (force) { if (N < 2 * (K + 32)) return false; }
    else {
        if (!olsnative_supported(K, L, &N) || (N != ((int64_t)1 << 20) && N != ((int64_t)1 << 21))) return false;
        // rows of 8192 samples (N = 2^21) halve the warm-up share of the recursion pass: 9.7 against 10.1 ms on the cfg-5 chain
        // (the plain pipeline is 5 % slower at 2^21 and stays at 2^20; profiles/r05_experiments.txt section 8)
        if (N == ((int64_t)1 << 20) && envi("TFX_OLS_SOS_N21", 1) != 0 && L >= ((int64_t)1 << 23) && 2 * (K + 32) <= ((int64_t)1 << 21))
            N = (int64_t)1 << 21;
    }
    // development (profiles/r06_experiments.txt): TFX_OLS_SOS_PROBE_A = 1 runs the recursion pass alone (passes B and C skipped, wrong
    // output), = 22 also takes rows of 16 384 samples (N = 2^22) -- what the warm-up share of such rows would buy pass A'
    if (envi("TFX_OLS_SOS_PROBE_A", 0) == 22 && !force) N = (int64_t)1 << 22;
    if (N_out) *N_out = N;
    return true;
}

// hop and frames per row of a block size on this path (what olsnative_forward computes for itself below)
void olsnative_geometry(int64_t K, int64_t Tn, int64_t pl, int64_t pr, int64_t N, int64_t *S_out, int64_t *F_out)
{
    const int64_t L = Tn + pl + pr, Tout = L - K + 1;
    (void)Tout;
    const bool align = envi("TFX_OLS_ALIGN", 1) != 0;
    const int64_t lead = align ? (32 - (pl % 32)) % 32 : 0;
    int64_t S = N - (K + lead) + 1;
    if (align && S > 64) S -= S % 32;
    if (S_out) *S_out = S;
    // rows that are not whole 128-byte lines shift their frame grid by up to 31 samples (row_shift): one more frame at most
    if (F_out) *F_out = ceil_div(Tout + ((align && (Tn % 32 != 0)) ? 31 : 0), S);
}

void olsnative_forward(const float *x, float *y, int64_t C, int64_t Tn, const float *kf_host, int64_t K,
                       int64_t pl, int64_t pr, int64_t N, hipStream_t stream, const float *hist, int64_t H, const Epilogue *ep,
                       const SosFuseHost *sosf)
{
    HostTrace tr;
    // g_np_mu guards the plan cache, the one-time function attributes and the creation of the internal streams (the three
    // short sections below); the launches themselves are not serialised, so two host threads that drive two streams overlap
    // (each stream has its own scratch slabs; the internal lanes are shared and ordered by the fork / join events)
    OlsGeom g;
    const int64_t L = Tn + pl + pr;
    g.Tn = Tn; g.Tout = L - K + 1; g.pad_left = pl; g.out_shift = 0;
    g.hist = hist; g.H = hist ? H : 0;
    g.ep_gain = ep ? (float)ep->gain : 1.0f; g.ep_scale = ep ? ep->scale : 0; g.ep_clamp = ep ? ep->clamp : 0;
    g.ep_stat = ep ? ep->stat_mode : -1; g.ep_partial = nullptr;
    // 128-byte aligned frames (rows themselves aligned): prepend `lead` zeros to the flipped taps so
    // that the left padding becomes a multiple of 32 samples, and round the hop down to a multiple
    // of 32: every 32-column segment the column passes read or write is then exactly one cache
    // line (misaligned segments straddle two lines: +8 % time on the forward pass, +26 % on the
    // inverse pass, measured)
    // Rows that are not whole lines (Tn % 32 != 0, or x starting inside a line) keep this: row c's frame grid is moved left by
    // sh(c) samples (row_shift) so that its frames start on lines of MEMORY; the column passes address the same way, only the
    // first and the last line of a row are partial.
    int64_t lead = 0;
    const bool align = envi("TFX_OLS_ALIGN", 1) != 0;
    if (align) lead = (32 - (pl % 32)) % 32;
    g.sh_base = (int)(((uintptr_t)x >> 2) & 31);
    g.sh_on = (align && (Tn % 32 != 0 || g.sh_base != 0)) ? 1 : 0;
    g.nf_flag = nullptr;
    g.nf_pair = nullptr;
    g.pad_left = pl + lead;
    g.S = N - (K + lead) + 1;
    if (align && g.S > 64) g.S -= g.S % 32;
    const int dev = current_device();
    // first call on this device: kernel attributes (code-object load) and the internal streams are set up on a helper thread
    // while this thread computes the spectrum (both are tens of milliseconds, one bound by the driver, one by the host's cores);
    // tfx_prewarm() starts the same helper earlier (the Python planner calls it before it merges taps)
    olsnative_prewarm();
    NativePlanPtr plan;
    std::exception_ptr plan_err;
    try {
        std::lock_guard<std::mutex> lk(g_np_mu);
        plan = get_native_plan(kf_host, K, N, lead, stream);
    } catch (...) { plan_err = std::current_exception(); }
    {
        std::shared_future<void> w;
        {
            std::lock_guard<std::mutex> lk(g_warm_mu);
            w = g_warm[dev];
        }
        if (w.valid()) {
            try {
                w.get();                                     // rethrows what the helper threw ...
            } catch (...) {
                std::lock_guard<std::mutex> lk(g_warm_mu);   // ... once: the next call starts the set-up again (advisor, round 4)
                g_warm[dev] = std::shared_future<void>();
                throw;
            }
        }
    }
    if (plan_err) std::rethrow_exception(plan_err);
    if (plan->ready && plan->ready_stream != stream) TFX_HIP(hipStreamWaitEvent(stream, plan->ready, 0));   // spectrum computed on another stream
    tr.mark("plan (spectrum, tables)");
    g.F = ceil_div(g.Tout + g.out_shift + (g.sh_on ? 31 : 0), g.S);
    g.nframes = C * g.F; g.N2 = plan->N2;
    g.P2 = g.N2 + (int)envi("TFX_OLS_PITCH_PAD", 0);
    g.nt = (int)envi("TFX_OLS_NT", 3);
    const int64_t npairs = ceil_div(g.nframes, 2);
    if (!sosf && C > 1 && (g.F & 1)) {     // some pair straddles two signal rows: see ols_col_fwd16_kernel
        g.nf_pair = (int *)scratch("olsn_nf_pair", (size_t)C * sizeof(int), stream);
        TFX_HIP(hipMemsetAsync(g.nf_pair, 0, (size_t)C * sizeof(int), stream));
    }
    if (g.ep_stat >= 0)                   // every (frame, column block) slot is written by exactly one workgroup
        g.ep_partial = (double *)scratch("olsn_ep_partial", (size_t)(g.nframes * (g.N2 / OLS_CB)) * sizeof(double), stream);
    // Slab = the frame pairs one A / B / C launch triple covers; slabs rotate over `nlanes` internal streams, each with its own
    // workspace.  Rounds 1-2 sized slabs for launch efficiency (1 GB: few, large launches).  Round 3 measured the other
    // regime: when the LIVE workspace (slab x lanes) fits the 256 MB Infinity Cache with room for the streaming signal, passes
    // B and C find the slab pass A / B just wrote in the cache instead of in HBM and the whole step gains 8-11 % despite
    // the smaller launches -- cfg 4: 9.4-9.5 ms at 3 x 1 GB, 8.4-8.5 ms at 2 x 64 MB; 48 MB x 3 is as good, 4 lanes or
    // >= 128 MB slabs are not (profiles/r03_experiments.txt).  Default: 64 MB slabs on two lanes.
    // Cascade in pass A: one workgroup per frame pair lives for a whole frame (190-320 column blocks), so a launch needs
    // hundreds of pairs to fill the chip -- slabs of 320 pairs (2.5 GB; N = 2^21: 240 pairs, 3.75 GB) on three lanes
    // (profiles/r05_experiments.txt).
    int nlanes = (int)(sosf ? envi("TFX_OLS_SOS_STREAMS", 3) : envi("TFX_OLS_STREAMS", 2));
    if (nlanes < 1) nlanes = 1;
    if (nlanes > MAXL) nlanes = MAXL;
    const int64_t pair_bytes = (int64_t)OLS_N1 * g.P2 * (int64_t)sizeof(cpx);
    int64_t slab = sosf ? envi("TFX_OLS_SOS_PAIRS", 0) : envi("TFX_OLS_PAIRS_PER_SLAB", 0);
    if (slab <= 0) {
        // The workspace lives outside PyTorch's caching allocator and is kept between calls (scratch(), released by
        // tfx_clear_caches); TFX_OLS_SLAB_MB bounds a lane's share, never more than 1/8 of the free memory over all lanes.
        int64_t slab_mb = sosf ? envi("TFX_OLS_SOS_SLAB_MB", N == ((int64_t)1 << 21) ? 3840 : 2560) : envi("TFX_OLS_SLAB_MB", 64);
        {
            // the cap follows the memory free when the device is first used (and again after tfx_clear_caches), not at every call:
            // a driver query per step costs tens of microseconds, is not allowed while a stream is capturing, and would make the
            // slab geometry depend on the allocator's state of the moment (advisor, round 3)
            static std::mutex cap_mu;
            std::lock_guard<std::mutex> cl(cap_mu);
            int64_t &cap_free_mb = g_free_mb[current_device()];
            if (cap_free_mb == 0) {
                size_t free_b = 0, total_b = 0;
                hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
                const bool capturing = hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
                if (!capturing && hipMemGetInfo(&free_b, &total_b) == hipSuccess) cap_free_mb = std::max<int64_t>(1, (int64_t)(free_b >> 20));
            }
            if (cap_free_mb > 0) {
                const int64_t cap_mb = cap_free_mb / (8 * nlanes);
                if (slab_mb > cap_mb) slab_mb = std::max<int64_t>(cap_mb, 8);
            }
        }
        slab = std::max<int64_t>(1, (slab_mb << 20) / pair_bytes);
        const int64_t per_lane = ceil_div(npairs, nlanes);             // short signals: still one slab per lane
        if (slab > per_lane) slab = std::max<int64_t>(per_lane, 1);
    }
    if (slab > npairs) slab = npairs;
    // Workspaces of all lanes, from the installed allocator (tfx_set_workspace_allocator: PyTorch's caching allocator under the
    // torch module) or hipMalloc.  When memory is short the slab is halved -- fewer frame pairs per launch, same results --
    // down to 8 pairs; below that the call fails with "out of device memory".
    static const char *lane_tags[8] = {"olsn_T", "olsn_T2", "olsn_T3", "olsn_T4", "olsn_T5", "olsn_T6", "olsn_T7", "olsn_T8"};
    cpx *Tlane[8] = {};
    for (;;) {
        const int want = (npairs <= slab) ? 1 : nlanes;
        const size_t bytes = (size_t)slab * (size_t)OLS_N1 * (size_t)g.P2 * sizeof(cpx);
        int got = 0;
        for (; got < want; ++got)
            if (!(Tlane[got] = (cpx *)scratch_try(lane_tags[got], bytes, stream))) break;
        if (got == want) break;
        if (slab <= 8) { (void)scratch(lane_tags[got], bytes, stream); break; }          // throws with the allocator's name in the message
        slab = std::max<int64_t>(8, slab / 2);
    }
    cpx *T = Tlane[0];
    tr.mark("workspaces");
    const size_t shm_col = OLS_SHM_COL;
    const size_t shm_row = (size_t)(g.N2 * 5) * sizeof(cpx);
    const int probe = (int)envi("TFX_OLS_PROBE", 0);           // development only (tools/archive/ols_knobs.py)
    const int nbf = envi("TFX_OLS_COL_THREADS", 512) == 512 ? 1 : 2;
    // XCD-aware row map (1) pays when a slab holds many pairs per spectrum row; with cache-sized slabs the plain map is faster
    const int rowmap = (int)envi("TFX_OLS_ROWMAP", slab >= 32 ? 1 : 0);
    const int colv = (probe & 3) ? (probe & 3) : (envi("TFX_OLS_PK", 1) ? 0 : 4);     // 4 = compiler-scheduled butterflies
    const colf_t colf = colf_tab[nbf == 1][colv];
    const coli_t coli = coli_tab[nbf == 1][colv];
    const int xch = (int)envi("TFX_OLS_ROW_XCH", envi("TFX_OLS_PK", 1) ? 3 : 2);
    const row_t rowk = row_tab[(rowmap == 0 ? 0 : 1) + 2 * (xch < 0 || xch > 3 ? 3 : xch)];
    ols_set_attributes(dev);
    tr.mark("function attributes");
    const int ncb = g.N2 / OLS_CB;
    SosFuse sosk{};
    bool sos_unit = false;
    if (sosf) {
        TFX_CHECK((g.N2 == 4096 || g.N2 == 8192 || envi("TFX_OLS_SOS_PROBE_A", 0) == 22) && align && !hist && sosf->K >= 1 && sosf->K <= SOSF_MAXK && sosf->warm >= 0,
                  "olsnative_forward: the cascade cannot run inside the column pass here (olsnative_sos_supported)");
        g.nf_flag = (int *)scratch("olsn_nf_flag", (size_t)g.nframes * sizeof(int), stream);
        sos_unit = envi("TFX_OLS_SOS_UNIT_B0", 1) != 0 && sos_unit_rows(sosf->sos, sosf->K, sosk.co);
        for (int64_t s = 0; s < sosf->K && !sos_unit; ++s) {
            const double *co = sosf->sos + 6 * s;                 // b0 b1 b2 a0 a1 a2; a0 is not used (iir_cpu.cpp:86)
            sosk.co[s][0] = co[0]; sosk.co[s][1] = co[1]; sosk.co[s][2] = co[2]; sosk.co[s][3] = -co[4]; sosk.co[s][4] = -co[5];
        }
        sosk.sections = sosf->sections;
        sosk.prio = (int)envi("TFX_OLS_SOS_PRIO", 1);
        const int64_t warm_dev = envi("TFX_OLS_SOS_WARM", -1);            // development: probe builds of the experiments log (wrong results)
        sosk.warm_blocks = (int)ceil_div(warm_dev >= 0 ? warm_dev : sosf->warm, OLS_CB);
    }
    // Internal streams (TFX_OLS_STREAMS, default 2), slabs rotate over them: while one slab drains the tail of a pass
    // (the last, partially filled round of workgroups) the other slab's pass fills the idle CUs.
    // Fork/join with events on the caller's stream; each lane has its own workspace.
    if (npairs <= slab) nlanes = 1;
    Lanes &ln_ = lanes_tab[dev];
    hipStream_t *lane_stream = ln_.stream;
    hipEvent_t &ev_fork = ln_.fork;
    hipEvent_t *ev_join = ln_.join;
    hipStream_t user_stream = stream;
    if (nlanes > 1) {
        ols_make_lanes(dev, nlanes);
        tr.mark("  lane streams / events");
        std::lock_guard<std::mutex> fl(lane_mu);
        TFX_HIP(hipEventRecord(ev_fork, user_stream));
        for (int i = 0; i < nlanes; ++i) TFX_HIP(hipStreamWaitEvent(lane_stream[i], ev_fork, 0));
    }
    tr.mark("lanes, workspaces, fork");
    int64_t slab_idx = 0;
    // Cascade in pass A: the lanes would otherwise march in step (all in pass A', then all in B, then all in C) -- the first
    // slab of lane i is cut to (i + 1) / nlanes of a slab so that the three kinds of pass meet on the chip
    // development: extra dynamic LDS for the recursion pass = one workgroup per CU, the rest of the CU stays free for passes B / C
    const size_t sos_lds_pad = sosf ? (size_t)envi("TFX_OLS_SOS_LDS_PAD", 0) : 0;
    const int64_t sos_sub = sosf ? envi("TFX_OLS_SOS_SUB_PAIRS", 0) : 0;
    const int stagger = sosf ? (int)envi("TFX_OLS_SOS_STAGGER", 0) : 0;
    for (int64_t p0 = 0, step_pairs = slab; p0 < npairs; p0 += step_pairs, ++slab_idx) {
        step_pairs = (stagger && nlanes > 1 && slab_idx < nlanes) ? std::max<int64_t>(1, slab * (slab_idx + 1) / nlanes) : slab;
        const int64_t np = (npairs - p0 < step_pairs) ? (npairs - p0) : step_pairs;
        const int ln = nlanes > 1 ? (int)(slab_idx % nlanes) : 0;
        hipStream_t stream = nlanes > 1 ? lane_stream[ln] : user_stream;   // shadows the parameter
        cpx *T = Tlane[ln];
        if (sosf) {
            ProfScope ps("ols_col_fwd16_sos_kernel", stream);
            hipLaunchKernelGGL(colsos_tab[(sosf->sections ? 1 : 0) + (sos_unit ? 2 : 0)][sosf->K - 1], dim3((unsigned)np), dim3(512), OLS_SHM_SOSF + sos_lds_pad, stream,
                               x, T, plan->tw256, g, 2 * p0, sosk);
            TFX_HIP(hipGetLastError());
        } else {
            ProfScope ps("ols_col_fwd16_kernel", stream);
            hipLaunchKernelGGL(colf, dim3((unsigned)(np * ncb)), dim3(512 / nbf), shm_col, stream,
                               x, T, plan->tw256, g, 2 * p0);
            TFX_HIP(hipGetLastError());
        }
        if (sosf && envi("TFX_OLS_SOS_PROBE_A", 0) != 0) continue;          // development: pass A' alone
        if (sosf && probe == 0 && sos_sub > 0 && sos_sub < np && g.N2 == 4096) {
            // behind the one wide launch of the recursion pass, passes B and C walk the slab in cache-sized pieces so that pass C
            // finds what pass B just wrote in the Infinity Cache (the regime of the plain pipeline's 64 MB slabs).  Measured
            // slower (10.4-10.5 against 9.7-10.2 ms, profiles/r05_experiments.txt): off by default (TFX_OLS_SOS_SUB_PAIRS)
            for (int64_t q0 = 0; q0 < np; q0 += sos_sub) {
                const int64_t nq = std::min(sos_sub, np - q0);
                cpx *Tq = T + q0 * ((int64_t)OLS_N1 * g.P2);
                {
                    ProfScope ps("ols_row4096_kernel", stream);
                    hipLaunchKernelGGL(row_tab[2 * (xch < 0 || xch > 3 ? 3 : xch)], dim3((unsigned)(nq * OLS_N1)), dim3(256), (size_t)(4096 + 256 + 512) * sizeof(cpx), stream,
                                       Tq, plan->Hp, plan->tw256, plan->t4lo, plan->t4hi, plan->tlo, plan->thi, plan->tu, N - 1, g.P2, nq);
                    TFX_HIP(hipGetLastError());
                }
                {
                    ProfScope ps("ols_col_inv16_kernel", stream);
                    hipLaunchKernelGGL(coli, dim3((unsigned)(nq * ncb)), dim3(512 / nbf), shm_col, stream, Tq, y, plan->tw256, g, 2 * (p0 + q0));
                    TFX_HIP(hipGetLastError());
                }
            }
            continue;
        }
        {
            const int64_t nrows = np * OLS_N1;
            ProfScope ps(g.N2 == 4096 ? "ols_row4096_kernel" : g.N2 == 8192 ? "ols_row8192_kernel" : (g.N2 == 1024 && envi("TFX_OLS_ROW_R4", 0) == 0 ? "ols_row1024_kernel" : "ols_row_kernel"), stream);
            if (g.N2 == 4096 && probe == 0) {
                hipLaunchKernelGGL(rowk, dim3((unsigned)nrows), dim3(256), (size_t)(4096 + 256 + 512) * sizeof(cpx), stream,
                                   T, plan->Hp, plan->tw256, plan->t4lo, plan->t4hi, plan->tlo, plan->thi, plan->tu,
                                   N - 1, g.P2, np);
            } else if (g.N2 == 4096) {
            } else if (g.N2 == 8192) {
                hipLaunchKernelGGL(rowmap == 0 ? ols_row8192_kernel<0> : ols_row8192_kernel<1>, dim3((unsigned)nrows), dim3(256),
                                   (size_t)(4096 + 256 + 512) * sizeof(cpx), stream,
                                   T, plan->Hp, plan->tw256, plan->t4lo, plan->tlo, plan->thi, plan->tu, plan->w8k, N - 1, g.P2, np);
            }
            else if (g.N2 == 1024 && envi("TFX_OLS_ROW_R4", 0) == 0)
                hipLaunchKernelGGL(envi("TFX_OLS_PK", 1) ? ols_row1024_kernel<true> : ols_row1024_kernel<false>, dim3((unsigned)ceil_div(nrows, 4)), dim3(256),
                                   (size_t)(1024 + 4 * (1024 + 64)) * sizeof(cpx), stream,
                                   T, plan->Hp, plan->twr, plan->tlo, plan->thi, plan->tu, nrows, N - 1, g.P2);
            else if (g.N2 == 1024)
                hipLaunchKernelGGL(ols_row_kernel<5>, dim3((unsigned)ceil_div(nrows, 4)), dim3(256), shm_row, stream,
                                   T, plan->Hp, plan->twr, plan->tlo, plan->thi, nrows, g.P2);
            else if (envi("TFX_OLS_PK", 1) != 0 && envi("TFX_OLS_ROW_R4", 0) == 0) {
                const int64_t groups = ceil_div(nrows, 16);
                const int64_t cap = 8 * 256 * 3;                 // a few rounds of resident workgroups: the lane twiddles are loaded once per workgroup
                hipLaunchKernelGGL(ols_row256pk_kernel, dim3((unsigned)std::min(groups, cap)), dim3(256), (size_t)16 * 272 * sizeof(cpx), stream,
                                   T, plan->Hp, plan->tw256, plan->tlo, plan->thi, nrows, g.P2);
            } else
                hipLaunchKernelGGL(ols_row_kernel<4>, dim3((unsigned)ceil_div(nrows, 4)), dim3(256), shm_row, stream,
                                   T, plan->Hp, plan->twr, plan->tlo, plan->thi, nrows, g.P2);
            TFX_HIP(hipGetLastError());
        }
        {
            ProfScope ps("ols_col_inv16_kernel", stream);
            hipLaunchKernelGGL(coli, dim3((unsigned)(np * ncb)), dim3(512 / nbf), shm_col, stream,
                               T, y, plan->tw256, g, 2 * p0);
            TFX_HIP(hipGetLastError());
        }
    }
    tr.mark("launches");
    if (nlanes > 1) {
        std::lock_guard<std::mutex> jl(lane_mu);
        for (int i = 0; i < nlanes; ++i) {
            TFX_HIP(hipEventRecord(ev_join[i], lane_stream[i]));
            TFX_HIP(hipStreamWaitEvent(user_stream, ev_join[i], 0));
        }
    }
    if (sosf) {                           // non-finite values stay in a recursion: see ols_sos_nonfinite_fix_kernel
        const unsigned chunks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(64, g.Tout >> 16));
        hipLaunchKernelGGL(ols_sos_nonfinite_fix_kernel, dim3((unsigned)C, chunks, (unsigned)(1 + (sosf->sections ? sosf->K : 0))), dim3(256), 0,
                           user_stream, y, sosf->sections, g, sosk.warm_blocks);
        TFX_HIP(hipGetLastError());
    }
    if (g.nf_pair) {
        hipLaunchKernelGGL(ols_straddle_fix_kernel, dim3((unsigned)C, (unsigned)std::max<int64_t>(1, std::min<int64_t>(64, g.S >> 12))), dim3(256), 0,
                           user_stream, y, g);
        TFX_HIP(hipGetLastError());
    }
    if (g.ep_stat >= 0)                   // after the join: all partials are in
        stat_finish(g.ep_partial, ep->per_row ? C : 1, (ep->per_row ? g.F : g.nframes) * (g.N2 / OLS_CB), g.ep_stat,
                    ep->stat_out, user_stream);
}

}  // namespace tfx
