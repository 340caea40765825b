// effects.hip -- the elementwise effects that sit between filters in a pipeline:
//   Gain       src/torchfx/effect.py:261-383   y = x * g  (+ clamp to [-1, 1])
//   Normalize  src/torchfx/effect.py:386-531 and its strategies :678-790
//              peak:        y = max|x| > 0 ? x / max|x| * peak : x          (global or per row)
//              rms:         y = rms    > 0 ? x / rms    * peak : x,  rms = sqrt(mean(x^2))
// All of them are pure streaming passes (HBM bound, 8 B/sample for the apply pass, 4 B/sample for the
// reduction).  One-shot grids -- one 256-thread workgroup per 16 KiB, no grid-stride loop: the
// hardware dispatch order keeps concurrently running workgroups on neighbouring addresses, which
// measures 10-20 % faster than persistent loops on this part (tools/ubench/copy_bw.hip).
// The statistics stay on the device (no host sync: the reference's `if max_val > 0` is a blocking
// .item()); the apply kernel reads them and handles the all-zero case itself.
#include "common.h"
#include "epilogue.h"
#include "../../include/torchfx_hip.h"

namespace tfx {

constexpr int EFX_THREADS = 256;
constexpr int EFX_U = 1;                                   // 16-byte vectors per thread (1: short-lived waves stream best, stream_copy2.hip)

template <typename T> struct Vec16;
template <> struct Vec16<float> { typedef float4 type; static constexpr int N = 4; };
template <> struct Vec16<double> { typedef double2 type; static constexpr int N = 2; };

// ---- Gain -------------------------------------------------------------------------------------
template <typename T, bool CLAMP>
__global__ void __launch_bounds__(EFX_THREADS)
gain_kernel(const T *__restrict__ x, T *__restrict__ y, int64_t n, T g)
{
    typedef typename Vec16<T>::type V;
    constexpr int N = Vec16<T>::N;
    const int64_t base = ((int64_t)blockIdx.x * EFX_U * EFX_THREADS + threadIdx.x) * N;
    if (base + (int64_t)(EFX_U - 1) * EFX_THREADS * N + N <= n && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
        V v[EFX_U];
#pragma unroll
        for (int u = 0; u < EFX_U; ++u) v[u] = ldg16_stream<V>(x + base + (int64_t)u * EFX_THREADS * N);
#pragma unroll
        for (int u = 0; u < EFX_U; ++u) {
            T *e = (T *)&v[u];
#pragma unroll
            for (int i = 0; i < N; ++i) {
                e[i] = e[i] * g;
                if (CLAMP) e[i] = clamp_unit(e[i]);
            }
            stg16_stream<V>(y + base + (int64_t)u * EFX_THREADS * N, v[u]);
        }
    } else {
#pragma unroll
        for (int u = 0; u < EFX_U; ++u)
            for (int i = 0; i < N; ++i) {
                const int64_t k = base + (int64_t)u * EFX_THREADS * N + i;
                if (k < n) {
                    T v = x[k] * g;
                    if (CLAMP) v = clamp_unit(v);
                    y[k] = v;
                }
            }
    }
}

// ---- statistics ---------------------------------------------------------------------------------
// Two deterministic stages, no atomics: every workgroup reduces EFX_RT consecutive tiles (128 KiB)
// to one float64 partial, a second kernel reduces a row's partials in a fixed order.
//   MODE 0  max|x|      -- carried as the BIT PATTERN of a non-negative double: such patterns order
//                          like unsigned integers and NaN sorts above inf, so a NaN anywhere wins,
//                          like torch.max
//   MODE 1  sum x^2     -- float64 accumulation
constexpr int EFX_RT = 8;

template <typename T, int MODE>
__global__ void __launch_bounds__(EFX_THREADS)
reduce_kernel(const T *__restrict__ x, int64_t T_, int64_t groups, double *__restrict__ partial)
{
    typedef typename Vec16<T>::type V;
    constexpr int N = Vec16<T>::N;
    constexpr int64_t TILE = (int64_t)EFX_U * EFX_THREADS * N;
    __shared__ double wred[EFX_THREADS / 64];
    const int64_t row = blockIdx.x / groups, grp = blockIdx.x % groups;
    const T *xr = x + row * T_;
    double acc = red_init<MODE>();
    const bool aligned = ((uintptr_t)xr & 15) == 0;
#pragma unroll 1
    for (int rt = 0; rt < EFX_RT; ++rt) {
        const int64_t t0 = (grp * EFX_RT + rt) * TILE;
        if (t0 >= T_) break;
        const int64_t base = t0 + (int64_t)threadIdx.x * N;
        if (aligned && t0 + TILE <= T_) {
            V v[EFX_U];
#pragma unroll
            for (int u = 0; u < EFX_U; ++u) v[u] = *(const V *)(xr + base + (int64_t)u * EFX_THREADS * N);
#pragma unroll
            for (int u = 0; u < EFX_U; ++u) {
                const T *e = (const T *)&v[u];
#pragma unroll
                for (int i = 0; i < N; ++i) acc = red_comb<MODE>(acc, red_elem<MODE>((double)e[i]));
            }
        } else {
#pragma unroll
            for (int u = 0; u < EFX_U; ++u)
                for (int i = 0; i < N; ++i) {
                    const int64_t k = base + (int64_t)u * EFX_THREADS * N + i;
                    if (k < T_) acc = red_comb<MODE>(acc, red_elem<MODE>((double)xr[k]));
                }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc = red_comb<MODE>(acc, __shfl_xor(acc, off));
    if ((threadIdx.x & 63) == 0) wred[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
        partial[blockIdx.x] = red_comb<MODE>(red_comb<MODE>(wred[0], wred[1]), red_comb<MODE>(wred[2], wred[3]));
}

template <int MODE>
__global__ void __launch_bounds__(1024) reduce_finish_kernel(const double *__restrict__ partial, int64_t groups,
                                                             double *__restrict__ out)
{
    // one workgroup per row: strided partials, then a fixed tree
    __shared__ double sh[1024];
    const double *p = partial + (int64_t)blockIdx.x * groups;
    double s = red_init<MODE>();
    for (int64_t i = threadIdx.x; i < groups; i += 1024) s = red_comb<MODE>(s, p[i]);
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int off = 512; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] = red_comb<MODE>(sh[threadIdx.x], sh[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = sh[0];
}

// ---- Normalize: apply ---------------------------------------------------------------------------
// stat: float64 per row (or one global value): max|x| (MODE 0) or the sum of squares (MODE 1, n =
// elements behind each stat).  y = s > 0 ? x / s * peak : x, evaluated
// in the signal dtype in the reference's order (divide, then multiply).
template <typename T, int MODE>
__global__ void __launch_bounds__(EFX_THREADS)
normalize_apply_kernel(const T *__restrict__ x, T *__restrict__ y, int64_t T_, int64_t tiles,
                       const void *__restrict__ stat, int per_row, double n_per_stat, T peak)
{
    typedef typename Vec16<T>::type V;
    constexpr int N = Vec16<T>::N;
    const int64_t row = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const double st = ((const double *)stat)[per_row ? row : 0];
    const T s = MODE == 0 ? (T)st : (T)sqrt(st / n_per_stat);       // max|x| converts back exactly
    const bool on = s > (T)0;                    // false for 0 and for NaN (the reference's `if x > 0`)
    const T *xr = x + row * T_;
    T *yr = y + row * T_;
    const int64_t base = (tile * EFX_U * EFX_THREADS + threadIdx.x) * N;
    if (base + (int64_t)(EFX_U - 1) * EFX_THREADS * N + N <= T_ && (((uintptr_t)xr | (uintptr_t)yr) & 15) == 0) {
        V v[EFX_U];
#pragma unroll
        for (int u = 0; u < EFX_U; ++u) v[u] = ldg16_stream<V>(xr + base + (int64_t)u * EFX_THREADS * N);
#pragma unroll
        for (int u = 0; u < EFX_U; ++u) {
            T *e = (T *)&v[u];
            if (on) {
#pragma unroll
                for (int i = 0; i < N; ++i) e[i] = (e[i] / s) * peak;
            }
            stg16_stream<V>(yr + base + (int64_t)u * EFX_THREADS * N, v[u]);
        }
    } else {
#pragma unroll
        for (int u = 0; u < EFX_U; ++u)
            for (int i = 0; i < N; ++i) {
                const int64_t k = base + (int64_t)u * EFX_THREADS * N + i;
                if (k < T_) yr[k] = on ? (xr[k] / s) * peak : xr[k];
            }
    }
}

// ---- branch sum (`+`) -----------------------------------------------------------------------------
// y = sum_i x_i in list order (zeros_like + in-place adds, __base.py:1022-1026): one pass over the N
// branch outputs, one 16-byte load per input in flight per thread, one-shot grid.
constexpr int SUM_MAX = 16;
struct SumArgs {
    const void *p[SUM_MAX];
    int n;
};
template <typename T>
__global__ void __launch_bounds__(EFX_THREADS) sum_kernel(SumArgs a, T *__restrict__ y, int64_t total, int aligned)
{
    typedef typename Vec16<T>::type V;
    constexpr int N = Vec16<T>::N;
    const int64_t base = ((int64_t)blockIdx.x * EFX_THREADS + threadIdx.x) * N;
    if (base >= total) return;
    if (aligned && base + N <= total) {
        V v[SUM_MAX];
#pragma unroll
        for (int i = 0; i < SUM_MAX; ++i)
            if (i < a.n) v[i] = ldg16_stream<V>((const T *)a.p[i] + base);
        V acc;
        T *ac = (T *)&acc;
#pragma unroll
        for (int e = 0; e < N; ++e) ac[e] = (T)0;
#pragma unroll
        for (int i = 0; i < SUM_MAX; ++i)
            if (i < a.n) {
                const T *e = (const T *)&v[i];
#pragma unroll
                for (int k = 0; k < N; ++k) ac[k] += e[k];
            }
        stg16_stream<V>(y + base, acc);
    } else {
        for (int k = 0; k < N && base + k < total; ++k) {
            T acc = (T)0;
            for (int i = 0; i < a.n; ++i) acc += ((const T *)a.p[i])[base + k];
            y[base + k] = acc;
        }
    }
}

void sum_forward(const void *const *xs_host, int n, void *y, int dtype, int64_t numel, hipStream_t stream)
{
    TFX_CHECK(n >= 1 && n <= SUM_MAX, "sum_forward: between 1 and %d inputs", SUM_MAX);
    TFX_CHECK(dtype == TFX_F32 || dtype == TFX_F64, "sum_forward: bad dtype");
    if (numel == 0) return;
    TFX_CHECK(numel > 0 && xs_host && y, "sum_forward: null pointer or negative size");
    for (int i = 0; i < n; ++i) TFX_CHECK(xs_host[i], "sum_forward: null input %d", i);
    SumArgs a;
    a.n = n;
    uintptr_t bits = (uintptr_t)y;
    for (int i = 0; i < n; ++i) {
        a.p[i] = xs_host[i];
        bits |= (uintptr_t)xs_host[i];
    }
    const int per_thread = dtype == TFX_F32 ? 4 : 2;
    const int64_t grid = ceil_div(numel, (int64_t)EFX_THREADS * per_thread);
    TFX_CHECK(grid < (1ll << 31), "sum_forward: grid too large");
    ProfScope ps("sum_kernel", stream);
    if (dtype == TFX_F32)
        hipLaunchKernelGGL(sum_kernel<float>, dim3((unsigned)grid), dim3(EFX_THREADS), 0, stream, a, (float *)y, numel,
                           (int)((bits & 15) == 0));
    else
        hipLaunchKernelGGL(sum_kernel<double>, dim3((unsigned)grid), dim3(EFX_THREADS), 0, stream, a, (double *)y, numel,
                           (int)((bits & 15) == 0));
    TFX_HIP(hipGetLastError());
}

// ---- delay line -----------------------------------------------------------------------------------
// y[n] = x[n] + coeff * x[n - D]  (src/torchfx/_csrc/cpu/delay_cpu.cpp:17-41), row-tiled one-shot grid;
// the delayed tap is a second, shifted read of the same row (served by the caches).
template <typename T>
__global__ void __launch_bounds__(EFX_THREADS)
delay_line_kernel(const T *__restrict__ x, T *__restrict__ y, int64_t T_, int64_t tiles, int64_t D, T coeff)
{
    typedef typename Vec16<T>::type V;
    constexpr int N = Vec16<T>::N;
    const int64_t row = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const T *xr = x + row * T_;
    T *yr = y + row * T_;
    const int64_t base = (tile * EFX_U * EFX_THREADS + threadIdx.x) * N;
    const bool vec = (((uintptr_t)xr | (uintptr_t)yr) & 15) == 0;
#pragma unroll
    for (int u = 0; u < EFX_U; ++u) {
        const int64_t n0 = base + (int64_t)u * EFX_THREADS * N;
        if (n0 >= T_) break;
        if (vec && n0 + N <= T_) {
            V v = *(const V *)(xr + n0);
            T *e = (T *)&v;
#pragma unroll
            for (int i = 0; i < N; ++i)
                if (n0 + i >= D) e[i] += coeff * xr[n0 + i - D];
            *(V *)(yr + n0) = v;
        } else {
            for (int i = 0; i < N && n0 + i < T_; ++i) {
                T v = xr[n0 + i];
                if (n0 + i >= D) v += coeff * xr[n0 + i - D];
                yr[n0 + i] = v;
            }
        }
    }
}

void delay_line_forward(const void *x, void *y, int dtype, int64_t C, int64_t T, int64_t delay, double coeff,
                        hipStream_t stream)
{
    TFX_CHECK(dtype == TFX_F32 || dtype == TFX_F64, "delay_line_forward: bad dtype %d", dtype);
    if (C == 0 || T == 0) return;
    TFX_CHECK(C > 0 && T > 0 && x && y, "delay_line_forward: null pointer or negative size");
    const int esz = dtype == TFX_F32 ? 4 : 8;
    const int64_t tiles = ceil_div(T, (int64_t)EFX_U * EFX_THREADS * (16 / esz));
    TFX_CHECK(C * tiles < (1ll << 31), "delay_line_forward: grid too large");
    ProfScope ps("delay_line_kernel", stream);
    if (dtype == TFX_F32)
        hipLaunchKernelGGL(delay_line_kernel<float>, dim3((unsigned)(C * tiles)), dim3(EFX_THREADS), 0, stream,
                           (const float *)x, (float *)y, T, tiles, delay, (float)coeff);
    else
        hipLaunchKernelGGL(delay_line_kernel<double>, dim3((unsigned)(C * tiles)), dim3(EFX_THREADS), 0, stream,
                           (const double *)x, (double *)y, T, tiles, delay, coeff);
    TFX_HIP(hipGetLastError());
}

// ---- host ---------------------------------------------------------------------------------------
static inline int64_t efx_tiles(int64_t T, int esz) { return ceil_div(T, (int64_t)EFX_U * EFX_THREADS * (16 / esz)); }

void gain_forward(const void *x, void *y, int dtype, int64_t n, double gain, int clamp, hipStream_t stream)
{
    TFX_CHECK(dtype == TFX_F32 || dtype == TFX_F64, "gain_forward: bad dtype %d", dtype);
    if (n == 0) return;
    TFX_CHECK(n > 0 && x && y, "gain_forward: null pointer or negative size");
    const int esz = dtype == TFX_F32 ? 4 : 8;
    const int64_t tiles = efx_tiles(n, esz);
    TFX_CHECK(tiles < (1ll << 31), "gain_forward: grid too large");
    ProfScope ps("gain_kernel", stream);
    if (dtype == TFX_F32) {
        if (clamp) hipLaunchKernelGGL((gain_kernel<float, true>), dim3((unsigned)tiles), dim3(EFX_THREADS), 0, stream, (const float *)x, (float *)y, n, (float)gain);
        else hipLaunchKernelGGL((gain_kernel<float, false>), dim3((unsigned)tiles), dim3(EFX_THREADS), 0, stream, (const float *)x, (float *)y, n, (float)gain);
    } else {
        if (clamp) hipLaunchKernelGGL((gain_kernel<double, true>), dim3((unsigned)tiles), dim3(EFX_THREADS), 0, stream, (const double *)x, (double *)y, n, gain);
        else hipLaunchKernelGGL((gain_kernel<double, false>), dim3((unsigned)tiles), dim3(EFX_THREADS), 0, stream, (const double *)x, (double *)y, n, gain);
    }
    TFX_HIP(hipGetLastError());
}

// stat_dev: [rows] float64 on the device: max|x| (mode 0) or sum of squares (mode 1) of each row
static void stat_launch(const void *x, int dtype, int64_t rows, int64_t T, int mode, double *stat_dev, hipStream_t stream)
{
    const int esz = dtype == TFX_F32 ? 4 : 8;
    const int64_t groups = ceil_div(efx_tiles(T, esz), (int64_t)EFX_RT);
    TFX_CHECK(rows * groups < (1ll << 31), "stat_forward: grid too large");
    double *partial = (double *)scratch("efx_partial", (size_t)(rows * groups) * sizeof(double), stream);
#define TFX_RED_LAUNCH(TT, MODE_)                                                                              \
    {                                                                                                          \
        {                                                                                                      \
            ProfScope ps(MODE_ == 0 ? "reduce_kernel<absmax>" : "reduce_kernel<sumsq>", stream);               \
            hipLaunchKernelGGL((reduce_kernel<TT, MODE_>), dim3((unsigned)(rows * groups)), dim3(EFX_THREADS), 0, \
                               stream, (const TT *)x, T, groups, partial);                                     \
        }                                                                                                      \
        ProfScope ps("reduce_finish_kernel", stream);                                                          \
        hipLaunchKernelGGL(reduce_finish_kernel<MODE_>, dim3((unsigned)rows), dim3(1024), 0, stream, partial,  \
                           groups, stat_dev);                                                                  \
    }
    if (dtype == TFX_F32) {
        if (mode == 0) TFX_RED_LAUNCH(float, 0) else TFX_RED_LAUNCH(float, 1)
    } else {
        if (mode == 0) TFX_RED_LAUNCH(double, 0) else TFX_RED_LAUNCH(double, 1)
    }
#undef TFX_RED_LAUNCH
    TFX_HIP(hipGetLastError());
}

// out_dev: [rows] float64 -- max|x| or sqrt(mean(x^2)) per row (rows = C when per_row, else 1)
__global__ void stat_decode_kernel(const double *stat, int mode, double n_per_stat, int64_t rows, double *out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    out[i] = mode == 0 ? stat[i] : sqrt(stat[i] / n_per_stat);
}

void stat_forward(const void *x, int dtype, int64_t C, int64_t T, int mode, int per_row, double *out_dev, hipStream_t stream)
{
    TFX_CHECK(dtype == TFX_F32 || dtype == TFX_F64, "stat_forward: bad dtype %d", dtype);
    TFX_CHECK(mode == 0 || mode == 1, "stat_forward: bad mode %d", mode);
    TFX_CHECK(C >= 0 && T >= 0 && out_dev && (x || C * T == 0), "stat_forward: null pointer or negative size");
    const int64_t rows = per_row ? C : 1, len = per_row ? T : C * T;
    if (rows == 0) return;
    double *stat = (double *)scratch("efx_stat", (size_t)rows * 8, stream);
    if (len == 0) {
        TFX_HIP(hipMemsetAsync(out_dev, 0, (size_t)rows * 8, stream));
        return;
    }
    stat_launch(x, dtype, rows, len, mode, stat, stream);
    hipLaunchKernelGGL(stat_decode_kernel, dim3((unsigned)ceil_div(rows, 256)), dim3(256), 0, stream,
                       (const double *)stat, mode, (double)len, rows, out_dev);
    TFX_HIP(hipGetLastError());
}

void stat_finish(const double *partial, int64_t rows, int64_t groups, int mode, double *stat_dev, hipStream_t stream)
{
    ProfScope ps("reduce_finish_kernel", stream);
    if (mode == 0) hipLaunchKernelGGL(reduce_finish_kernel<0>, dim3((unsigned)rows), dim3(1024), 0, stream, partial, groups, stat_dev);
    else hipLaunchKernelGGL(reduce_finish_kernel<1>, dim3((unsigned)rows), dim3(1024), 0, stream, partial, groups, stat_dev);
    TFX_HIP(hipGetLastError());
}

// a producer without a fused epilogue (direct FIR, the rocFFT path, float64 corner cases): the same
// arithmetic as separate streaming passes over its output
void epilogue_as_passes(void *y, int dtype, int64_t C, int64_t T, const Epilogue &ep, hipStream_t stream)
{
    if (C * T == 0) {
        if (ep.stat_mode >= 0 && ep.stat_out) TFX_HIP(hipMemsetAsync(ep.stat_out, 0, (size_t)(ep.per_row ? C : 1) * 8, stream));
        return;
    }
    if (ep.scale || ep.clamp) gain_forward(y, y, dtype, C * T, ep.scale ? ep.gain : 1.0, ep.clamp, stream);
    if (ep.stat_mode >= 0) {
        TFX_CHECK(ep.stat_out, "epilogue: statistic requested without an output buffer");
        stat_launch(y, dtype, ep.per_row ? C : 1, ep.per_row ? T : C * T, ep.stat_mode, ep.stat_out, stream);
    }
}

void normalize_apply_forward(const void *x, void *y, int dtype, int64_t C, int64_t T, int mode, int per_row, double peak,
                             const double *stat, hipStream_t stream);

void normalize_forward(const void *x, void *y, int dtype, int64_t C, int64_t T, int mode, int per_row, double peak,
                       hipStream_t stream)
{
    TFX_CHECK(dtype == TFX_F32 || dtype == TFX_F64, "normalize_forward: bad dtype %d", dtype);
    TFX_CHECK(mode == 0 || mode == 1, "normalize_forward: bad mode %d", mode);
    if (C == 0 || T == 0) return;
    TFX_CHECK(C > 0 && T > 0 && x && y && peak == peak, "normalize_forward: null pointer, negative size or NaN peak");
    const int64_t rows = per_row ? C : 1, len = per_row ? T : C * T;
    double *stat = (double *)scratch("efx_stat", (size_t)rows * 8, stream);
    stat_launch(x, dtype, rows, len, mode, stat, stream);
    normalize_apply_forward(x, y, dtype, C, T, mode, per_row, peak, stat, stream);
}

// the apply pass alone: `stat` = the raw statistic (max|x| or sum x^2, float64, [rows or 1]) a producing
// kernel's epilogue left on the device
void normalize_apply_forward(const void *x, void *y, int dtype, int64_t C, int64_t T, int mode, int per_row, double peak,
                             const double *stat, hipStream_t stream)
{
    TFX_CHECK(dtype == TFX_F32 || dtype == TFX_F64, "normalize_apply: bad dtype %d", dtype);
    TFX_CHECK(mode == 0 || mode == 1, "normalize_apply: bad mode %d", mode);
    if (C == 0 || T == 0) return;
    TFX_CHECK(C > 0 && T > 0 && x && y && stat && peak == peak, "normalize_apply: null pointer, negative size or NaN peak");
    const int64_t rows = per_row ? C : 1, len = per_row ? T : C * T;
    const int esz = dtype == TFX_F32 ? 4 : 8;
    // the apply pass walks the same (rows, len) view, so per-row statistics line up with blockIdx
    const int64_t tiles = efx_tiles(len, esz);
    ProfScope ps("normalize_apply_kernel", stream);
#define TFX_NORM_LAUNCH(TT, MODE_)                                                                          \
    hipLaunchKernelGGL((normalize_apply_kernel<TT, MODE_>), dim3((unsigned)(rows * tiles)), dim3(EFX_THREADS), 0, \
                       stream, (const TT *)x, (TT *)y, len, tiles, (const void *)stat, per_row, (double)len, (TT)peak)
    if (dtype == TFX_F32) {
        if (mode == 0) TFX_NORM_LAUNCH(float, 0); else TFX_NORM_LAUNCH(float, 1);
    } else {
        if (mode == 0) TFX_NORM_LAUNCH(double, 0); else TFX_NORM_LAUNCH(double, 1);
    }
#undef TFX_NORM_LAUNCH
    TFX_HIP(hipGetLastError());
}

}  // namespace tfx
