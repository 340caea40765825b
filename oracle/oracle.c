/*
 * oracle/oracle.c -- CPU restatement of the torchfx.filter hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library, and only as the checker /
 * reported CPU baseline -- never as the thing measured or shipped.  The product
 * (torchfx_amd + libtorchfx_hip.so) never links, imports or calls it.
 *
 * Pinned against the reference: oracle/make_golden.py (run in the build
 * container, where /root/reference is importable) checks every function below
 * against the real reference (its compiled CPU extension oracle/_ref and its
 * Python FIR / fft_conv1d) and commits the vectors under tests/golden/;
 * tests/test_oracle_golden.py re-checks the oracle against those vectors.
 *
 * Plain C11, no dependencies.  Arithmetic order follows the cited reference
 * lines; citations are relative to /root/reference/.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Channel loops are OpenMP-parallel (tests want them fast); bench.py's cpu_baseline pins the
 * thread count explicitly so that "cores" in its report is what actually ran. */
int oracle_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}


/* ---------------------------------------------------------------------------
 * Fused K-section Direct-Form-I SOS cascade, float64.
 * Follows src/torchfx/_csrc/cpu/iir_cpu.cpp:64-159 (sos_forward_cpu):
 *   for each channel (OpenMP, :106) / for each sample (:132) / for each
 *   section (:135-144):
 *     yn = b0*val + b1*sx0 + b2*sx1 - a1*sy0 - a2*sy1 ; shift state ; val = yn
 * sos is [K,6] = [b0,b1,b2,a0,a1,a2]; a0 is ignored (:86).
 * state_x/state_y are [K,C,2] = {v[n-1], v[n-2]} and are UPDATED IN PLACE
 * (the caller passes clones, as the reference does at :72-73).
 * y_sections (optional, may be NULL): [K,C,T] output of every section, used
 * for the section-by-section parity checks.
 * ------------------------------------------------------------------------- */
void oracle_sos_df1_f64(const double *x, double *y, int64_t C, int64_t T,
                        const double *sos, int64_t K,
                        double *state_x, double *state_y,
                        double *y_sections)
{
#pragma omp parallel for schedule(static) if (C > 1)
    for (int64_t c = 0; c < C; ++c) {
        double *sx0 = (double *)malloc(sizeof(double) * 4 * (size_t)(K > 0 ? K : 1));
        double *sx1 = sx0 + K, *sy0 = sx1 + K, *sy1 = sy0 + K;
        for (int64_t s = 0; s < K; ++s) {
            sx0[s] = state_x[(s * C + c) * 2 + 0];
            sx1[s] = state_x[(s * C + c) * 2 + 1];
            sy0[s] = state_y[(s * C + c) * 2 + 0];
            sy1[s] = state_y[(s * C + c) * 2 + 1];
        }
        const double *xc = x + c * T;
        double *yc = y + c * T;
        for (int64_t n = 0; n < T; ++n) {
            double val = xc[n];
            for (int64_t s = 0; s < K; ++s) {
                const double *co = sos + s * 6;
                double yn = co[0] * val + co[1] * sx0[s] + co[2] * sx1[s]
                          - co[4] * sy0[s] - co[5] * sy1[s];
                sx1[s] = sx0[s];
                sx0[s] = val;
                sy1[s] = sy0[s];
                sy0[s] = yn;
                val = yn;
                if (y_sections) y_sections[(s * C + c) * T + n] = yn;
            }
            yc[n] = val;
        }
        for (int64_t s = 0; s < K; ++s) {
            state_x[(s * C + c) * 2 + 0] = sx0[s];
            state_x[(s * C + c) * 2 + 1] = sx1[s];
            state_y[(s * C + c) * 2 + 0] = sy0[s];
            state_y[(s * C + c) * 2 + 1] = sy1[s];
        }
        free(sx0);
    }
}

/* ---------------------------------------------------------------------------
 * The whole reference IIR call path for float32 signals:
 *   upcast  x.to(float64)             src/torchfx/_ops.py:149
 *   filter  sos_forward_cpu           src/torchfx/_csrc/cpu/iir_cpu.cpp:64-159
 *   downcast out.to(x.dtype)          src/torchfx/filter/iir.py:176
 * y_sections (optional) is float64 [K,C,T] (pre-downcast section outputs).
 * ------------------------------------------------------------------------- */
void oracle_sos_forward_f32(const float *x, float *y, int64_t C, int64_t T,
                            const double *sos, int64_t K,
                            double *state_x, double *state_y,
                            double *y_sections)
{
    double *xd = (double *)malloc(sizeof(double) * (size_t)(C * T > 0 ? C * T : 1));
    double *yd = (double *)malloc(sizeof(double) * (size_t)(C * T > 0 ? C * T : 1));
    for (int64_t i = 0; i < C * T; ++i) xd[i] = (double)x[i];
    oracle_sos_df1_f64(xd, yd, C, T, sos, K, state_x, state_y, y_sections);
    for (int64_t i = 0; i < C * T; ++i) y[i] = (float)yd[i];
    free(xd);
    free(yd);
}

/* ---------------------------------------------------------------------------
 * Single biquad, DF1, float64: src/torchfx/_csrc/cpu/iir_cpu.cpp:10-62
 * (biquad_forward_cpu).  b = [b0,b1,b2]; state [C,2] updated in place.
 * ------------------------------------------------------------------------- */
void oracle_biquad_df1_f64(const double *x, double *y, int64_t C, int64_t T,
                           const double *b, double a1, double a2,
                           double *state_x, double *state_y)
{
#pragma omp parallel for schedule(static) if (C > 1)
    for (int64_t c = 0; c < C; ++c) {
        double sx0 = state_x[c * 2], sx1 = state_x[c * 2 + 1];
        double sy0 = state_y[c * 2], sy1 = state_y[c * 2 + 1];
        for (int64_t n = 0; n < T; ++n) {
            double xn = x[c * T + n];
            double yn = b[0] * xn + b[1] * sx0 + b[2] * sx1 - a1 * sy0 - a2 * sy1;
            y[c * T + n] = yn;
            sx1 = sx0; sx0 = xn; sy1 = sy0; sy0 = yn;
        }
        state_x[c * 2] = sx0; state_x[c * 2 + 1] = sx1;
        state_y[c * 2] = sy0; state_y[c * 2 + 1] = sy1;
    }
}

/* ---------------------------------------------------------------------------
 * Direct FIR, the conv_mode="direct" branch of FIR.forward
 * (src/torchfx/filter/fir.py:556-568): left-pad K-1 zeros, then depthwise
 * cross-correlation with the stored FLIPPED kernel (fir.py:516-518):
 *     y[c,n] = sum_{j=0}^{K-1} kernel[j] * xpad[c, n+j],  xpad[m] = x[m-(K-1)]
 * which equals lfilter(b, [1], x).  Arithmetic in the input dtype (f32 here;
 * F.conv1d's internal summation order is unspecified, the reference's own
 * tests pin it only to 1e-4: tests/test_fir.py:79-90).
 * ------------------------------------------------------------------------- */
void oracle_fir_direct_f32(const float *x, float *y, int64_t C, int64_t T,
                           const float *kernel, int64_t K)
{
#pragma omp parallel for schedule(static) if (C > 1)
    for (int64_t c = 0; c < C; ++c) {
        const float *xc = x + c * T;
        float *yc = y + c * T;
        for (int64_t n = 0; n < T; ++n) {
            float acc = 0.0f;
            /* xpad index n+j  ->  x index n+j-(K-1) */
            int64_t j0 = (K - 1) - n;
            if (j0 < 0) j0 = 0;
            for (int64_t j = j0; j < K; ++j)
                acc += kernel[j] * xc[n + j - (K - 1)];
            yc[n] = acc;
        }
    }
}

void oracle_fir_direct_f64(const double *x, double *y, int64_t C, int64_t T,
                           const double *kernel, int64_t K)
{
#pragma omp parallel for schedule(static) if (C > 1)
    for (int64_t c = 0; c < C; ++c) {
        const double *xc = x + c * T;
        double *yc = y + c * T;
        for (int64_t n = 0; n < T; ++n) {
            double acc = 0.0;
            int64_t j0 = (K - 1) - n;
            if (j0 < 0) j0 = 0;
            for (int64_t j = j0; j < K; ++j)
                acc += kernel[j] * xc[n + j - (K - 1)];
            yc[n] = acc;
        }
    }
}

/* ---------------------------------------------------------------------------
 * Delay line (kept only because the ext must export it; out of hot-path scope):
 * src/torchfx/_csrc/cpu/delay_cpu.cpp:17-41: y = x + (mix*decay)*x[n-D], n>=D.
 * ------------------------------------------------------------------------- */
void oracle_delay_line_f32(const float *x, float *y, int64_t C, int64_t T,
                           int64_t delay, float coeff)
{
    for (int64_t c = 0; c < C; ++c)
        for (int64_t n = 0; n < T; ++n)
            y[c * T + n] = x[c * T + n] + (n >= delay ? coeff * x[c * T + n - delay] : 0.0f);
}
