/*
 * torchfx_hip.h -- C ABI of libtorchfx_hip.so: the MI355X (gfx950) backend for
 * the torchfx.filter hot path.
 *
 * This is the drop-in boundary.  Every entry point takes plain device pointers,
 * sizes and a hipStream_t (as void*); no torch types.  Each one cites the
 * reference interface it replaces (paths relative to the reference repo,
 * matteospanio/torchfx v0.5.3).  The Python host side (torchfx_amd/torchfx_ext.py)
 * binds these with ctypes and re-exposes the reference's own names and
 * signatures (torchfx_ext.biquad_forward / sos_forward / delay_line_forward,
 * FIR.forward, fft_conv1d); INTEGRATION.md shows the stub a reference
 * maintainer would add.
 *
 * Conventions
 *   - all signal buffers are DEVICE memory, row-major [C, T], time-minor,
 *     contiguous (row stride == T);
 *   - coefficient arrays marked HOST are read on the host during the call
 *     (they are O(K), like the reference's `sos_cpu` argument);
 *   - state buffers are DEVICE float64, layout identical to the reference
 *     ([K, C, 2] = {v[n-1], v[n-2]} per section and channel);
 *   - calls are asynchronous on `stream` (no host sync inside), outputs never
 *     alias inputs, inputs are never written;
 *   - return value: 0 = ok, non-zero = error; tfx_last_error() gives the text
 *     (thread-local).  The Python layer turns it into RuntimeError, as
 *     TORCH_CHECK does in the reference (binding.cpp:47,63,78).
 */
#ifndef TORCHFX_HIP_H
#define TORCHFX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *tfx_stream_t; /* hipStream_t */

enum tfx_dtype { TFX_F32 = 0, TFX_F64 = 1 };

/* Arithmetic used INSIDE the IIR kernel (I/O dtypes are independent of it).
 *   TFX_PREC_F64  : float64 recurrences -- what the reference does
 *                   (_ops.py:95,149; iir_cpu.cpp is all double).  Default.
 *   TFX_PREC_F32  : float32 recurrences, float64 only on the host tables.
 *   TFX_PREC_AUTO : F32 when a host-side worst-case error bound derived from
 *                   the SOS is below `TFX_AUTO_F32_BOUND`, else F64.
 */
enum tfx_precision { TFX_PREC_AUTO = 0, TFX_PREC_F32 = 1, TFX_PREC_F64 = 2 };

int tfx_version(void);
const char *tfx_last_error(void);

/* Device / build facts for reports: fills name[len] with the gfx arch string,
 * returns #CUs in *cus (either pointer may be NULL). */
int tfx_device_info(char *name, int len, int *cus);

/* ---------------------------------------------------------------------------
 * tfx_sos_forward -- fused K-section DF1 SOS cascade.
 *
 * Replaces  torchfx_ext.sos_forward(x, sos, sos_cpu, state_x, state_y)
 *           src/torchfx/_csrc/binding.cpp:52-66,88-91
 *           -> sos_forward_cpu  src/torchfx/_csrc/cpu/iir_cpu.cpp:64-159
 *           -> sos_forward_cuda src/torchfx/_csrc/cuda/biquad_forward.cu:49-92
 * and, with in/out dtype f32, also the casts around it
 *           x.to(float64)       src/torchfx/_ops.py:149
 *           out.to(x.dtype)     src/torchfx/filter/iir.py:176
 *
 *   x        DEVICE [C,T] of x_dtype
 *   y        DEVICE [C,T] of y_dtype (written)
 *   sos_host HOST   [K,6] float64 rows [b0,b1,b2,a0,a1,a2]; a0 ignored
 *            (iir_cpu.cpp:86) -- this is the reference's `sos_cpu` argument
 *   state_x_in / state_y_in   DEVICE [K,C,2] float64, or NULL for zeros
 *            (_ops.py:144-147)
 *   state_x_out / state_y_out DEVICE [K,C,2] float64 (written), may be NULL
 *   y_sections  optional DEVICE [K,C,T] of y_dtype: output of every section
 *            (section-by-section parity checks); NULL in production
 *   precision  enum tfx_precision
 * ------------------------------------------------------------------------- */
int tfx_sos_forward(const void *x, int x_dtype, void *y, int y_dtype,
                    int64_t C, int64_t T,
                    const double *sos_host, int64_t K,
                    const double *state_x_in, const double *state_y_in,
                    double *state_x_out, double *state_y_out,
                    void *y_sections, int precision, tfx_stream_t stream);

/* ---------------------------------------------------------------------------
 * tfx_sos_bank_forward -- filter bank: n_bands independent K-section cascades applied to the SAME
 * input rows in one launch (x is read from HBM once, the bands' re-reads hit cache).
 * Replaces the loop of LogFilterBank.forward (src/torchfx/filter/filterbank.py:157-185:
 * `torch.stack([f(x) for f in self.filters])`) -- SURVEY.md 8(f) rank 2.
 *   x DEVICE [C,T];  y DEVICE [n_bands, C, T];  sos_host HOST [n_bands, K, 6];
 *   states DEVICE [K, n_bands*C, 2] float64 (band-major rows), NULL = zeros.
 * ------------------------------------------------------------------------- */
int tfx_sos_bank_forward(const void *x, int x_dtype, void *y, int y_dtype,
                         int64_t C, int64_t T,
                         const double *sos_host, int64_t n_bands, int64_t K,
                         const double *state_x_in, const double *state_y_in,
                         double *state_x_out, double *state_y_out,
                         int precision, tfx_stream_t stream);

/* ---------------------------------------------------------------------------
 * tfx_sos_bank_sum_forward -- `f1 + f2 + ...` of IIR branches in ONE launch: n_bands independent
 * K-section cascades applied to the same input rows and their outputs accumulated,
 *   y[C,T] = sum_b cascade_b(x[C,T]),
 * each branch rounded to the output dtype and added in branch order, exactly like
 * ParallelFilterCombination.forward (src/torchfx/filter/__base.py:1019-1026: zeros_like + in-place
 * adds of the branch outputs) -- 8 B/sample instead of n x 8 + (n + 1) x 4.  Shorter branches are
 * padded by the caller with identity sections [1,0,0,1,0,0].  x and y must have the same dtype.
 *   sos_host HOST [n_bands, K, 6];  states DEVICE [K, n_bands*C, 2] float64 (band-major rows).
 * ------------------------------------------------------------------------- */
int tfx_sos_bank_sum_forward(const void *x, int x_dtype, void *y, int y_dtype,
                             int64_t C, int64_t T,
                             const double *sos_host, int64_t n_bands, int64_t K,
                             const double *state_x_in, const double *state_y_in,
                             double *state_x_out, double *state_y_out,
                             int precision, tfx_stream_t stream);

/* What AUTO would pick for this SOS, and the plan facts (for DESIGN/bench
 * reporting and tests): *precision (TFX_PREC_F32/F64), *warmup (samples of
 * warm-up halo per time segment; -1 = filter memory too long, sequential
 * segments), *err_bound (worst-case |error| of the f32 path for |x|<=1). */
int tfx_sos_plan_info(const double *sos_host, int64_t K,
                      int *precision, int64_t *warmup, double *err_bound);

/* ---------------------------------------------------------------------------
 * tfx_biquad_forward -- single DF1 biquad.
 * Replaces  torchfx_ext.biquad_forward(x, b, a1, a2, state_x, state_y)
 *           binding.cpp:30-50,84-87 -> biquad_forward_cpu iir_cpu.cpp:10-62 /
 *           biquad_forward_cuda cuda/biquad_forward.cu:7-47.
 * b_host = HOST [3]; states DEVICE [C,2] float64 (NULL = zeros).
 * ------------------------------------------------------------------------- */
int tfx_biquad_forward(const void *x, int x_dtype, void *y, int y_dtype,
                       int64_t C, int64_t T,
                       const double *b_host, double a1, double a2,
                       const double *state_x_in, const double *state_y_in,
                       double *state_x_out, double *state_y_out,
                       int precision, tfx_stream_t stream);

/* ---------------------------------------------------------------------------
 * tfx_fir_direct_forward -- causal depthwise FIR, direct form.
 * Replaces the conv_mode="direct" branch of FIR.forward
 *           src/torchfx/filter/fir.py:556-568  (F.pad + F.conv1d(groups=C)):
 *   y[c,n] = sum_{j<K} kernel[j] * xpad[c,n+j],  xpad = x left-padded by K-1,
 * with `kernel` the FLIPPED taps exactly as FIR stores them (fir.py:516-518).
 * kernel_host: HOST [K] of `dtype`.  x, y: DEVICE [C,T] of `dtype`.
 * ------------------------------------------------------------------------- */
int tfx_fir_direct_forward(const void *x, void *y, int dtype,
                           int64_t C, int64_t T,
                           const void *kernel_host, int64_t K,
                           tfx_stream_t stream);

/* ---------------------------------------------------------------------------
 * tfx_fft_conv_forward -- overlap-save FFT convolution (rocFFT + HIP kernels).
 * Replaces  fft_conv1d(x, kernel, padding=(l,r))
 *           src/torchfx/filter/_fftconv.py:70-141
 * and through it the default conv_mode="fft" branch of FIR.forward
 *           src/torchfx/filter/fir.py:552-555.
 *   x  DEVICE [C,T];  kernel_host HOST [K] FLIPPED taps (the reference's
 *   [1,1,K] buffer);  y DEVICE [C, T+pad_left+pad_right-K+1] (written).
 * Errors like the reference: rc != 0 with "kernel size" in the message when
 * T+l+r < K (_fftconv.py:111-115).
 * The FFT block size is chosen for MI355X (power of two), not the
 * reference's int(5*K); results agree to float rounding.
 * ------------------------------------------------------------------------- */
int tfx_fft_conv_forward(const void *x, void *y, int dtype,
                         int64_t C, int64_t T,
                         const void *kernel_host, int64_t K,
                         int64_t pad_left, int64_t pad_right,
                         tfx_stream_t stream);

/* ---------------------------------------------------------------------------
 * Epilogues: `filter | Gain | Normalize` without a streaming pass per effect.
 * Replaces, when it follows a filter in a pipeline,
 *   Gain.forward                      src/torchfx/effect.py:361-383   y = x * gain, optional clip to [-1, 1]
 *   the reduction half of Normalize   src/torchfx/effect.py:696-698 (peak), 719-721 (RMS), 775-786 (per channel)
 * The producing kernel (the SOS cascade; the last pass of the overlap-save convolution) multiplies and
 * clips every sample it stores -- in the output dtype, on the rounded value a standalone Gain pass would
 * have read: bit-identical -- and gathers max|y| or sum y^2 on the fly; `stat_out` receives that raw
 * statistic (float64, DEVICE, [C] when stat_per_row else [1]) and tfx_normalize_apply turns it into
 *   y = s > 0 ? x / s * peak : x,   s = max|x|  (mode 0)  or  sqrt(sum x^2 / n)  (mode 1)
 * in one pass.  Producers without a fused epilogue (direct FIR, the rocFFT path) run the same arithmetic
 * as separate passes inside the call.
 * ------------------------------------------------------------------------- */
typedef struct tfx_epilogue {
    double gain;          /* linear factor; 1.0 = none */
    int clamp;            /* != 0: clip to [-1, 1] after the gain */
    int stat_mode;        /* -1 none, 0 max|y|, 1 sum of y^2 */
    int stat_per_row;     /* != 0: one statistic per output row, else one for the whole tensor */
    double *stat_out;     /* DEVICE float64 [C] or [1]; required when stat_mode >= 0 */
} tfx_epilogue;

/* tfx_sos_forward (no section taps) + epilogue */
int tfx_sos_forward_ep(const void *x, int x_dtype, void *y, int y_dtype, int64_t C, int64_t T,
                       const double *sos_host, int64_t K,
                       const double *state_x_in, const double *state_y_in,
                       double *state_x_out, double *state_y_out,
                       int precision, const tfx_epilogue *epilogue, tfx_stream_t stream);

/* tfx_fft_conv_forward + epilogue */
int tfx_fft_conv_forward_ep(const void *x, void *y, int dtype, int64_t C, int64_t T,
                            const void *kernel_host, int64_t K, int64_t pad_left, int64_t pad_right,
                            const tfx_epilogue *epilogue, tfx_stream_t stream);

/* ---------------------------------------------------------------------------
 * tfx_sos_fft_conv_forward -- a zero-state SOS cascade followed by an FFT-mode FIR as ONE overlap-save
 * pipeline, in the reference's own arithmetic.  Replaces two consecutive steps of Wave._materialize
 *           src/torchfx/wave.py:207-239
 * namely  parallel_iir_forward(x, sos, None, None)   src/torchfx/_ops.py:119-176  (float64 DF1 recursion,
 *           src/torchfx/_csrc/cpu/iir_cpu.cpp:132-147; fresh FusedSOSCascade: zero state, src/torchfx/filter/fused.py:62-64),
 * the downcast to the signal's float32             src/torchfx/filter/iir.py:84-184,
 * and    fft_conv1d(., kernel, padding=(l, r))      src/torchfx/filter/_fftconv.py:70-141.
 * The recursion runs in float64 registers inside the forward column pass of the three-pass transform (one
 * thread per 4096-sample row, exact warm-up from zero state), so it costs no pass over the signal of its own.
 *   x DEVICE float32 [C,T]; sos_host HOST float64 [K,6]; kernel_host HOST float32 [taps] FLIPPED;
 *   y DEVICE float32 [C, T+pad_left+pad_right-taps+1];  y_sections: optional DEVICE float64 [K,C,T], every
 *   section's output ("IIR compared section-by-section"), or NULL;  force_block 1 / 2: take the 2^20 / 2^21-point
 *   block even when the row is shorter than one block (fixture-sized parity tests).
 * tfx_sos_fft_conv_supported answers 1 when the geometry is served (T and T+l+r-taps+1 multiples of 32,
 * K <= 8 sections whose memory fades within 4096 samples, taps that select the 2^20-point block), else 0:
 * callers stage the two steps (tfx_sos_forward, tfx_fft_conv_forward) then.
 * ------------------------------------------------------------------------- */
int tfx_sos_fft_conv_supported(int64_t T, const double *sos_host, int64_t K, int64_t taps,
                               int64_t pad_left, int64_t pad_right, int force_block);
/* 1 + the geometry the fused pipeline would use (*N block length: 2^20, or 2^21 = 256 rows of 8192 samples on rows of at
 * least 2^23 samples; *S hop; *F frames per row; *warmup samples), 0 when tfx_sos_fft_conv_supported would say no.  Host-only. */
int tfx_sos_fft_conv_plan_info(int64_t T, const double *sos_host, int64_t K, int64_t taps,
                               int64_t pad_left, int64_t pad_right, int force_block,
                               int64_t *N, int64_t *S, int64_t *F, int64_t *warmup);
/* samples a row's recursion starts early from zero state inside the column pass (-1: more than 8 sections / no decay) */
int64_t tfx_sos_fft_conv_warmup(const double *sos_host, int64_t K);
int tfx_sos_fft_conv_forward(const float *x, float *y, int64_t C, int64_t T,
                             const double *sos_host, int64_t K,
                             const float *kernel_host, int64_t taps,
                             int64_t pad_left, int64_t pad_right,
                             double *y_sections, int force_block,
                             const tfx_epilogue *epilogue, tfx_stream_t stream);

/* the apply half of Normalize on a statistic left by an epilogue (or by tfx_stat_forward's raw form) */
int tfx_normalize_apply(const void *x, void *y, int dtype, int64_t C, int64_t T, int mode, int per_row,
                        double peak, const double *stat_dev, tfx_stream_t stream);

/* ---------------------------------------------------------------------------
 * tfx_fir_stream_forward -- one chunk of a stateful FIR (streaming).
 * The reference's FIR is stateless (src/torchfx/filter/fir.py:526-579: every call left-pads with K-1
 * zeros), so StreamProcessor (src/torchfx/realtime/stream.py:164-347) is only seamless for FIR stages
 * with overlap >= K-1.  Here the chunk is filtered as the continuation of what came before:
 *   y[c,n] = sum_{j<K} kernel[j] * xv[c, n+j],   xv = [hist_in[c, 0..K-2] | x[c, 0..T-1]],
 * the kernels read the K-1 history samples and the chunk from their two buffers (no concatenated copy),
 * and hist_out receives the last K-1 samples of xv for the next call.
 *   hist_in  DEVICE [C, K-1] of `dtype` or NULL (= zeros: first chunk);  hist_out DEVICE [C, K-1] or NULL,
 *   a different buffer than hist_in;  direct != 0: time-domain kernels, else overlap-save.
 * ------------------------------------------------------------------------- */
int tfx_fir_stream_forward(const void *x, void *y, int dtype, int64_t C, int64_t T,
                           const void *kernel_host, int64_t K, int direct,
                           const void *hist_in, void *hist_out, tfx_stream_t stream);

/* tfx_quantile_abs -- out_dev[0] (DEVICE float64) = the q-quantile (0 <= q <= 1, linear interpolation) of |x| over all n float32
 * elements: the threshold of PercentileNormalizationStrategy (src/torchfx/effect.py:723-755,
 * `torch.quantile(torch.abs(waveform), p / 100, interpolation="linear")`) as a three-pass radix SELECT instead of a sort -- same
 * value as torch.quantile wherever that runs (float32 rank arithmetic and lerp of ATen), no 16 M element limit, no host sync.
 * Feed out_dev to tfx_normalize_apply (mode 0) for the scaling.  NaN anywhere in x -> NaN. */
int tfx_quantile_abs(const float *x, int64_t n, double q, double *out_dev, tfx_stream_t stream);

/* ---------------------------------------------------------------------------
 * tfx_chunk_forward -- ONE launch for one small streaming chunk: SOS cascade -> stateful direct FIR -> gain / clip.
 * Replaces, for the reference's small-block caller (RealtimeProcessor._audio_callback,
 * src/torchfx/realtime/processor.py:253-292, and StreamProcessor's chunk loop, realtime/stream.py:234-273), the
 * per-effect launches of   IIR ... | FIR | Gain   on a [C, T] float32 block:
 *   u = cascade(x) with carried DF1 state (layout and rounding of tfx_sos_forward, float32 output),
 *   y[c,n] = clip(gain * sum_{j<Kf} taps[j] * [hist_in[c] | u[c]][n+j]),  hist_out[c] = last Kf-1 samples of [hist_in[c] | u[c]].
 * Same arithmetic as tfx_sos_forward -> tfx_fir_stream_forward(direct) -> tfx_gain_forward (the cascade walks 16-sample lane
 * chunks instead of 64: results agree to float64 round-off of the recursion, i.e. to the last float32 bit in all but rare samples).
 *   K = 0: no cascade (u = x);  Kf = 1 with taps {1}: no FIR;  scale / clamp = 0: no gain stage.
 *   Limits (tfx_chunk_supported): T <= 4096, K <= 64, Kf <= 4096, T * Kf <= 2^22 -- a latency path, not a throughput one.
 *   x_pitch: elements between consecutive rows of x (a chunk is usually a column window of a longer [C, T_total] buffer:
 *   no contiguous copy needed); 0 = T.  y is contiguous [C, T].
 *   state pointers: DEVICE float64 [K, C, 2] (in: NULL = zeros);  hist: DEVICE float32 [C, Kf-1], in and out distinct;
 *   sos_host [K, 6] HOST float64;  taps_host HOST float32 (flipped, like the module's kernel buffer).
 * ------------------------------------------------------------------------- */
int tfx_chunk_supported(int64_t C, int64_t T, int64_t K, int64_t Kf);
int tfx_chunk_forward(const float *x, int64_t x_pitch, float *y, int64_t C, int64_t T,
                      const double *sos_host, int64_t K,
                      const double *state_x_in, const double *state_y_in, double *state_x_out, double *state_y_out,
                      const float *taps_host, int64_t Kf, const float *hist_in, float *hist_out,
                      double gain, int scale, int clamp, int precision, tfx_stream_t stream);

/* Block geometry the overlap-save op would use for a [*, T] signal and a K-tap kernel with
 * padding (l, r): *N = FFT block length, *S = hop (valid outputs per block), *F = blocks per row,
 * *native = 1 when a hand-written LDS-FFT path runs -- the three-pass pipeline OR (since round 4) a one-launch kernel --
 * and 0 for the rocFFT path: callers that price traffic must use tfx_ols_plan_info2's *path (the 20 N/S + 4 model
 * holds for path 1 only).  No GPU needed. */
int tfx_ols_plan_info(int64_t K, int64_t T, int64_t pad_left, int64_t pad_right,
                      int64_t *N, int64_t *S, int64_t *F, int *native);
/* The same for a signal of `dtype` (tfx_ols_plan_info answers for float32).  *path = 2: one launch, the
 * whole transform of a block in LDS and registers (*N = 4096, 8192 or 16384: K <= 8192 in float32, K <= 4096 in
 * float64; e N/S + e bytes of HBM traffic per output sample, e = element size); 1: the three-pass four-step
 * pipeline (float32, 20 N/S + 4); 0: rocFFT (~95, float64 ~190). */
int tfx_ols_plan_info2(int64_t K, int64_t T, int64_t pad_left, int64_t pad_right, int dtype,
                       int64_t *N, int64_t *S, int64_t *F, int *path);

/* Start, on a helper thread, the one-time per-device set-up of the overlap-save path (kernel attributes = load of the
 * library's code object, internal streams and events: 20-35 ms of driver time in the first call of a process) for the
 * device current at the call; returns at once, the first overlap-save call waits for it.  Optional: a caller that has
 * host work of its own before its first call (the Python planner merges taps, src/torchfx/wave.py:207-239 is the
 * reference's counterpart) overlaps the two.  No reference counterpart; needs a device. */
int tfx_prewarm(void);
/* TFX_* tuning knobs are read from the environment ONCE per process (a dispatch asks for about ten of them); a process
 * that changes them at run time calls this afterwards -- or sets TFX_ENV_DYNAMIC=1 before the first call, which makes
 * every lookup a fresh getenv (the test suite does). */
int tfx_env_reload(void);

/* ---------------------------------------------------------------------------
 * tfx_delay_line_forward -- kept because the reference extension exports it
 * (binding.cpp:68-81,92-95; tests/test_ops_dispatch.py:29-35); out of the
 * hot-path scope.  y = x + (mix*decay) * x[n-delay]  (delay_cpu.cpp:17-41).
 * ------------------------------------------------------------------------- */
int tfx_delay_line_forward(const void *x, void *y, int dtype, int64_t C, int64_t T,
                           int64_t delay, double decay, double mix, tfx_stream_t stream);

/* ---------------------------------------------------------------------------
 * tfx_sum_forward -- y = sum_i xs[i]  (the accumulate of
 * ParallelFilterCombination.forward, src/torchfx/filter/__base.py:1019-1026).
 * xs_host: HOST array of n DEVICE pointers, each [numel] of dtype.
 * ------------------------------------------------------------------------- */
int tfx_sum_forward(const void *const *xs_host, int n, void *y, int dtype,
                    int64_t numel, tfx_stream_t stream);

/* ---------------------------------------------------------------------------
 * Elementwise effects that sit between filters in a pipeline.
 *
 * tfx_gain_forward -- Gain.forward, src/torchfx/effect.py:361-383:  y = x * gain, then (clamp != 0)
 *   clip to [-1, 1].  `gain` is the LINEAR factor (the host layer maps "db" / "power" through
 *   10^(g/20), effect.py:132-136,372-378) and is rounded to the signal dtype like torch does for
 *   tensor * python_float.  y may alias x (pure elementwise).
 *
 * tfx_stat_forward -- the reductions behind the normalization strategies (effect.py:678-790):
 *   mode TFX_STAT_ABSMAX: max|x| ;  TFX_STAT_RMS: sqrt(mean(x^2)) (float64 accumulation, deterministic).
 *   per_row != 0: one value per row -> out_dev[C];  else one value over all C*T -> out_dev[1].
 *   out_dev: DEVICE float64.  NaN anywhere gives NaN (torch.max semantics).
 *
 * tfx_normalize_forward -- Normalize.forward with PeakNormalizationStrategy (mode ABSMAX, per_row 0,
 *   effect.py:696-698), PerChannelNormalizationStrategy (ABSMAX, per_row 1, :775-786) or
 *   RMSNormalizationStrategy (RMS, per_row 0, :719-721):  y = s > 0 ? (x / s) * peak : x, evaluated in
 *   the signal dtype in that order.  The statistic never leaves the device (the reference's
 *   `if max_val > 0` is a blocking host read).  y may alias x.
 * ------------------------------------------------------------------------- */
enum tfx_stat { TFX_STAT_ABSMAX = 0, TFX_STAT_RMS = 1 };
int tfx_gain_forward(const void *x, void *y, int dtype, int64_t numel, double gain, int clamp,
                     tfx_stream_t stream);
int tfx_stat_forward(const void *x, int dtype, int64_t C, int64_t T, int mode, int per_row,
                     double *out_dev, tfx_stream_t stream);
int tfx_normalize_forward(const void *x, void *y, int dtype, int64_t C, int64_t T, int mode, int per_row,
                          double peak, tfx_stream_t stream);

/* ---------------------------------------------------------------------------
 * The data-format edge: decoded audio is INTERLEAVED frames [F, C]; the filter
 * path works on PLANAR rows [C, F].  The reference transposes on the host (`data_np.T.copy()`,
 * src/torchfx/wave.py:448-452, and `.numpy().T` before writing, :566-573); these do it on the device so
 * the host buffer can be uploaded as it is, in chunks, and 16-bit PCM can cross PCIe as 2-byte samples.
 *
 * tfx_deinterleave_forward: in DEVICE [F, C] of float32 (in_kind TFX_PCM_F32) or int16 (TFX_PCM_S16,
 *   converted as v * scale; scale = 1/32768 is libsndfile's normalisation);
 *   out DEVICE float32 rows of pitch ld_out: out[c * ld_out + f_base + f] = in[f * C + c].
 *   f_base / ld_out let a file be uploaded chunk by chunk into one [C, F_total] tensor.
 * tfx_interleave_forward: the inverse, out[f * C + c] = in[c * ld_in + f_base + f], float32.
 * ------------------------------------------------------------------------- */
enum tfx_pcm { TFX_PCM_F32 = 0, TFX_PCM_S16 = 1 };
int tfx_deinterleave_forward(const void *in, int in_kind, void *out, int64_t F, int64_t C, int64_t ld_out,
                             int64_t f_base, double scale, tfx_stream_t stream);
int tfx_interleave_forward(const void *in, void *out, int64_t F, int64_t C, int64_t ld_in, int64_t f_base,
                           tfx_stream_t stream);

/* Timing hooks for bench.py: HIP events recorded on the SAME stream the
 * kernels are launched on (torch.cuda.Event only sees torch's current stream).
 * tfx_prof_enable(1) makes every kernel launch inside the library bracket
 * itself with events; tfx_prof_collect() synchronises and returns, per kernel
 * name, call count and total milliseconds as a JSON string (static buffer). */
int tfx_prof_enable(int on);
const char *tfx_prof_collect(void);

/* Drop all cached plans / device workspaces (tests, memory pressure). */
int tfx_clear_caches(void);

/* Workspace under the caller's control.  The overlap-save pipelines keep device workspaces between calls (the default chain
 * step: three lanes x up to 3.75 GB); by default they come from hipMalloc / hipFree.  A host that runs its own device allocator
 * installs it here -- the torch module (csrc/ext/torchfx_ext.cpp) routes the workspaces through PyTorch's caching allocator, so
 * torch.cuda.memory_allocated(), its free-cached-blocks-and-retry and its OutOfMemoryError cover them (the reference allocates
 * every temporary as a torch tensor: src/torchfx/filter/_fftconv.py:119-140).
 *   alloc_fn(bytes, device, stream, ctx) -> device pointer, or NULL when it cannot (the pipeline then asks for a smaller slab:
 *       fewer frame pairs per launch, down to 8; below that the call fails with "out of device memory")
 *   free_fn(ptr, device, ctx)               called after a device synchronise
 * Both NULL restores hipMalloc / hipFree.  Buffers held at the time of the call keep the allocator they came from.
 * tfx_workspace_bytes(): bytes of workspace held right now (all streams and devices); tfx_clear_caches() releases them. */
typedef void *(*tfx_alloc_fn)(size_t bytes, int device, void *stream, void *ctx);
typedef void (*tfx_free_fn)(void *ptr, int device, void *ctx);
int tfx_set_workspace_allocator(tfx_alloc_fn alloc_fn, tfx_free_fn free_fn, void *ctx);
int64_t tfx_workspace_bytes(void);

#ifdef __cplusplus
}
#endif
#endif /* TORCHFX_HIP_H */
