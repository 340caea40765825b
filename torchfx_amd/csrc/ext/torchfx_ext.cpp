// torchfx_ext -- the compiled boundary module of the HIP backend.
//
// The reference binds its native kernels through a pybind11 torch extension named `torchfx_ext`
// (src/torchfx/_csrc/binding.cpp:83-96, imported as `from torchfx import torchfx_ext`).  This file is
// that module for MI355X: the same three entry points with the same signatures on at::Tensor
// (biquad_forward, sos_forward, delay_line_forward), launched on PyTorch's current HIP stream, plus a
// TORCH_LIBRARY(torchfx_hip) registration of every op of the backend so that Python reaches the kernels
// through the dispatcher (torch.ops.torchfx_hip.*) -- no ctypes on the tensor path.  It is a thin,
// host-only translation unit (compiled with g++): tensors are checked, outputs allocated, and the
// extern "C" ABI of libtorchfx_hip.so (include/torchfx_hip.h) is called -- the C ABI stays the one
// boundary non-torch hosts and this module share.
//
// Host tensors: the reference's module dispatches on x.is_cuda() (binding.cpp:30-81), so the three pybind entry points
// at the bottom do too -- their host branch is host_branch.h (this module's own Direct Form I loop).  Everything else is
// device-only: the dispatcher ops are registered for the CUDA key (= ROCm device tensors in a ROCm build of PyTorch) and
// for Meta (shape inference), and a CPU tensor gets an explicit error there.
#include <torch/extension.h>
#include <torch/library.h>

#include <c10/hip/HIPCachingAllocator.h>
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>

#include <cstdlib>
#include <exception>
#include <optional>
#include <string>
#include <tuple>
#include <vector>

#include "../../../include/torchfx_hip.h"
#include "host_branch.h"

namespace {

using at::Tensor;
using OptTensor = std::optional<Tensor>;

// ---- workspaces through PyTorch's caching allocator (tfx_set_workspace_allocator, include/torchfx_hip.h) -------------------------
// The overlap-save pipelines keep multi-GB workspaces between calls.  Allocated here they show up in torch.cuda.memory_allocated(),
// an allocation under memory pressure first frees torch's cached blocks and retries, and when it still fails the library asks
// for a smaller slab; only if even 8 frame pairs do not fit does the op fail -- with torch's own OutOfMemoryError (kept in
// `pending_oom` by the hook, rethrown by check_rc), not a raw hipErrorOutOfMemory.  TORCHFX_AMD_WORKSPACE=hip keeps hipMalloc.
thread_local std::exception_ptr pending_oom;

void *torch_ws_alloc(size_t bytes, int device, void *stream, void *)
{
    try {
        c10::hip::HIPGuard guard((c10::DeviceIndex)device);
        return c10::hip::HIPCachingAllocator::raw_alloc_with_stream(bytes, (hipStream_t)stream);
    } catch (...) {
        pending_oom = std::current_exception();
        return nullptr;
    }
}

void torch_ws_free(void *ptr, int device, void *)
{
    try {
        c10::hip::HIPGuard guard((c10::DeviceIndex)device);
        c10::hip::HIPCachingAllocator::raw_delete(ptr);
    } catch (...) {            // interpreter shutdown: the allocator may be gone
    }
}

struct WorkspaceHook {
    WorkspaceHook()
    {
        const char *e = std::getenv("TORCHFX_AMD_WORKSPACE");
        if (!(e && std::string(e) == "hip")) tfx_set_workspace_allocator(torch_ws_alloc, torch_ws_free, nullptr);
    }
} workspace_hook;

void check_rc(int rc, const char *what)
{
    std::exception_ptr oom;
    std::swap(oom, pending_oom);           // an allocation that failed on the way to a smaller slab is not an error of a call that succeeded
    if (rc != 0 && oom) std::rethrow_exception(oom);
    TORCH_CHECK(rc == 0, what, ": ", tfx_last_error());
}

int dtype_code(const Tensor &t, const char *what)
{
    if (t.scalar_type() == at::kFloat) return TFX_F32;
    if (t.scalar_type() == at::kDouble) return TFX_F64;
    TORCH_CHECK(false, what, ": expected a float32 or float64 tensor, got ", t.scalar_type());
}

void need_device(const Tensor &t, const char *what)
{
    TORCH_CHECK(t.is_cuda(), "torchfx_amd: ", what, " must live on a ROCm device (got ", t.device(),
                "); this backend has no CPU path -- move the tensor with .to('cuda').");
}

tfx_stream_t stream_of(const Tensor &t)
{
    return (tfx_stream_t)c10::hip::getCurrentHIPStream(t.get_device()).stream();
}

// small coefficient tensor -> contiguous host float64 (O(K) bytes; the C ABI takes coefficients on the host)
Tensor host_f64(const Tensor &t, int64_t last, const char *what)
{
    Tensor h = t.detach().to(at::kCPU, at::kDouble).contiguous();
    TORCH_CHECK(h.dim() >= 1 && h.size(-1) == last, what, ": expected last dimension ", last, ", got shape ", h.sizes());
    return h;
}

const double *state_ptr(const OptTensor &s, at::IntArrayRef shape, const Tensor &x, const char *what, Tensor &keep)
{
    if (!s.has_value() || !s->defined()) return nullptr;
    TORCH_CHECK(s->sizes() == shape, what, " must have shape ", shape, ", got ", s->sizes());
    keep = s->to(x.device(), at::kDouble).contiguous();
    return keep.data_ptr<double>();
}

int precision_or_default(int64_t precision)
{
    if (precision >= 0) return (int)precision;
    const char *e = getenv("TORCHFX_AMD_IIR_PRECISION");
    if (!e || !*e) return TFX_PREC_F64;
    const std::string s(e);
    if (s == "f64" || s == "float64") return TFX_PREC_F64;
    if (s == "f32" || s == "float32") return TFX_PREC_F32;
    if (s == "auto") return TFX_PREC_AUTO;
    TORCH_CHECK(false, "TORCHFX_AMD_IIR_PRECISION=", s, ": expected f64, f32 or auto");
}

at::ScalarType out_type(const Tensor &x, const std::optional<at::ScalarType> &out_dtype)
{
    return out_dtype.has_value() ? *out_dtype : x.scalar_type();
}

// ---------------------------------------------------------------------------------------------------
// SOS cascade (binding.cpp:52-66) and its filter-bank / sum forms
// ---------------------------------------------------------------------------------------------------
std::tuple<Tensor, Tensor, Tensor, Tensor> sos_impl(const Tensor &x_in, const Tensor &sos_cpu, const OptTensor &state_x,
                                                    const OptTensor &state_y, std::optional<at::ScalarType> out_dtype,
                                                    int64_t precision, bool sections)
{
    TORCH_CHECK(x_in.dim() == 2, "sos_forward: x must be [C, T], got ", x_in.sizes());
    need_device(x_in, "x");
    const Tensor x = x_in.contiguous();
    const Tensor sos = host_f64(sos_cpu, 6, "sos_forward");
    TORCH_CHECK(sos.dim() == 2, "sos_forward: sos must be [K, 6]");
    const int64_t C = x.size(0), T = x.size(1), K = sos.size(0);
    Tensor kx, ky;
    const double *sx = state_ptr(state_x, {K, C, 2}, x, "state_x", kx);
    const double *sy = state_ptr(state_y, {K, C, 2}, x, "state_y", ky);
    const auto odt = out_type(x, out_dtype);
    Tensor y = at::empty({C, T}, x.options().dtype(odt));
    Tensor nsx = at::empty({K, C, 2}, x.options().dtype(at::kDouble));
    Tensor nsy = at::empty({K, C, 2}, x.options().dtype(at::kDouble));
    Tensor sec = sections ? at::empty({K, C, T}, x.options().dtype(odt)) : at::empty({0}, x.options().dtype(odt));
    c10::hip::HIPGuard guard(x.get_device());
    check_rc(tfx_sos_forward(x.data_ptr(), dtype_code(x, "sos_forward"), y.data_ptr(), dtype_code(y, "sos_forward"), C, T,
                             sos.data_ptr<double>(), K, sx, sy, nsx.data_ptr<double>(), nsy.data_ptr<double>(),
                             sections ? sec.data_ptr() : nullptr, precision_or_default(precision), stream_of(x)),
             "sos_forward");
    return {y, nsx, nsy, sec};
}

std::tuple<Tensor, Tensor, Tensor> sos_op(const Tensor &x, const Tensor &sos_cpu, const OptTensor &sx, const OptTensor &sy,
                                          std::optional<at::ScalarType> out_dtype, int64_t precision)
{
    auto r = sos_impl(x, sos_cpu, sx, sy, out_dtype, precision, false);
    return {std::get<0>(r), std::get<1>(r), std::get<2>(r)};
}

std::tuple<Tensor, Tensor, Tensor, Tensor> sos_sections_op(const Tensor &x, const Tensor &sos_cpu, const OptTensor &sx,
                                                           const OptTensor &sy, std::optional<at::ScalarType> out_dtype,
                                                           int64_t precision)
{
    return sos_impl(x, sos_cpu, sx, sy, out_dtype, precision, true);
}

std::tuple<Tensor, Tensor, Tensor> bank_impl(const Tensor &x_in, const Tensor &banks_cpu, const OptTensor &state_x,
                                             const OptTensor &state_y, std::optional<at::ScalarType> out_dtype,
                                             int64_t precision, bool sum)
{
    const char *what = sum ? "sos_bank_sum_forward" : "sos_bank_forward";
    TORCH_CHECK(x_in.dim() == 2, what, ": x must be [C, T], got ", x_in.sizes());
    need_device(x_in, "x");
    const Tensor x = x_in.contiguous();
    const Tensor banks = host_f64(banks_cpu, 6, what);
    TORCH_CHECK(banks.dim() == 3, what, ": sos_banks must be [NB, K, 6]");
    const int64_t C = x.size(0), T = x.size(1), NB = banks.size(0), K = banks.size(1);
    Tensor kx, ky;
    const double *sx = state_ptr(state_x, {K, NB * C, 2}, x, "state_x", kx);
    const double *sy = state_ptr(state_y, {K, NB * C, 2}, x, "state_y", ky);
    const auto odt = sum ? x.scalar_type() : out_type(x, out_dtype);
    Tensor y = sum ? at::empty({C, T}, x.options()) : at::empty({NB, C, T}, x.options().dtype(odt));
    Tensor nsx = at::empty({K, NB * C, 2}, x.options().dtype(at::kDouble));
    Tensor nsy = at::empty({K, NB * C, 2}, x.options().dtype(at::kDouble));
    c10::hip::HIPGuard guard(x.get_device());
    auto fn = sum ? tfx_sos_bank_sum_forward : tfx_sos_bank_forward;
    check_rc(fn(x.data_ptr(), dtype_code(x, what), y.data_ptr(), dtype_code(y, what), C, T, banks.data_ptr<double>(), NB, K,
                sx, sy, nsx.data_ptr<double>(), nsy.data_ptr<double>(), precision_or_default(precision), stream_of(x)),
             what);
    return {y, nsx, nsy};
}

std::tuple<Tensor, Tensor, Tensor> bank_op(const Tensor &x, const Tensor &banks, const OptTensor &sx, const OptTensor &sy,
                                           std::optional<at::ScalarType> out_dtype, int64_t precision)
{
    return bank_impl(x, banks, sx, sy, out_dtype, precision, false);
}
std::tuple<Tensor, Tensor, Tensor> bank_sum_op(const Tensor &x, const Tensor &banks, const OptTensor &sx, const OptTensor &sy,
                                               int64_t precision)
{
    return bank_impl(x, banks, sx, sy, std::nullopt, precision, true);
}

// cascade + epilogue (Gain / clamp / statistic for Normalize applied by the producing kernel, include/torchfx_hip.h)
tfx_epilogue make_epilogue(double gain, bool clamp, int64_t stat_mode, bool per_row, Tensor &stat, const Tensor &like, int64_t rows)
{
    TORCH_CHECK(stat_mode >= -1 && stat_mode <= 1, "epilogue: stat_mode must be -1 (none), 0 (max|y|) or 1 (sum y^2)");
    stat = at::empty({stat_mode >= 0 ? (per_row ? rows : 1) : 0}, like.options().dtype(at::kDouble));
    tfx_epilogue ep;
    ep.gain = gain; ep.clamp = clamp ? 1 : 0; ep.stat_mode = (int)stat_mode; ep.stat_per_row = per_row ? 1 : 0;
    ep.stat_out = stat_mode >= 0 ? stat.data_ptr<double>() : nullptr;
    return ep;
}

std::tuple<Tensor, Tensor, Tensor, Tensor> sos_ep_op(const Tensor &x_in, const Tensor &sos_cpu, const OptTensor &state_x,
                                                     const OptTensor &state_y, double gain, bool clamp, int64_t stat_mode,
                                                     bool per_row, std::optional<at::ScalarType> out_dtype, int64_t precision)
{
    TORCH_CHECK(x_in.dim() == 2, "sos_forward_ep: x must be [C, T], got ", x_in.sizes());
    need_device(x_in, "x");
    const Tensor x = x_in.contiguous();
    const Tensor sos = host_f64(sos_cpu, 6, "sos_forward_ep");
    TORCH_CHECK(sos.dim() == 2, "sos_forward_ep: sos must be [K, 6]");
    const int64_t C = x.size(0), T = x.size(1), K = sos.size(0);
    Tensor kx, ky, stat;
    const double *sx = state_ptr(state_x, {K, C, 2}, x, "state_x", kx);
    const double *sy = state_ptr(state_y, {K, C, 2}, x, "state_y", ky);
    Tensor y = at::empty({C, T}, x.options().dtype(out_type(x, out_dtype)));
    Tensor nsx = at::empty({K, C, 2}, x.options().dtype(at::kDouble));
    Tensor nsy = at::empty({K, C, 2}, x.options().dtype(at::kDouble));
    const tfx_epilogue ep = make_epilogue(gain, clamp, stat_mode, per_row, stat, x, C);
    c10::hip::HIPGuard guard(x.get_device());
    check_rc(tfx_sos_forward_ep(x.data_ptr(), dtype_code(x, "sos_forward_ep"), y.data_ptr(), dtype_code(y, "sos_forward_ep"), C, T,
                                sos.data_ptr<double>(), K, sx, sy, nsx.data_ptr<double>(), nsy.data_ptr<double>(),
                                precision_or_default(precision), &ep, stream_of(x)),
             "sos_forward_ep");
    return {y, nsx, nsy, stat};
}

// single biquad (binding.cpp:30-50): b [3] tensor, a1 / a2 scalars, states [C, 2]
std::tuple<Tensor, Tensor, Tensor> biquad_op(const Tensor &x_in, const Tensor &b, double a1, double a2, const OptTensor &state_x,
                                             const OptTensor &state_y, std::optional<at::ScalarType> out_dtype, int64_t precision)
{
    TORCH_CHECK(x_in.dim() == 2, "biquad_forward: x must be [C, T], got ", x_in.sizes());
    need_device(x_in, "x");
    const Tensor x = x_in.contiguous();
    const Tensor bh = host_f64(b.reshape({-1}), 3, "biquad_forward");
    const int64_t C = x.size(0), T = x.size(1);
    Tensor kx, ky;
    const double *sx = state_ptr(state_x, {C, 2}, x, "state_x", kx);
    const double *sy = state_ptr(state_y, {C, 2}, x, "state_y", ky);
    Tensor y = at::empty({C, T}, x.options().dtype(out_type(x, out_dtype)));
    Tensor nsx = at::empty({C, 2}, x.options().dtype(at::kDouble));
    Tensor nsy = at::empty({C, 2}, x.options().dtype(at::kDouble));
    c10::hip::HIPGuard guard(x.get_device());
    check_rc(tfx_biquad_forward(x.data_ptr(), dtype_code(x, "biquad_forward"), y.data_ptr(), dtype_code(y, "biquad_forward"), C, T,
                                bh.data_ptr<double>(), a1, a2, sx, sy, nsx.data_ptr<double>(), nsy.data_ptr<double>(),
                                precision_or_default(precision), stream_of(x)),
             "biquad_forward");
    return {y, nsx, nsy};
}

// delay line (binding.cpp:68-81 / delay_cpu.cpp:43-85): the input itself when the signal is not longer than the delay
Tensor delay_line_op(const Tensor &x, int64_t delay_samples, double decay, double mix)
{
    need_device(x, "x");
    const int64_t T = x.dim() ? x.size(-1) : 1;
    if (T <= delay_samples) return x;
    const Tensor xc = x.contiguous();
    const int64_t rows = xc.numel() / T;
    Tensor y = at::empty_like(xc);
    c10::hip::HIPGuard guard(x.get_device());
    check_rc(tfx_delay_line_forward(xc.data_ptr(), y.data_ptr(), dtype_code(xc, "delay_line_forward"), rows, T, delay_samples,
                                    decay, mix, stream_of(x)),
             "delay_line_forward");
    return y;
}

// ---------------------------------------------------------------------------------------------------
// FIR (fir.py:556-568) and overlap-save FFT convolution (_fftconv.py:70-141)
// ---------------------------------------------------------------------------------------------------
Tensor taps_host(const Tensor &kernel, const Tensor &x)
{
    return kernel.detach().reshape({-1}).to(at::kCPU, x.scalar_type()).contiguous();
}

Tensor fir_direct_op(const Tensor &x_in, const Tensor &kernel)
{
    TORCH_CHECK(x_in.dim() == 2, "fir_direct_forward: x must be [C, T], got ", x_in.sizes());
    need_device(x_in, "x");
    const Tensor x = x_in.contiguous();
    const Tensor k = taps_host(kernel, x);
    Tensor y = at::empty_like(x);
    c10::hip::HIPGuard guard(x.get_device());
    check_rc(tfx_fir_direct_forward(x.data_ptr(), y.data_ptr(), dtype_code(x, "fir_direct_forward"), x.size(0), x.size(1),
                                    k.data_ptr(), k.numel(), stream_of(x)),
             "fir_direct_forward");
    return y;
}

Tensor fft_conv_op(const Tensor &x_in, const Tensor &kernel, int64_t pad_left, int64_t pad_right)
{
    TORCH_CHECK(x_in.dim() == 2, "fft_conv_forward: x must be [C, T], got ", x_in.sizes());
    need_device(x_in, "x");
    const Tensor x = x_in.contiguous();
    const Tensor k = taps_host(kernel, x);
    const int64_t C = x.size(0), T = x.size(1), K = k.numel();
    const int64_t tout = T + pad_left + pad_right - K + 1;
    Tensor y = at::empty({C, tout > 0 ? tout : 0}, x.options());
    c10::hip::HIPGuard guard(x.get_device());
    check_rc(tfx_fft_conv_forward(x.data_ptr(), y.data_ptr(), dtype_code(x, "fft_conv_forward"), C, T, k.data_ptr(), K, pad_left,
                                  pad_right, stream_of(x)),
             "fft_conv_forward");
    return y;
}

std::tuple<Tensor, Tensor> fft_conv_ep_op(const Tensor &x_in, const Tensor &kernel, int64_t pad_left, int64_t pad_right, double gain,
                                          bool clamp, int64_t stat_mode, bool per_row)
{
    TORCH_CHECK(x_in.dim() == 2, "fft_conv_forward_ep: x must be [C, T], got ", x_in.sizes());
    need_device(x_in, "x");
    const Tensor x = x_in.contiguous();
    const Tensor k = taps_host(kernel, x);
    const int64_t C = x.size(0), T = x.size(1), K = k.numel();
    const int64_t tout = T + pad_left + pad_right - K + 1;
    Tensor y = at::empty({C, tout > 0 ? tout : 0}, x.options()), stat;
    const tfx_epilogue ep = make_epilogue(gain, clamp, stat_mode, per_row, stat, x, C);
    c10::hip::HIPGuard guard(x.get_device());
    check_rc(tfx_fft_conv_forward_ep(x.data_ptr(), y.data_ptr(), dtype_code(x, "fft_conv_forward_ep"), C, T, k.data_ptr(), K, pad_left,
                                     pad_right, &ep, stream_of(x)),
             "fft_conv_forward_ep");
    return {y, stat};
}

// zero-state SOS cascade | FFT-mode FIR as one overlap-save pipeline (tfx_sos_fft_conv_forward): y, the statistic of the
// epilogue and -- on request -- every section's float64 output [K, C, T]
std::tuple<Tensor, Tensor, Tensor> sos_fft_conv_op(const Tensor &x_in, const Tensor &sos_cpu, const Tensor &kernel, int64_t pad_left,
                                                   int64_t pad_right, bool sections, int64_t force_block, double gain, bool clamp,
                                                   int64_t stat_mode, bool per_row)
{
    TORCH_CHECK(x_in.dim() == 2, "sos_fft_conv_forward: x must be [C, T], got ", x_in.sizes());
    need_device(x_in, "x");
    TORCH_CHECK(x_in.scalar_type() == at::kFloat, "sos_fft_conv_forward: float32 signals only, got ", x_in.scalar_type());
    const Tensor x = x_in.contiguous();
    const Tensor sos = host_f64(sos_cpu, 6, "sos_fft_conv_forward");
    TORCH_CHECK(sos.dim() == 2, "sos_fft_conv_forward: sos must be [K, 6]");
    const Tensor k = taps_host(kernel, x);
    const int64_t C = x.size(0), T = x.size(1), K = sos.size(0), taps = k.numel();
    const int64_t tout = T + pad_left + pad_right - taps + 1;
    Tensor y = at::empty({C, tout > 0 ? tout : 0}, x.options()), stat;
    Tensor sec = sections ? at::empty({K, C, T}, x.options().dtype(at::kDouble)) : at::empty({0}, x.options().dtype(at::kDouble));
    const tfx_epilogue ep = make_epilogue(gain, clamp, stat_mode, per_row, stat, x, C);
    c10::hip::HIPGuard guard(x.get_device());
    check_rc(tfx_sos_fft_conv_forward(x.data_ptr<float>(), y.data_ptr<float>(), C, T, sos.data_ptr<double>(), K, k.data_ptr<float>(), taps,
                                      pad_left, pad_right, sections ? sec.data_ptr<double>() : nullptr, (int)force_block, &ep,
                                      stream_of(x)),
             "sos_fft_conv_forward");
    return {y, stat, sec};
}

// the apply half of Normalize on a statistic an epilogue left on the device
Tensor normalize_apply_op(const Tensor &x, const Tensor &stat, double peak, int64_t mode, bool per_row)
{
    need_device(x, "x");
    need_device(stat, "stat");
    const Tensor xc = x.contiguous();
    const int64_t T = xc.dim() ? xc.size(-1) : 1, rows = T ? xc.numel() / T : 0;
    TORCH_CHECK(stat.scalar_type() == at::kDouble && stat.numel() == (per_row ? rows : 1), "normalize_apply: stat must be float64 [",
                per_row ? rows : 1, "], got ", stat.sizes(), " ", stat.scalar_type());
    const Tensor st = stat.contiguous();
    Tensor y = at::empty_like(xc);
    c10::hip::HIPGuard guard(x.get_device());
    check_rc(tfx_normalize_apply(xc.data_ptr(), y.data_ptr(), dtype_code(xc, "normalize_apply"), rows, T, (int)mode, per_row ? 1 : 0,
                                 peak, st.data_ptr<double>(), stream_of(x)),
             "normalize_apply");
    return y;
}

// one chunk of a stateful FIR: history and chunk are read from their two buffers, the new history is returned
std::tuple<Tensor, Tensor> fir_stream_op(const Tensor &x_in, const Tensor &kernel, const OptTensor &hist, bool direct)
{
    TORCH_CHECK(x_in.dim() == 2, "fir_stream_forward: x must be [C, T], got ", x_in.sizes());
    need_device(x_in, "x");
    const Tensor x = x_in.contiguous();
    const Tensor k = taps_host(kernel, x);
    const int64_t C = x.size(0), T = x.size(1), K = k.numel();
    TORCH_CHECK(K >= 1, "fir_stream_forward: empty kernel");
    Tensor hin;
    const void *hp = nullptr;
    if (hist.has_value() && hist->defined() && K > 1) {
        TORCH_CHECK(hist->dim() == 2 && hist->size(0) == C && hist->size(1) == K - 1, "fir_stream_forward: history must be [C, K-1] = [",
                    C, ", ", K - 1, "], got ", hist->sizes());
        hin = hist->to(x.device(), x.scalar_type()).contiguous();
        hp = hin.data_ptr();
    }
    Tensor y = at::empty_like(x);
    Tensor hout = at::empty({C, K - 1}, x.options());
    c10::hip::HIPGuard guard(x.get_device());
    check_rc(tfx_fir_stream_forward(x.data_ptr(), y.data_ptr(), dtype_code(x, "fir_stream_forward"), C, T, k.data_ptr(), K,
                                    direct ? 1 : 0, hp, K > 1 ? hout.data_ptr() : nullptr, stream_of(x)),
             "fir_stream_forward");
    return {y, hout};
}

// one small streaming chunk through cascade -> stateful direct FIR -> gain / clip in ONE launch (tfx_chunk_forward).
// Returns (y, new_state_x, new_state_y, new_hist); sos_cpu [K, 6] may have K = 0, kernel one tap.
std::tuple<Tensor, Tensor, Tensor, Tensor> chunk_op(const Tensor &x_in, const Tensor &sos_cpu, const OptTensor &state_x,
                                                    const OptTensor &state_y, const Tensor &kernel, const OptTensor &hist,
                                                    double gain, bool scale, bool clamp, int64_t precision)
{
    TORCH_CHECK(x_in.dim() == 2 && x_in.scalar_type() == at::kFloat, "chunk_forward: x must be float32 [C, T], got ", x_in.sizes());
    need_device(x_in, "x");
    // a chunk is usually a column window of a longer buffer: rows with unit stride are taken as they are (row pitch)
    const Tensor x = (x_in.stride(1) == 1 && x_in.stride(0) >= x_in.size(1)) ? x_in : x_in.contiguous();
    const Tensor sos = host_f64(sos_cpu, 6, "chunk_forward");
    TORCH_CHECK(sos.dim() == 2, "chunk_forward: sos must be [K, 6]");
    const Tensor k = taps_host(kernel, x);
    const int64_t C = x.size(0), T = x.size(1), K = sos.size(0), Kf = k.numel();
    TORCH_CHECK(tfx_chunk_supported(C, T, K, Kf), "chunk_forward: unsupported geometry C=", C, " T=", T, " K=", K, " taps=", Kf);
    Tensor kx, ky, hin;
    const double *sx = K ? state_ptr(state_x, {K, C, 2}, x, "state_x", kx) : nullptr;
    const double *sy = K ? state_ptr(state_y, {K, C, 2}, x, "state_y", ky) : nullptr;
    const float *hp = nullptr;
    if (hist.has_value() && hist->defined() && Kf > 1) {
        TORCH_CHECK(hist->dim() == 2 && hist->size(0) == C && hist->size(1) == Kf - 1, "chunk_forward: history must be [C, K-1] = [",
                    C, ", ", Kf - 1, "], got ", hist->sizes());
        hin = hist->to(x.device(), at::kFloat).contiguous();
        hp = hin.data_ptr<float>();
    }
    Tensor y = at::empty({C, T}, x.options());
    Tensor nsx = at::empty({K, C, 2}, x.options().dtype(at::kDouble));
    Tensor nsy = at::empty({K, C, 2}, x.options().dtype(at::kDouble));
    Tensor hout = at::empty({C, Kf - 1}, x.options());
    c10::hip::HIPGuard guard(x.get_device());
    check_rc(tfx_chunk_forward(x.data_ptr<float>(), C > 1 ? x.stride(0) : T, y.data_ptr<float>(), C, T, K ? sos.data_ptr<double>() : nullptr, K, sx, sy,
                               K ? nsx.data_ptr<double>() : nullptr, K ? nsy.data_ptr<double>() : nullptr, k.data_ptr<float>(), Kf,
                               hp, Kf > 1 ? hout.data_ptr<float>() : nullptr, gain, scale ? 1 : 0, clamp ? 1 : 0,
                               precision_or_default(precision), stream_of(x)),
             "chunk_forward");
    return {y, nsx, nsy, hout};
}

// ---------------------------------------------------------------------------------------------------
// `+` of branch outputs, Gain / Normalize passes, layout kernels
// ---------------------------------------------------------------------------------------------------
Tensor sum_op(at::TensorList tensors)
{
    TORCH_CHECK(!tensors.empty(), "sum_forward: need at least one tensor");
    std::vector<Tensor> ts;
    for (const Tensor &t : tensors) {
        need_device(t, "branch output");
        TORCH_CHECK(t.sizes() == tensors[0].sizes() && t.scalar_type() == tensors[0].scalar_type(),
                    "sum_forward: branch outputs differ in shape or dtype");
        ts.push_back(t.contiguous());
    }
    Tensor out = at::empty_like(ts[0]);
    c10::hip::HIPGuard guard(out.get_device());
    for (size_t i = 0; i < ts.size(); i += 15) {            // the kernel takes up to 16 inputs per launch
        std::vector<const void *> grp;
        if (i > 0) grp.push_back(out.data_ptr());
        for (size_t j = i; j < ts.size() && j < i + 15; ++j) grp.push_back(ts[j].data_ptr());
        check_rc(tfx_sum_forward(grp.data(), (int)grp.size(), out.data_ptr(), dtype_code(out, "sum_forward"), out.numel(),
                                 stream_of(out)),
                 "sum_forward");
    }
    return out;
}

// q-quantile (linear interpolation) of |x| over all elements, float64 [1] on the device: the percentile threshold as a radix select
Tensor quantile_abs_op(const Tensor &x, double q)
{
    need_device(x, "x");
    TORCH_CHECK(x.scalar_type() == at::kFloat, "quantile_abs: float32 signals only (got ", x.scalar_type(), ")");
    TORCH_CHECK(x.numel() >= 1, "quantile_abs: empty input");
    const Tensor xc = x.contiguous();
    Tensor out = at::empty({1}, xc.options().dtype(at::kDouble));
    c10::hip::HIPGuard guard(x.get_device());
    check_rc(tfx_quantile_abs(xc.data_ptr<float>(), xc.numel(), q, out.data_ptr<double>(), stream_of(x)), "quantile_abs");
    return out;
}

Tensor gain_op(const Tensor &x, double gain, bool clamp)
{
    need_device(x, "x");
    const Tensor xc = x.contiguous();
    Tensor y = at::empty_like(xc);
    c10::hip::HIPGuard guard(x.get_device());
    check_rc(tfx_gain_forward(xc.data_ptr(), y.data_ptr(), dtype_code(xc, "gain_forward"), xc.numel(), gain, clamp ? 1 : 0,
                              stream_of(x)),
             "gain_forward");
    return y;
}

Tensor stat_op(const Tensor &x, int64_t mode, bool per_row)
{
    need_device(x, "x");
    const Tensor xc = x.contiguous();
    const int64_t T = xc.dim() ? xc.size(-1) : 1, rows = T ? xc.numel() / T : 0;
    Tensor out = at::empty({per_row ? rows : 1}, x.options().dtype(at::kDouble));
    c10::hip::HIPGuard guard(x.get_device());
    check_rc(tfx_stat_forward(xc.data_ptr(), dtype_code(xc, "stat_forward"), rows, T, (int)mode, per_row ? 1 : 0,
                              out.data_ptr<double>(), stream_of(x)),
             "stat_forward");
    return out;
}

Tensor normalize_op(const Tensor &x, double peak, int64_t mode, bool per_row)
{
    need_device(x, "x");
    const Tensor xc = x.contiguous();
    const int64_t T = xc.dim() ? xc.size(-1) : 1, rows = T ? xc.numel() / T : 0;
    Tensor y = at::empty_like(xc);
    c10::hip::HIPGuard guard(x.get_device());
    check_rc(tfx_normalize_forward(xc.data_ptr(), y.data_ptr(), dtype_code(xc, "normalize_forward"), rows, T, (int)mode,
                                   per_row ? 1 : 0, peak, stream_of(x)),
             "normalize_forward");
    return y;
}

void deinterleave_into_op(const Tensor &frames, Tensor out, int64_t frame_base, double scale)
{
    need_device(frames, "frames");
    TORCH_CHECK(frames.dim() == 2 && (frames.scalar_type() == at::kFloat || frames.scalar_type() == at::kShort),
                "deinterleave_forward: expected [F, C] float32 or int16, got ", frames.sizes(), " ", frames.scalar_type());
    const Tensor fr = frames.contiguous();
    const int64_t F = fr.size(0), C = fr.size(1);
    TORCH_CHECK(out.dim() == 2 && out.size(0) == C && out.scalar_type() == at::kFloat && out.is_contiguous() &&
                    out.device() == fr.device(),
                "deinterleave_forward: out must be a contiguous float32 [C, F_total] tensor on the same device");
    c10::hip::HIPGuard guard(fr.get_device());
    check_rc(tfx_deinterleave_forward(fr.data_ptr(), fr.scalar_type() == at::kFloat ? 0 : 1, out.data_ptr(), F, C, out.size(1),
                                      frame_base, scale, stream_of(fr)),
             "deinterleave_forward");
}

Tensor deinterleave_op(const Tensor &frames, double scale)
{
    TORCH_CHECK(frames.dim() == 2, "deinterleave_forward: expected [F, C], got ", frames.sizes());
    Tensor out = at::empty({frames.size(1), frames.size(0)}, frames.options().dtype(at::kFloat));
    deinterleave_into_op(frames, out, 0, scale);
    return out;
}

Tensor interleave_op(const Tensor &x, int64_t frame_base, int64_t frames)
{
    need_device(x, "x");
    TORCH_CHECK(x.dim() == 2 && x.scalar_type() == at::kFloat, "interleave_forward: expected float32 [C, F], got ", x.sizes(), " ",
                x.scalar_type());
    const Tensor xc = x.contiguous();
    const int64_t C = xc.size(0), Ft = xc.size(1);
    const int64_t F = frames < 0 ? Ft - frame_base : frames;
    Tensor out = at::empty({F > 0 ? F : 0, C}, x.options());
    c10::hip::HIPGuard guard(x.get_device());
    check_rc(tfx_interleave_forward(xc.data_ptr(), out.data_ptr(), F, C, Ft, frame_base, stream_of(x)), "interleave_forward");
    return out;
}

// ---------------------------------------------------------------------------------------------------
// Meta kernels (shape / dtype inference: torch.compile, fake tensors)
// ---------------------------------------------------------------------------------------------------
std::tuple<Tensor, Tensor, Tensor> sos_meta(const Tensor &x, const Tensor &sos_cpu, const OptTensor &, const OptTensor &,
                                            std::optional<at::ScalarType> out_dtype, int64_t)
{
    const int64_t K = sos_cpu.size(0), C = x.size(0);
    Tensor st = at::empty({K, C, 2}, x.options().dtype(at::kDouble));
    return {at::empty(x.sizes(), x.options().dtype(out_type(x, out_dtype))), st, at::empty_like(st)};
}
std::tuple<Tensor, Tensor, Tensor, Tensor> sos_sections_meta(const Tensor &x, const Tensor &sos_cpu, const OptTensor &a,
                                                             const OptTensor &b, std::optional<at::ScalarType> out_dtype, int64_t p)
{
    auto r = sos_meta(x, sos_cpu, a, b, out_dtype, p);
    return {std::get<0>(r), std::get<1>(r), std::get<2>(r),
            at::empty({sos_cpu.size(0), x.size(0), x.size(1)}, x.options().dtype(out_type(x, out_dtype)))};
}
std::tuple<Tensor, Tensor, Tensor> bank_meta(const Tensor &x, const Tensor &banks, const OptTensor &, const OptTensor &,
                                             std::optional<at::ScalarType> out_dtype, int64_t)
{
    const int64_t NB = banks.size(0), K = banks.size(1), C = x.size(0);
    Tensor st = at::empty({K, NB * C, 2}, x.options().dtype(at::kDouble));
    return {at::empty({NB, C, x.size(1)}, x.options().dtype(out_type(x, out_dtype))), st, at::empty_like(st)};
}
std::tuple<Tensor, Tensor, Tensor> bank_sum_meta(const Tensor &x, const Tensor &banks, const OptTensor &, const OptTensor &, int64_t)
{
    const int64_t NB = banks.size(0), K = banks.size(1), C = x.size(0);
    Tensor st = at::empty({K, NB * C, 2}, x.options().dtype(at::kDouble));
    return {at::empty_like(x), st, at::empty_like(st)};
}
std::tuple<Tensor, Tensor, Tensor> biquad_meta(const Tensor &x, const Tensor &, double, double, const OptTensor &, const OptTensor &,
                                               std::optional<at::ScalarType> out_dtype, int64_t)
{
    Tensor st = at::empty({x.size(0), 2}, x.options().dtype(at::kDouble));
    return {at::empty(x.sizes(), x.options().dtype(out_type(x, out_dtype))), st, at::empty_like(st)};
}
Tensor fft_conv_meta(const Tensor &x, const Tensor &kernel, int64_t pad_left, int64_t pad_right)
{
    const int64_t tout = x.size(1) + pad_left + pad_right - kernel.numel() + 1;
    return at::empty({x.size(0), tout > 0 ? tout : 0}, x.options());
}

// CPU tensors: an explicit error instead of the dispatcher's "no kernel for backend CPU"

}  // namespace

TORCH_LIBRARY(torchfx_hip, m)
{
    m.def("sos_forward(Tensor x, Tensor sos_cpu, Tensor? state_x=None, Tensor? state_y=None, *, ScalarType? out_dtype=None, "
          "int precision=-1) -> (Tensor, Tensor, Tensor)");
    m.def("sos_forward_sections(Tensor x, Tensor sos_cpu, Tensor? state_x=None, Tensor? state_y=None, *, ScalarType? out_dtype=None, "
          "int precision=-1) -> (Tensor, Tensor, Tensor, Tensor)");
    m.def("sos_bank_forward(Tensor x, Tensor sos_banks_cpu, Tensor? state_x=None, Tensor? state_y=None, *, ScalarType? out_dtype=None, "
          "int precision=-1) -> (Tensor, Tensor, Tensor)");
    m.def("sos_bank_sum_forward(Tensor x, Tensor sos_banks_cpu, Tensor? state_x=None, Tensor? state_y=None, *, int precision=-1) "
          "-> (Tensor, Tensor, Tensor)");
    m.def("biquad_forward(Tensor x, Tensor b, float a1, float a2, Tensor? state_x=None, Tensor? state_y=None, *, "
          "ScalarType? out_dtype=None, int precision=-1) -> (Tensor, Tensor, Tensor)");
    m.def("delay_line_forward(Tensor(a) x, int delay_samples, float decay, float mix) -> Tensor(a)");
    m.def("fir_direct_forward(Tensor x, Tensor kernel) -> Tensor");
    m.def("fft_conv_forward(Tensor x, Tensor kernel, int pad_left, int pad_right) -> Tensor");
    m.def("fir_stream_forward(Tensor x, Tensor kernel, Tensor? hist, bool direct) -> (Tensor, Tensor)");
    m.def("chunk_forward(Tensor x, Tensor sos_cpu, Tensor? state_x, Tensor? state_y, Tensor kernel, Tensor? hist, float gain, "
          "bool scale, bool clamp, int precision=-1) -> (Tensor, Tensor, Tensor, Tensor)");
    m.def("sos_forward_ep(Tensor x, Tensor sos_cpu, Tensor? state_x, Tensor? state_y, float gain, bool clamp, int stat_mode, "
          "bool per_row, *, ScalarType? out_dtype=None, int precision=-1) -> (Tensor, Tensor, Tensor, Tensor)");
    m.def("fft_conv_forward_ep(Tensor x, Tensor kernel, int pad_left, int pad_right, float gain, bool clamp, int stat_mode, "
          "bool per_row) -> (Tensor, Tensor)");
    m.def("sos_fft_conv_forward(Tensor x, Tensor sos_cpu, Tensor kernel, int pad_left, int pad_right, bool sections=False, "
          "int force_block=0, float gain=1.0, bool clamp=False, int stat_mode=-1, bool per_row=False) -> (Tensor, Tensor, Tensor)");
    m.def("normalize_apply(Tensor x, Tensor stat, float peak, int mode, bool per_row) -> Tensor");
    m.def("sum_forward(Tensor[] tensors) -> Tensor");
    m.def("gain_forward(Tensor x, float gain, bool clamp) -> Tensor");
    m.def("quantile_abs(Tensor x, float q) -> Tensor");
    m.def("stat_forward(Tensor x, int mode, bool per_row) -> Tensor");
    m.def("normalize_forward(Tensor x, float peak, int mode, bool per_row) -> Tensor");
    m.def("deinterleave_forward(Tensor frames, float scale=3.0517578125e-05) -> Tensor");
    m.def("deinterleave_into(Tensor frames, Tensor(a!) out, int frame_base=0, float scale=3.0517578125e-05) -> ()");
    m.def("interleave_forward(Tensor x, int frame_base=0, int frames=-1) -> Tensor");
}

TORCH_LIBRARY_IMPL(torchfx_hip, CUDA, m)          // "CUDA" is the dispatch key of ROCm device tensors
{
    m.impl("sos_forward", sos_op);
    m.impl("sos_forward_sections", sos_sections_op);
    m.impl("sos_bank_forward", bank_op);
    m.impl("sos_bank_sum_forward", bank_sum_op);
    m.impl("biquad_forward", biquad_op);
    m.impl("delay_line_forward", delay_line_op);
    m.impl("fir_direct_forward", fir_direct_op);
    m.impl("fft_conv_forward", fft_conv_op);
    m.impl("fir_stream_forward", fir_stream_op);
    m.impl("chunk_forward", chunk_op);
    m.impl("sos_forward_ep", sos_ep_op);
    m.impl("fft_conv_forward_ep", fft_conv_ep_op);
    m.impl("sos_fft_conv_forward", sos_fft_conv_op);
    m.impl("normalize_apply", normalize_apply_op);
    m.impl("sum_forward", sum_op);
    m.impl("gain_forward", gain_op);
    m.impl("quantile_abs", quantile_abs_op);
    m.impl("stat_forward", stat_op);
    m.impl("normalize_forward", normalize_op);
    m.impl("deinterleave_forward", deinterleave_op);
    m.impl("deinterleave_into", deinterleave_into_op);
    m.impl("interleave_forward", interleave_op);
}

TORCH_LIBRARY_IMPL(torchfx_hip, Meta, m)
{
    m.impl("sos_forward", sos_meta);
    m.impl("sos_forward_sections", sos_sections_meta);
    m.impl("sos_bank_forward", bank_meta);
    m.impl("sos_bank_sum_forward", bank_sum_meta);
    m.impl("biquad_forward", biquad_meta);
    m.impl("fir_direct_forward", [](const Tensor &x, const Tensor &) { return at::empty_like(x); });
    m.impl("fft_conv_forward", fft_conv_meta);
    m.impl("fir_stream_forward", [](const Tensor &x, const Tensor &kernel, const OptTensor &, bool) {
        return std::make_tuple(at::empty_like(x), at::empty({x.size(0), kernel.numel() - 1}, x.options()));
    });
    m.impl("gain_forward", [](const Tensor &x, double, bool) { return at::empty_like(x); });
    m.impl("normalize_forward", [](const Tensor &x, double, int64_t, bool) { return at::empty_like(x); });
}

// No CPU branch, by design: every op of the namespace answers a host tensor with the same error (one boxed function
// registered for the CPU key of each op; per-namespace fallbacks are not supported by the dispatcher).
static void no_cpu_boxed(const c10::OperatorHandle &op, c10::DispatchKeySet, torch::jit::Stack *)
{
    TORCH_CHECK(false, "torchfx_amd: ", op.schema().name(),
                ": tensors must live on a ROCm device; this backend has no CPU path -- move them with .to('cuda').");
}
TORCH_LIBRARY_IMPL(torchfx_hip, CPU, m)
{
    for (const char *name : {"sos_forward", "sos_forward_sections", "sos_bank_forward", "sos_bank_sum_forward", "biquad_forward",
                             "delay_line_forward", "fir_direct_forward", "fft_conv_forward", "fir_stream_forward", "chunk_forward", "sos_forward_ep",
                             "fft_conv_forward_ep", "sos_fft_conv_forward", "normalize_apply", "sum_forward", "gain_forward", "quantile_abs", "stat_forward",
                             "normalize_forward", "deinterleave_forward", "deinterleave_into", "interleave_forward"})
        m.impl(name, torch::CppFunction::makeFromBoxedFunction<&no_cpu_boxed>());
}

// The reference's module surface (binding.cpp:83-96): exactly these three names and argument lists.
PYBIND11_MODULE(torchfx_ext, m)
{
    m.doc() = "torchfx native extension for MI355X (HIP kernels behind the reference's torchfx_ext interface)";
    m.def("biquad_forward",
          [](const Tensor &x, const Tensor &b, double a1, double a2, const OptTensor &state_x, const OptTensor &state_y) {
              if (!x.is_cuda()) return host::biquad_forward(x, b, a1, a2, state_x, state_y);      // binding.cpp:30-50 dispatches the same way
              return biquad_op(x, b, a1, a2, state_x, state_y, std::nullopt, -1);
          },
          "Biquad forward pass (x, b, a1, a2, state_x, state_y) -> (y, new_state_x, new_state_y)", py::arg("x"), py::arg("b"),
          py::arg("a1"), py::arg("a2"), py::arg("state_x"), py::arg("state_y"));
    m.def("sos_forward",
          [](const Tensor &x, const OptTensor &sos, const Tensor &sos_cpu, const OptTensor &state_x, const OptTensor &state_y) {
              // host tensors: binding.cpp:52-66 hands `sos` (the 2nd argument) to sos_forward_cpu; `sos_cpu` only when it is absent
              if (!x.is_cuda()) return host::sos_forward(x, (sos.has_value() && sos->defined() && !sos->is_cuda()) ? *sos : sos_cpu, state_x, state_y);
              // device tensors: `sos` is the reference's device copy of the coefficients (its sync-avoidance argument), unused here
              return sos_op(x, sos_cpu, state_x, state_y, std::nullopt, -1);
          },
          "SOS cascade forward pass (x, sos, sos_cpu, state_x, state_y) -> (y, new_state_x, new_state_y)", py::arg("x"),
          py::arg("sos"), py::arg("sos_cpu"), py::arg("state_x"), py::arg("state_y"));
    m.def("delay_line_forward",
          [](const Tensor &x, int64_t delay_samples, double decay, double mix) {
              if (!x.is_cuda()) return host::delay_line_forward(x, delay_samples, decay, mix);    // binding.cpp:68-81
              return delay_line_op(x, delay_samples, decay, mix);
          },
          "Delay line forward pass (x, delay_samples, decay, mix) -> y", py::arg("x"), py::arg("delay_samples"), py::arg("decay"),
          py::arg("mix"));
}
