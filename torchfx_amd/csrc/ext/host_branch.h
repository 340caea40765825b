// host_branch.h -- the HOST-tensor branch of the reference's three module entry points.
//
// The reference's module dispatches on x.is_cuda() (src/torchfx/_csrc/binding.cpp:30-81): device tensors go to its CUDA
// kernels, host tensors to src/torchfx/_csrc/cpu/{iir_cpu,delay_cpu}.cpp.  A drop-in for that module has to answer host
// tensors too, or every CPU caller of the reference breaks (its tests/test_ops_dispatch.py:54-127 run on the CPU).  This
// file is that branch, written for this module: Direct Form I over raw row pointers, rows spread over at::parallel_for
// (the reference uses an OpenMP loop over channels, iir_cpu.cpp:106), the GIL released while it runs.  It is reached
// ONLY through the pybind entry points below (`torchfx_ext.sos_forward / biquad_forward / delay_line_forward` called with
// host tensors); the dispatcher ops (torch.ops.torchfx_hip.*), the torchfx_amd Python package, bench.py and every -m gpu
// test stay device-only, and nothing here touches the test oracle.
#pragma once
#include <torch/extension.h>

#include <ATen/Parallel.h>

#include <optional>
#include <tuple>
#include <vector>

namespace host {

using at::Tensor;
using OptT = std::optional<Tensor>;

inline Tensor state_or_zeros(const OptT &s, at::IntArrayRef shape, const char *what)
{
    if (!s.has_value() || !s->defined()) return at::zeros(shape, at::TensorOptions().dtype(at::kDouble));
    TORCH_CHECK(s->sizes() == shape, what, " must have shape ", shape, ", got ", s->sizes());
    return s->detach().to(at::kCPU, at::kDouble).contiguous().clone();          // inputs are never modified (iir_cpu.cpp:72-73)
}

// K sections over every row: y_s[n] = b0 v + b1 v[n-1] + b2 v[n-2] - a1 y_s[n-1] - a2 y_s[n-2], v = y_{s-1}[n]
// (iir_cpu.cpp:132-147).  `co` = K rows of (b0, b1, b2, a1, a2); states [K, C, 2] in the reference's layout
// (state_x[s, c] = [x_s[n-1], x_s[n-2]], state_y likewise: iir_cpu.cpp:125-130), updated in place.
inline void df1_rows(const double *x, double *y, int64_t C, int64_t T, const double *co, int64_t K, double *sx, double *sy)
{
    at::parallel_for(0, C, 1, [&](int64_t c0, int64_t c1) {
        std::vector<double> hx1(K), hx2(K), hy1(K), hy2(K);
        for (int64_t c = c0; c < c1; ++c) {
            for (int64_t s = 0; s < K; ++s) {
                hx1[s] = sx[(s * C + c) * 2]; hx2[s] = sx[(s * C + c) * 2 + 1];
                hy1[s] = sy[(s * C + c) * 2]; hy2[s] = sy[(s * C + c) * 2 + 1];
            }
            const double *xr = x + c * T;
            double *yr = y + c * T;
            for (int64_t n = 0; n < T; ++n) {
                double v = xr[n];
                for (int64_t s = 0; s < K; ++s) {
                    const double *q = co + 5 * s;
                    const double out = q[0] * v + q[1] * hx1[s] + q[2] * hx2[s] - q[3] * hy1[s] - q[4] * hy2[s];
                    hx2[s] = hx1[s]; hx1[s] = v;
                    hy2[s] = hy1[s]; hy1[s] = out;
                    v = out;
                }
                yr[n] = v;
            }
            for (int64_t s = 0; s < K; ++s) {
                sx[(s * C + c) * 2] = hx1[s]; sx[(s * C + c) * 2 + 1] = hx2[s];
                sy[(s * C + c) * 2] = hy1[s]; sy[(s * C + c) * 2 + 1] = hy2[s];
            }
        }
    });
}

// sos_forward on host tensors (binding.cpp:52-66 -> sos_forward_cpu, iir_cpu.cpp:64-159)
inline std::tuple<Tensor, Tensor, Tensor> sos_forward(const Tensor &x_in, const Tensor &sos_in, const OptT &state_x, const OptT &state_y)
{
    TORCH_CHECK(!x_in.is_cuda(), "host branch called with a device tensor");
    TORCH_CHECK(x_in.dim() == 2, "sos_forward: x must be [C, T], got ", x_in.sizes());
    TORCH_CHECK(x_in.scalar_type() == at::kFloat || x_in.scalar_type() == at::kDouble, "sos_forward: expected a float32 or float64 tensor");
    const Tensor sos = sos_in.detach().to(at::kCPU, at::kDouble).contiguous();
    TORCH_CHECK(sos.dim() == 2 && sos.size(1) == 6, "sos_forward: sos must be [K, 6], got ", sos.sizes());
    const int64_t C = x_in.size(0), T = x_in.size(1), K = sos.size(0);
    const Tensor x = x_in.detach().to(at::kDouble).contiguous();
    Tensor sx = state_or_zeros(state_x, {K, C, 2}, "state_x"), sy = state_or_zeros(state_y, {K, C, 2}, "state_y");
    Tensor y = at::empty({C, T}, x.options());
    std::vector<double> co((size_t)(5 * K));
    const double *sp = sos.data_ptr<double>();
    for (int64_t s = 0; s < K; ++s) {                      // a0 is not read (iir_cpu.cpp:86)
        co[5 * s] = sp[6 * s]; co[5 * s + 1] = sp[6 * s + 1]; co[5 * s + 2] = sp[6 * s + 2];
        co[5 * s + 3] = sp[6 * s + 4]; co[5 * s + 4] = sp[6 * s + 5];
    }
    {
        pybind11::gil_scoped_release nogil;
        df1_rows(x.data_ptr<double>(), y.data_ptr<double>(), C, T, co.data(), K, sx.data_ptr<double>(), sy.data_ptr<double>());
    }
    return {y, sx, sy};          // float64 whatever x is, like iir_cpu.cpp (y = empty_like(x_f64)); the caller downcasts (_ops.py:149-176)
}

// biquad_forward on host tensors (binding.cpp:30-50 -> biquad_forward_cpu, iir_cpu.cpp:10-62): states [C, 2]
inline std::tuple<Tensor, Tensor, Tensor> biquad_forward(const Tensor &x_in, const Tensor &b, double a1, double a2, const OptT &state_x,
                                                         const OptT &state_y)
{
    TORCH_CHECK(!x_in.is_cuda(), "host branch called with a device tensor");
    TORCH_CHECK(x_in.dim() == 2, "biquad_forward: x must be [C, T], got ", x_in.sizes());
    TORCH_CHECK(x_in.scalar_type() == at::kFloat || x_in.scalar_type() == at::kDouble, "biquad_forward: expected a float32 or float64 tensor");
    const Tensor bh = b.detach().to(at::kCPU, at::kDouble).contiguous();
    TORCH_CHECK(bh.numel() == 3, "biquad_forward: b must hold 3 coefficients");
    const int64_t C = x_in.size(0), T = x_in.size(1);
    const Tensor x = x_in.detach().to(at::kDouble).contiguous();
    Tensor sx = state_or_zeros(state_x, {C, 2}, "state_x"), sy = state_or_zeros(state_y, {C, 2}, "state_y");
    Tensor y = at::empty({C, T}, x.options());
    const double co[5] = {bh.data_ptr<double>()[0], bh.data_ptr<double>()[1], bh.data_ptr<double>()[2], a1, a2};
    {
        pybind11::gil_scoped_release nogil;
        df1_rows(x.data_ptr<double>(), y.data_ptr<double>(), C, T, co, 1, sx.data_ptr<double>(), sy.data_ptr<double>());
    }
    return {y, sx, sy};          // float64 whatever x is, like iir_cpu.cpp (y = empty_like(x_f64)); the caller downcasts (_ops.py:149-176)
}

// delay_line_forward on host tensors (binding.cpp:68-81 -> delay_cpu.cpp:17-66): y[n] = x[n] + mix * decay * x[n - delay];
// a signal not longer than the delay is returned as it is (the same tensor, delay_cpu.cpp:61-63)
inline Tensor delay_line_forward(const Tensor &x, int64_t delay, double decay, double mix)
{
    TORCH_CHECK(!x.is_cuda(), "host branch called with a device tensor");
    TORCH_CHECK(x.scalar_type() == at::kFloat || x.scalar_type() == at::kDouble, "delay_line_forward: expected a float32 or float64 tensor");
    const int64_t T = x.dim() ? x.size(-1) : 1;
    if (T <= delay) return x;
    const Tensor xc = x.contiguous();
    const int64_t rows = xc.numel() / T;
    Tensor y = at::empty_like(xc);
    auto run = [&](auto *in, auto *out, auto coeff) {
        pybind11::gil_scoped_release nogil;
        at::parallel_for(0, rows, 1, [&](int64_t r0, int64_t r1) {
            for (int64_t r = r0; r < r1; ++r) {
                const auto *a = in + r * T;
                auto *o = out + r * T;
                for (int64_t n = 0; n < delay; ++n) o[n] = a[n];
                for (int64_t n = delay; n < T; ++n) o[n] = a[n] + coeff * a[n - delay];
            }
        });
    };
    if (xc.scalar_type() == at::kFloat) run(xc.data_ptr<float>(), y.data_ptr<float>(), (float)(mix * decay));
    else run(xc.data_ptr<double>(), y.data_ptr<double>(), mix * decay);
    return y;
}

}  // namespace host
