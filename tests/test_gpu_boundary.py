"""GPU parity -- the drop-in boundary called the way the reference calls it.

`/root/reference/src/torchfx/_ops.py:57-116,119-176,179-191` is the only caller of the reference's native module:
it upcasts the signal to float64, builds zero states `[K, C, 2]`, moves the SOS to the signal's device and calls

    _ext.sos_forward(x_f64, sos_device, sos_cpu, sx, sy)
    _ext.biquad_forward(x_f64, b_f64, a1_f64, a2_f64, sx, sy)
    _ext.delay_line_forward(x, delay_samples, decay, mix)

These tests make exactly those calls on OUR compiled pybind module (`torchfx_amd.native.load()`,
csrc/ext/torchfx_ext.cpp: the three lambdas with their defaults `precision -1`, `out_dtype nullopt`) and compare
with the fixtures generated from the real reference (tests/golden, oracle/make_golden.py): float64 out, the state
layouts, inputs untouched, errors for host tensors."""
import numpy as np
import pytest
import torch

from tests.gpu_common import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


def _mod():
    from torchfx_amd import native
    return native.load()


def _as_ops_calls_sos(x, sos, state_x=None, state_y=None):
    """`torchfx._ops.parallel_iir_forward` (_ops.py:119-176), restated around our module."""
    C = x.shape[0] if x.ndim >= 2 else 1
    K = sos.shape[0]
    device, dtype = x.device, torch.float64
    if state_x is None:
        state_x = torch.zeros(K, C, 2, device=device, dtype=dtype)
    if state_y is None:
        state_y = torch.zeros(K, C, 2, device=device, dtype=dtype)
    x_f64 = x if x.dtype == dtype else x.to(dtype=dtype)
    sos_device = sos if (sos.device == device and sos.dtype == dtype) else sos.to(device=device, dtype=dtype)
    sx = state_x if (state_x.device == device and state_x.dtype == dtype) else state_x.to(device=device, dtype=dtype)
    sy = state_y if (state_y.device == device and state_y.dtype == dtype) else state_y.to(device=device, dtype=dtype)
    sos_cpu = sos.detach().to(dtype=dtype, device="cpu")
    return _mod().sos_forward(x_f64, sos_device, sos_cpu, sx, sy), (x_f64, sos_device, sos_cpu, sx, sy)


def test_sos_forward_called_like_the_reference_ops(golden):
    g = golden("iir_cfg2_sections")
    x = dev(g["x"])
    sos = torch.from_numpy(g["sos"])
    (y, sx, sy), args = _as_ops_calls_sos(x, sos)
    assert y.dtype == torch.float64 and y.shape == x.shape and y.is_cuda          # float64 out, like sos_forward_cuda
    assert sx.shape == sy.shape == (4, 4, 2) and sx.dtype == sy.dtype == torch.float64
    close(y, g["y_sections"][-1], TOL_IIR_F64OUT, "cascade output (float64) vs the reference's last section")
    close(y.to(torch.float32), g["y"], TOL_IIR_F32OUT, "downcast as iir.py:176")
    close(sx, g["state_x"], TOL_STATE, "state_x [K, C, 2]")
    close(sy, g["state_y"], TOL_STATE, "state_y [K, C, 2]")
    # pure function: the zero states and the signal it was handed are untouched, outputs are new storage
    assert float(args[3].abs().max()) == 0.0 and float(args[4].abs().max()) == 0.0
    assert torch.equal(args[0].to(torch.float32), x)
    assert y.data_ptr() != args[0].data_ptr() and sx.data_ptr() != args[3].data_ptr()
    # section by section through the same entry point: K = 1 calls chained by hand == the reference's taps
    cur = x.double()
    for s in range(4):
        (cur, _, _), _ = _as_ops_calls_sos(cur, sos[s:s + 1])
        close(cur, g["y_sections"][s], TOL_IIR_F64OUT, f"section {s}")


def test_sos_forward_states_carry_and_shapes(golden):
    g = golden("iir_chunked")
    sos = torch.from_numpy(g["sos"])
    x = dev(g["x"])
    (y1, sx, sy), _ = _as_ops_calls_sos(x[:, :1024], sos)
    close(y1, g["y1"], TOL_IIR_F64OUT, "chunk 1")
    close(sx, g["mid_state_x"], TOL_STATE, "mid state x")
    (y2, sx2, sy2), _ = _as_ops_calls_sos(x[:, 1024:], sos, sx, sy)
    close(y2, g["y2"], TOL_IIR_F64OUT, "chunk 2")
    close(sx2, g["state_x"], TOL_STATE, "end state x")
    close(sy2, g["state_y"], TOL_STATE, "end state y")
    s = golden("iir_shapes")                                  # caller-supplied, mutually inconsistent states (T = 500, 1, 2)
    (ys, nsx, nsy), _ = _as_ops_calls_sos(dev(s["xs"]), torch.from_numpy(s["s_sos"]), dev(s["isx"]), dev(s["isy"]))
    close(ys, s["ys"], TOL_IIR_F64OUT, "given states")
    close(nsx, s["nsx"], TOL_STATE, "nsx")
    close(nsy, s["nsy"], TOL_STATE, "nsy")
    for T in (1, 2):
        (yt, tx, ty), _ = _as_ops_calls_sos(dev(s["xs"][:, :T].copy()), torch.from_numpy(s["s_sos"]), dev(s["isx"]), dev(s["isy"]))
        close(yt, s[f"t{T}_y"], TOL_IIR_F64OUT, f"T={T}")
        close(tx, s[f"t{T}_sx"], TOL_STATE, f"T={T} sx")
        close(ty, s[f"t{T}_sy"], TOL_STATE, f"T={T} sy")


def test_biquad_forward_called_like_the_reference_ops(golden):
    """_ops.py:57-116: b as a float64 tensor on the signal's device, a1 / a2 as Python floats, states [C, 2]."""
    s = golden("iir_shapes")
    row = s["bq_sos"][0]
    x = dev(s["x1d"]).reshape(1, -1)
    dtype = torch.float64
    b = torch.tensor(row[:3], dtype=dtype, device=x.device)
    sx0 = torch.zeros(1, 2, device=x.device, dtype=dtype)
    sy0 = torch.zeros(1, 2, device=x.device, dtype=dtype)
    y, sx, sy = _mod().biquad_forward(x.to(dtype), b, float(row[4]), float(row[5]), sx0, sy0)
    assert y.dtype == dtype and sx.shape == sy.shape == (1, 2)
    close(y.to(torch.float32)[0], s["y1d"], TOL_IIR_F32OUT, "biquad y")
    close(sx, s["bq_sx"][0], TOL_STATE, "biquad state_x [C, 2]")
    close(sy, s["bq_sy"][0], TOL_STATE, "biquad state_y [C, 2]")
    assert float(sx0.abs().max()) == 0.0 and float(sy0.abs().max()) == 0.0
    # b on the host is accepted too (the reference's CUDA path pulls it to the host itself, binding.cpp:40-43)
    y2, _, _ = _mod().biquad_forward(x.to(dtype), b.cpu(), float(row[4]), float(row[5]), sx0, sy0)
    assert torch.equal(y, y2)


def test_delay_line_forward_called_like_the_reference_ops(golden):
    g = golden("delay")
    x = dev(g["x"])
    y = _mod().delay_line_forward(x, 100, 0.5, 0.3)               # _ops.py:179-191
    assert y.dtype == x.dtype and y.shape == x.shape
    close(y, g["y"], 1e-7, "delay line")
    short = dev(np.zeros((2, 50), np.float32))
    assert _mod().delay_line_forward(short, 100, 0.5, 0.3).data_ptr() == short.data_ptr()    # delay_cpu.cpp:61-63: the input itself


def test_boundary_errors_are_runtime_errors():
    m = _mod()
    x = torch.zeros(2, 16, dtype=torch.float64)
    sos = torch.tensor([[1.0, 0, 0, 1, 0, 0]], dtype=torch.float64)
    z = torch.zeros(1, 2, 2, dtype=torch.float64)
    yh, _, _ = m.sos_forward(x, sos, sos, z, z)                    # host tensors: the module's own host branch (binding.cpp:52-66)
    assert torch.equal(yh, x)
    with pytest.raises(RuntimeError):
        m.sos_forward(x.to(DEV), sos.to(DEV), torch.zeros(1, 5, dtype=torch.float64), z.to(DEV), z.to(DEV))    # not [K, 6]
