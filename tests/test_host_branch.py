"""The compiled boundary module on HOST tensors: `torchfx_ext.{sos,biquad,delay_line}_forward` dispatch on
`x.is_cuda()` like the reference's module (`src/torchfx/_csrc/binding.cpp:30-81`); the host branch is the module's own
C++ (`torchfx_amd/csrc/ext/host_branch.h`), compared here with the fixtures generated from the real reference.  It serves
the drop-in boundary only: the dispatcher ops and the torchfx_amd package stay device-only (tests/test_capi_exports.py),
and bench.py / the -m gpu tests never run it."""
import numpy as np
import pytest
import torch


def _mod():
    from torchfx_amd import native
    return native.load()


def _err(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max())


def test_cfg1_literally(golden):
    """BASELINE.json configs[0]: single LoButterworth order-4, 1 ch x 1 s @ 48 kHz float32 on the CPU path."""
    g = golden("iir_cfg1")
    y, sx, sy = _mod().sos_forward(torch.from_numpy(g["x"]), None, torch.from_numpy(g["sos"]), None, None)
    # float64 out whatever x is (iir_cpu.cpp: y = empty_like(x_f64)); the reference's Python downcasts (_ops.py:149-176)
    assert y.dtype == torch.float64 and np.array_equal(y.to(torch.float32).numpy(), g["y"])
    assert _err(sx, g["state_x"]) <= 2e-10 and _err(sy, g["state_y"]) <= 2e-10
    # the coefficients come from `sos` (2nd argument) on host tensors, like binding.cpp:52-66; `sos_cpu` when it is absent
    wrong = torch.zeros_like(torch.from_numpy(g["sos"]))
    y2, _, _ = _mod().sos_forward(torch.from_numpy(g["x"]), torch.from_numpy(g["sos"]), wrong, None, None)
    assert torch.equal(y2, y)


def test_cfg2_section_by_section(golden):
    g = golden("iir_cfg2_sections")
    m = _mod()
    sos = torch.from_numpy(g["sos"])
    cur = torch.from_numpy(g["x"]).double()
    for k in range(sos.shape[0]):
        cur, _, _ = m.sos_forward(cur, None, sos[k:k + 1], None, None)
        assert _err(cur, g["y_sections"][k]) <= 2e-11 * max(1.0, float(np.abs(g["y_sections"][k]).max())), k
    y, sx, sy = m.sos_forward(torch.from_numpy(g["x"]), None, sos, None, None)
    assert np.array_equal(y.to(torch.float32).numpy(), g["y"])
    assert _err(sx, g["state_x"]) <= 2e-10 and _err(sy, g["state_y"]) <= 2e-10


def test_chunked_state_carry_and_inputs_untouched(golden):
    g = golden("iir_chunked")
    m = _mod()
    sos = torch.from_numpy(g["sos"])
    x = torch.from_numpy(g["x"])
    y1, sx, sy = m.sos_forward(x[:, :1024].contiguous(), None, sos, None, None)
    assert _err(y1, g["y1"]) <= 2e-7 and _err(sx, g["mid_state_x"]) <= 2e-10
    keep = sx.clone()
    y2, sx2, sy2 = m.sos_forward(x[:, 1024:].contiguous(), None, sos, sx, sy)
    assert torch.equal(sx, keep)                                      # const Tensor& in, fresh tensors out (iir_cpu.cpp:72-73)
    assert _err(y2, g["y2"]) <= 2e-7 and _err(sx2, g["state_x"]) <= 2e-10 and _err(sy2, g["state_y"]) <= 2e-10


def test_biquad_and_delay():
    import scipy.signal as sg
    m = _mod()
    x = torch.from_numpy(np.random.default_rng(0).standard_normal((3, 5000)))
    b, a = sg.butter(2, 0.15)
    y, sx, sy = m.biquad_forward(x, torch.from_numpy(b), float(a[1]), float(a[2]), None, None)
    assert _err(y, sg.lfilter(b, a, x.numpy(), axis=-1)) <= 1e-12 and sx.shape == (3, 2) and sy.dtype == torch.float64
    d = m.delay_line_forward(x.float(), 100, 0.5, 0.3)
    ex = x.float().clone(); ex[:, 100:] += 0.15 * x.float()[:, :-100]
    assert _err(d, ex) <= 1e-6
    short = x[:, :50]
    assert m.delay_line_forward(short, 100, 0.5, 0.3) is short          # delay_cpu.cpp:61-63: the input itself
    with pytest.raises(RuntimeError):
        m.sos_forward(x, None, torch.zeros(2, 5, dtype=torch.float64), None, None)      # not [K, 6]
    with pytest.raises(RuntimeError):
        m.sos_forward(x, None, torch.tensor([[1., 0, 0, 1, 0, 0]]).double(), torch.zeros(1, 4, 2).double(), None)   # state shape


def test_the_package_itself_stays_device_only():
    from torchfx_amd import torchfx_ext as E
    with pytest.raises(RuntimeError, match="no CPU path"):
        E.sos_forward(torch.zeros(1, 8), None, torch.tensor([[1., 0, 0, 1, 0, 0]]), None, None)
