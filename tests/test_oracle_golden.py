"""The CPU oracle against the golden vectors generated from the REAL reference
(oracle/make_golden.py).  This is what pins the oracle on machines where /root/reference
does not exist.  Tolerances: IIR float64 4e-12 relative to the data scale (the reference is
built with -ffast-math, so its own summation order is not fixed), bit-level (1e-7) after the
float32 cast; FIR / FFT 2e-5 (reference tests use 1e-4: tests/test_fir.py:90, test_fftconv.py:77)."""
import numpy as np
import pytest

from oracle import oracle as O


def close(got, exp, tol):
    exp = np.asarray(exp, dtype=np.float64)
    scale = max(1.0, float(np.abs(exp).max())) if exp.size else 1.0
    err = float(np.abs(np.asarray(got, dtype=np.float64) - exp).max()) if exp.size else 0.0
    assert err <= tol * scale, f"max err {err} > {tol * scale}"


def test_cfg1_lobutterworth_1ch_1s(golden):
    g = golden("iir_cfg1")
    y, sx, sy = O.iir_module_forward(g["x"], g["sos"])
    assert y.dtype == np.float32
    close(y, g["y"], 1e-7)
    close(sx, g["state_x"], 4e-12)
    close(sy, g["state_y"], 4e-12)


def test_cfg2_section_by_section(golden):
    g = golden("iir_cfg2_sections")
    y, sx, sy, sec = O.sos_forward(g["x"], g["sos"], sections=True)
    close(sec, g["y_sections"], 4e-12)
    close(y.astype(np.float32), g["y"], 1e-7)
    close(sx, g["state_x"], 4e-12)
    close(sy, g["state_y"], 4e-12)
    # reference invariant: section s+1's input history is section s's output history
    assert np.array_equal(sx[1:], sy[:-1])


def test_chunked_equals_contiguous(golden):
    g = golden("iir_chunked")
    y1, sx, sy = O.sos_forward(g["x"][:, :1024], g["sos"])
    close(y1, g["y1"], 4e-12)
    close(sx, g["mid_state_x"], 4e-12)
    y2, sx, sy = O.sos_forward(g["x"][:, 1024:], g["sos"], sx, sy)
    close(y2, g["y2"], 4e-12)
    close(sx, g["state_x"], 4e-12)
    close(sy, g["state_y"], 4e-12)
    yc, _, _ = O.sos_forward(g["x"], g["sos"])
    close(np.concatenate([y1, y2], axis=1), yc, 1e-13)


@pytest.mark.parametrize("name", ["hicheby1_20", "hibutter_20_o5", "lobutter_40_o8", "ellip_o12",
                                  "notch_q30", "butter_o20"])
def test_ill_conditioned_filters(golden, name):
    g = golden("iir_hard")
    y, sx, sy = O.iir_module_forward(g["x"], g[name + "_sos"])
    close(y, g[name + "_y"], 1e-6)
    close(sy, g[name + "_sy"], 1e-9)


def test_states_shapes_and_edges(golden):
    g = golden("iir_shapes")
    y, sx, sy = O.sos_forward(g["xs"], g["s_sos"], g["isx"], g["isy"])
    close(y, g["ys"], 4e-12)
    close(sx, g["nsx"], 4e-12)
    close(sy, g["nsy"], 4e-12)
    for t in (1, 2, 3):
        y, sx, sy = O.sos_forward(g["xs"][:, :t], g["s_sos"], g["isx"], g["isy"])
        close(y, g[f"t{t}_y"], 4e-12)
        close(sx, g[f"t{t}_sx"], 4e-12)
        close(sy, g[f"t{t}_sy"], 4e-12)
    y, _, _ = O.iir_module_forward(g["x1d"][None], g["bq_sos"])
    close(y[0], g["y1d"], 1e-7)
    y, _, _ = O.iir_module_forward(g["x3d"].reshape(6, -1), g["lr_sos"])
    close(y.reshape(g["y3d"].shape), g["y3d"], 4e-12)


@pytest.mark.parametrize("K", [5, 32, 1024])
def test_fir_direct_and_fft(golden, K):
    g = golden("fir")
    close(O.fir_direct(g["x"], g[f"k{K}"]), g[f"direct{K}"], 2e-5)
    close(O.fir_forward(g["x"], g[f"k{K}"], "fft"), g[f"fft{K}"], 2e-5)


def test_fir_short_signals(golden):
    g = golden("fir")
    close(O.fir_forward(g["xs"], g["ks"], "fft"), g["ys_fft"], 2e-5)
    close(O.fir_forward(g["xs"], g["ks"], "direct"), g["ys_direct"], 2e-5)
    close(O.fir_forward(g["xt"], g["kt"], "fft"), g["yt_fft"], 4e-12)       # T < K, float64
    close(O.fir_forward(g["xt"], g["kt"], "direct"), g["yt_direct"], 4e-12)


def test_fftconv(golden):
    g = golden("fftconv")
    for K in (64, 4097):
        close(O.fft_conv1d(g["x"], g[f"k{K}"], (K - 1, 0)), g[f"y{K}"], 2e-5)
    close(O.fft_conv1d(g["x"], g["k16"], (8, 7)), g["y16_pad87"], 2e-5)
    with pytest.raises(RuntimeError, match="kernel size"):
        O.fft_conv1d(g["x"][:, :10], g["k16"])
    with pytest.raises(RuntimeError, match="Block ratio"):
        O.fft_conv1d(g["x"], g["k16"], block_ratio=0.5)


def reverb_ir():
    K = 65536
    ir = np.random.default_rng(0).standard_normal(K) * np.exp(-np.arange(K) / 8000.0)
    return (ir / np.abs(ir).sum()).astype(np.float32)


def test_fftconv_65536_taps(golden):
    g = golden("fftconv")
    kf = reverb_ir()[::-1].copy()
    close(O.fft_conv1d(g["x_long"], kf, (65535, 0)), g["y_long"], 2e-5)


def test_chain_and_parallel(golden):
    g = golden("chain")
    y = O.chain_forward(g["x"], g["sos_run1"], [g["fir_k"]])
    y, _, _ = O.iir_module_forward(y, g["sos_run2"])
    close(y, g["y_chain"], 2e-5)
    a, _, _ = O.iir_module_forward(g["x"], g["p1_sos"])
    b, _, _ = O.iir_module_forward(g["x"], g["p2_sos"])
    close(a + b, g["y_par"], 1e-6)
    close(O.chain_forward(g["xc"], g["c_sos"], [g["c_fir"], g["c_ir"]]), g["yc"], 2e-5)


def test_delay(golden):
    g = golden("delay")
    close(O.delay_line(g["x"], 100, 0.5, 0.3), g["y"], 1e-7)


def test_chain_with_iir_gain_staged_like_the_reference(golden):
    """cascade with +12 dB shelf / +9 dB high-Q peak / 80 Hz high-pass -> FIR-257 -> FIR-2049, staged."""
    g = golden("chain_gain")
    close(O.chain_forward(g["x"], g["sos"], [g["fir"], g["ir"]]), g["y"], 2e-5)
    assert float(np.abs(g["y"]).max()) > 1.5          # the fixture really has gain
