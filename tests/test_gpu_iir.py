"""GPU parity -- SOS cascade kernel: golden vectors, section taps, states, segmentation, dtypes, non-finite input.
Tolerances and helpers: tests/gpu_common.py."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.gpu_common import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


def test_library_loaded_and_device():
    from torchfx_amd import _lib
    lib = _lib.load()
    assert lib.tfx_version() >= 100
    import ctypes
    name = ctypes.create_string_buffer(64)
    cus = ctypes.c_int(0)
    assert lib.tfx_device_info(name, 64, ctypes.byref(cus)) == 0
    assert name.value.decode().startswith("gfx950"), name.value
    assert cus.value >= 200


def test_cfg1_golden(golden, sos_variant):
    g = golden("iir_cfg1")
    y, sx, sy = ext().sos_forward(dev(g["x"]), None, torch.from_numpy(g["sos"]), None, None)
    assert y.dtype == torch.float32
    close(y, g["y"], TOL_IIR_F32OUT, "y")
    close(sx, g["state_x"], TOL_STATE, "sx")
    close(sy, g["state_y"], TOL_STATE, "sy")


def test_cfg2_section_by_section_golden(golden, sos_variant):
    """IIR compared after EVERY section (north_star), against the reference's own
    section-by-section float64 outputs."""
    g = golden("iir_cfg2_sections")
    sos = torch.from_numpy(g["sos"])
    # (a) fused kernel with its per-section taps, float64 in/out
    y, sx, sy, sec = ext().sos_forward(dev(g["x"].astype(np.float64)), None, sos, None, None,
                                       return_sections=True)
    for k in range(g["sos"].shape[0]):
        close(sec[k], g["y_sections"][k], TOL_IIR_F64OUT, f"section {k} (fused taps)")
    close(y, g["y_sections"][-1], TOL_IIR_F64OUT, "final")
    close(sx, g["state_x"], TOL_STATE, "sx")
    close(sy, g["state_y"], TOL_STATE, "sy")
    # (b) one launch per section through the public op, like the fixture was generated
    cur = dev(g["x"].astype(np.float64))
    for k in range(g["sos"].shape[0]):
        cur, _, _ = ext().sos_forward(cur, None, sos[k:k + 1], None, None)
        close(cur, g["y_sections"][k], TOL_IIR_F64OUT, f"section {k} (staged)")
    # (c) production path: float32 in / float32 out
    y32, _, _ = ext().sos_forward(dev(g["x"]), None, sos, None, None)
    close(y32, g["y"], TOL_IIR_F32OUT, "f32 out")


def test_chunked_state_carry_golden(golden, sos_variant):
    g = golden("iir_chunked")
    sos = torch.from_numpy(g["sos"])
    x = dev(g["x"])
    y1, sx, sy = ext().sos_forward(x[:, :1024].contiguous(), None, sos, None, None)
    close(y1, g["y1"], TOL_IIR_F64OUT, "y1")
    close(sx, g["mid_state_x"], TOL_STATE, "mid sx")
    close(sy, g["mid_state_y"], TOL_STATE, "mid sy")
    y2, sx, sy = ext().sos_forward(x[:, 1024:].contiguous(), None, sos, sx, sy)
    close(y2, g["y2"], TOL_IIR_F64OUT, "y2")
    close(sx, g["state_x"], TOL_STATE, "sx")
    close(sy, g["state_y"], TOL_STATE, "sy")


@pytest.mark.parametrize("name", ["hicheby1_20", "hibutter_20_o5", "lobutter_40_o8", "ellip_o12",
                                  "notch_q30", "butter_o20"])
def test_ill_conditioned_golden(golden, name, sos_variant):
    g = golden("iir_hard")
    y, sx, sy = ext().sos_forward(dev(g["x"]), None, torch.from_numpy(g[name + "_sos"]), None, None)
    close(y, g[name + "_y"], 2.5e-7, name)      # pole radius ~0.999: allow 2 ulp
    close(sy, g[name + "_sy"], 1e-8, name + " sy")


def test_states_edges_golden(golden, sos_variant):
    g = golden("iir_shapes")
    sos = torch.from_numpy(g["s_sos"])
    isx, isy = dev(g["isx"]), dev(g["isy"])
    y, sx, sy = ext().sos_forward(dev(g["xs"]), None, sos, isx, isy)
    close(y, g["ys"], TOL_IIR_F64OUT, "y")
    close(sx, g["nsx"], TOL_STATE, "sx")
    close(sy, g["nsy"], TOL_STATE, "sy")
    assert torch.equal(isx.cpu(), torch.from_numpy(g["isx"]))      # inputs never modified
    for t in (1, 2, 3):
        y, sx, sy = ext().sos_forward(dev(g["xs"][:, :t]), None, sos, isx, isy)
        close(y, g[f"t{t}_y"], TOL_IIR_F64OUT, f"T={t} y")
        close(sx, g[f"t{t}_sx"], TOL_STATE, f"T={t} sx")
        close(sy, g[f"t{t}_sy"], TOL_STATE, f"T={t} sy")


def test_biquad_entry_point(golden):
    g = golden("iir_shapes")
    s = g["bq_sos"][0]
    x = dev(g["x1d"][None])
    y, sx, sy = ext().biquad_forward(x, torch.tensor(s[:3]), float(s[4]), float(s[5]), None, None)
    close(y[0], g["y1d"], TOL_IIR_F32OUT, "biquad y")
    assert sx.shape == (1, 2) and sy.shape == (1, 2)
    close(sx, g["bq_sx"][0], TOL_STATE)
    close(sy, g["bq_sy"][0], TOL_STATE)


@pytest.mark.parametrize("C,T,K", [(1, 1, 1), (3, 7, 2), (5, 63, 3), (2, 2049, 4), (7, 4097, 1),
                                   (1, 100003, 5), (64, 8192, 4), (3, 200000, 16)])
def test_random_shapes_vs_oracle(C, T, K, sos_variant):
    """Odd lengths (unaligned rows -> dword path), tiny inputs, many sections, random states."""
    rng = np.random.default_rng(C * 1000 + T + K)
    from scipy.signal import butter
    sos = np.vstack([butter(2, rng.uniform(0.02, 0.6), output="sos") for _ in range(K)])
    x = rnd((C, T), T)
    sx0, sy0 = rng.standard_normal((K, C, 2)), rng.standard_normal((K, C, 2))
    ey, esx, esy = O.sos_forward(x, sos, sx0, sy0)
    y, sx, sy = ext().sos_forward(dev(x), None, torch.from_numpy(sos), dev(sx0), dev(sy0))
    close(y, ey.astype(np.float32), 2.5e-7, "y")
    close(sx, esx, TOL_STATE, "sx")
    close(sy, esy, TOL_STATE, "sy")


def test_time_segmentation_is_exact(monkeypatch):
    """Segments with a warm-up halo (parallel over time) == one sequential segment per row."""
    from scipy.signal import butter
    sos = np.vstack([butter(6, 2000 / 24000, output="sos"), butter(2, 300 / 24000, "highpass", output="sos")])
    x = dev(rnd((4, 1_500_000), 3).astype(np.float64))
    monkeypatch.setenv("TFX_SOS_NSEG", "1")
    y1, sx1, sy1 = ext().sos_forward(x, None, torch.from_numpy(sos), None, None)
    for nseg in ("0", "7", "64", "300"):
        monkeypatch.setenv("TFX_SOS_NSEG", nseg)
        y2, sx2, sy2 = ext().sos_forward(x, None, torch.from_numpy(sos), None, None)
        close(y2, y1.cpu().numpy(), 1e-13, f"nseg={nseg}")
        close(sy2, sy1.cpu().numpy(), 1e-13, f"nseg={nseg} state")


@pytest.mark.parametrize("T", [257, 300, 2304, 4351, 4353, 10_000, 12_345, 65_536, 100_003])
def test_time_segmentation_geometry_edges(T, monkeypatch):
    """The halo is part of every stream's tile grid (stream g starts at g * (tiles * TILE - warm)): lengths around the
    halo / tile boundaries, rows that are not 16-byte aligned (dword path), carried-in states, every tile size."""
    from scipy.signal import butter
    rng = np.random.default_rng(T)
    sos = np.vstack([butter(4, 0.2, output="sos"), butter(2, 0.05, "highpass", output="sos")])
    K = sos.shape[0]
    x = dev(rnd((3, T), T).astype(np.float64))
    sx0, sy0 = dev(rng.standard_normal((K, 3, 2))), dev(rng.standard_normal((K, 3, 2)))
    for variant in ("2", "4", "1"):
        monkeypatch.setenv("TFX_SOS_VARIANT", variant)
        monkeypatch.setenv("TFX_SOS_NSEG", "1")
        y1, sx1, sy1 = ext().sos_forward(x, None, torch.from_numpy(sos), sx0, sy0)
        for nseg in ("2", "3", "5", "17", "64"):
            monkeypatch.setenv("TFX_SOS_NSEG", nseg)
            y2, sx2, sy2 = ext().sos_forward(x, None, torch.from_numpy(sos), sx0, sy0)
            close(y2, y1.cpu().numpy(), 1e-13, f"T={T} variant={variant} nseg={nseg}")
            close(sx2, sx1.cpu().numpy(), 1e-13, f"T={T} variant={variant} nseg={nseg} state_x")
            close(sy2, sy1.cpu().numpy(), 1e-13, f"T={T} variant={variant} nseg={nseg} state_y")


def test_empty_inputs():
    """No rows ([0, T]) or no samples ([C, 0]): empty outputs, the state passes through (iir_cpu.cpp writes back what it
    loaded); the overlap-save op keeps the reference's "kernel size" error for a signal shorter than the taps."""
    sos = torch.tensor([[0.2, 0.4, 0.2, 1.0, -0.5, 0.2], [0.3, 0.1, 0.2, 1.0, -0.3, 0.1]], dtype=torch.float64)
    for shape in ((0, 100), (2, 0), (0, 0)):
        x = torch.zeros(*shape, device=DEV)
        y, sx, sy = ext().sos_forward(x, None, sos, None, None)
        assert y.shape == shape and sx.shape == (2, shape[0], 2) and sy.shape == (2, shape[0], 2)
        assert ext().biquad_forward(x, sos[0, :3], -0.5, 0.2, None, None)[0].shape == shape
        assert ext().fir_direct_forward(x, torch.ones(5)).shape == shape
        assert ext().gain_forward(x, 0.5).shape == shape
    sx0 = torch.full((2, 2, 2), 3.0, dtype=torch.float64, device=DEV)
    sy0 = torch.full((2, 2, 2), 4.0, dtype=torch.float64, device=DEV)
    y, sx, sy = ext().sos_forward(torch.zeros(2, 0, device=DEV), None, sos, sx0, sy0)
    assert torch.equal(sx, sx0) and torch.equal(sy, sy0) and sx.data_ptr() != sx0.data_ptr()
    assert ext().fft_conv_forward(torch.zeros(0, 100, device=DEV), torch.ones(5), (4, 0)).shape == (0, 100)
    with pytest.raises(RuntimeError, match="kernel size"):
        ext().fft_conv_forward(torch.zeros(2, 0, device=DEV), torch.ones(5), (4, 0))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.int16])
def test_narrow_signal_dtypes_round_once_like_the_reference(dtype):
    """float16 / bfloat16 / integer signals: the reference computes in float64 and casts back (`_ops.py:95,149`,
    `iir.py:176`); here the float64 result is rounded once to the signal's dtype."""
    from torchfx_amd import filter as F
    f = F.LoButterworth(3000, order=4, fs=48000)
    if dtype.is_floating_point:
        x = torch.from_numpy(rnd((3, 5000), 17)).to(dtype)
    else:
        x = (torch.from_numpy(rnd((3, 5000), 17)) * 20000).to(dtype)
    y = f(x.to(DEV))
    assert y.dtype == dtype and y.shape == x.shape
    e, _, _ = O.sos_forward(x.to(torch.float64).numpy(), f._sos.cpu().numpy())
    want = torch.from_numpy(e).to(dtype)
    if dtype.is_floating_point:
        ulp = 2.0 ** (-10 if dtype == torch.float16 else -7)
        diff = (y.cpu().double() - want.double()).abs()
        assert (diff <= ulp * want.double().abs().clamp_min(1e-3)).all()          # at most the last bit (a tie the other way)
        assert (diff > 0).double().mean().item() < 0.01
    else:
        assert (y.cpu().int() - want.int()).abs().max().item() <= 1
    b = F.BiquadLPF(cutoff=1000, q=0.7, fs=48000)
    assert b(x.to(DEV)).dtype == dtype


def test_long_memory_filter_falls_back_to_sequential():
    """A pole pair at radius 0.999999 never decays within 2^26 samples -> nseg = 1, still exact."""
    r, th = 0.999999, 0.01
    sos = np.array([[1e-6, 0, 0, 1, -2 * r * np.cos(th), r * r]])
    from torchfx_amd import torchfx_ext as E
    info = E.sos_plan_info(sos)
    x = rnd((2, 300_000), 5)
    ey, _, esy = O.sos_forward(x, sos)
    y, _, sy = ext().sos_forward(dev(x), None, torch.from_numpy(sos), None, None, out_dtype=torch.float64)
    close(y, ey, 1e-9, f"y (warmup={info['warmup']})")
    close(sy, esy, 1e-9, "sy")


def test_f32_arithmetic_mode(golden):
    g = golden("iir_cfg2_sections")
    y, _, _ = ext().sos_forward(dev(g["x"]), None, torch.from_numpy(g["sos"]), None, None, precision="f32")
    close(y, g["y"], TOL_IIR_F32MATH, "f32 math")


def test_mixed_io_dtypes(golden):
    g = golden("iir_cfg2_sections")
    sos = torch.from_numpy(g["sos"])
    y, _, _ = ext().sos_forward(dev(g["x"]), None, sos, None, None, out_dtype=torch.float64)
    assert y.dtype == torch.float64
    close(y, g["y_sections"][-1], TOL_IIR_F64OUT, "f32 in / f64 out")
    y, _, _ = ext().sos_forward(dev(g["x"].astype(np.float64)), None, sos, None, None, out_dtype=torch.float32)
    close(y, g["y"], TOL_IIR_F32OUT, "f64 in / f32 out")


def test_errors_are_runtime_errors():
    x = torch.zeros(2, 16)
    with pytest.raises(RuntimeError, match="ROCm device"):
        ext().sos_forward(x, None, torch.eye(6)[:1].double(), None, None)
    xd = x.to(DEV)
    with pytest.raises(RuntimeError, match="state_x"):
        ext().sos_forward(xd, None, torch.tensor([[1., 0, 0, 1, 0, 0]]).double(), torch.zeros(3, 2, 2).to(DEV), None)
    with pytest.raises(RuntimeError, match="non-finite"):
        ext().sos_forward(xd, None, torch.tensor([[float("nan"), 0, 0, 1, 0, 0]]).double(), None, None)
    with pytest.raises(RuntimeError, match="kernel size"):
        ext().fft_conv_forward(xd, torch.ones(64))


def test_unaligned_base_pointer_and_wide_batches():
    """Contiguous views whose storage offset breaks 16-byte alignment take the dword path;
    many rows / few samples and few rows / many sections are also exercised."""
    from scipy.signal import butter, ellip
    sos = butter(4, 0.2, output="sos")
    big = dev(rnd((3 * 5000 + 8,), 1))
    for off in (1, 2, 3, 5):
        x = big[off:off + 3 * 5000].view(3, 5000)
        assert x.is_contiguous() and x.data_ptr() % 16 != 0
        ey, _, esy = O.sos_forward(x.cpu().numpy(), sos)
        y, _, sy = ext().sos_forward(x, None, torch.from_numpy(sos), None, None)
        close(y, ey.astype(np.float32), TOL_IIR_F32OUT, f"offset {off}")
        close(sy, esy, TOL_STATE)
        k = rnd((200,), off)
        close(ext().fir_direct_forward(x, k), O.fir_direct(x.cpu().numpy(), k), TOL_CONV_F32)
        close(ext().fft_conv_forward(x, k, (199, 0)), O.fir_direct(x.cpu().numpy(), k), TOL_CONV_F32)
    # many rows
    x = rnd((1000, 3000), 2)
    ey, _, _ = O.sos_forward(x, sos)
    y, _, _ = ext().sos_forward(dev(x), None, torch.from_numpy(sos), None, None)
    close(y, ey.astype(np.float32), TOL_IIR_F32OUT, "1000 rows")
    # many sections (order-40 elliptic: K = 20) and a 64-section cascade of mild biquads
    for sosk in (ellip(40, 0.5, 60, 0.3, output="sos"),
                 np.vstack([butter(2, f, output="sos") for f in np.linspace(0.05, 0.8, 64)])):
        x = rnd((2, 20000), sosk.shape[0])
        ey, esx, esy = O.sos_forward(x, sosk)
        y, sx, sy = ext().sos_forward(dev(x), None, torch.from_numpy(sosk), None, None, out_dtype=torch.float64)
        close(y, ey, 1e-9, f"K={sosk.shape[0]}")
        close(sy, esy, 1e-8 * max(1.0, np.abs(esy).max()))


@pytest.mark.parametrize("seed", range(6))
def test_random_stable_cascades_near_the_unit_circle(seed):
    """Random pole/zero placements with pole radii up to 0.9995 (memory of ~80k samples):
    segments + warm-up halo vs the sequential oracle, float64 in/out."""
    rng = np.random.default_rng(100 + seed)
    K = int(rng.integers(1, 7))
    rows = []
    for _ in range(K):
        r, th = rng.uniform(0.5, 0.9995), rng.uniform(0.01, 3.1)
        zr, zth = rng.uniform(0.0, 1.2), rng.uniform(0.0, 3.14)
        b = np.array([1.0, -2 * zr * np.cos(zth), zr * zr]) * rng.uniform(0.2, 1.0)
        rows.append([*b, 1.0, -2 * r * np.cos(th), r * r])
    sos = np.array(rows)
    C, T = int(rng.integers(1, 5)), int(rng.integers(200_000, 900_000))
    x = rnd((C, T), seed).astype(np.float64)
    ey, esx, esy = O.sos_forward(x, sos)
    y, sx, sy = ext().sos_forward(dev(x), None, torch.from_numpy(sos), None, None)
    scale = max(1.0, float(np.abs(ey).max()))
    close(y, ey, 1e-9 * scale / max(1.0, scale) , f"seed {seed}: K={K} T={T}")
    close(sy, esy, 1e-8 * max(1.0, float(np.abs(esy).max())))


def test_very_long_cascades_and_the_section_limit():
    """100 all-pass sections (|H| = 1, so nothing decays): above 96 sections the host skips the O(K^3)
    warm-up analysis and runs one sequential segment per row -- still exact; above 512 sections the call
    is refused with a clear message."""
    rng = np.random.default_rng(5)
    K = 100
    r, th = rng.uniform(0.3, 0.9, K), rng.uniform(0.2, 2.9, K)
    a1, a2 = -2 * r * np.cos(th), r * r
    sos = np.stack([a2, a1, np.ones(K), np.ones(K), a1, a2], axis=1)
    assert ext().sos_plan_info(sos)["warmup"] == -1
    x = rnd((2, 20000), 9, np.float64)
    y, _, sy = ext().sos_forward(dev(x), None, torch.from_numpy(sos), None, None)
    ey, _, esy = O.sos_forward(x, sos)
    close(y, ey, 1e-10, "100 all-pass sections")
    close(sy, esy, 1e-9, "states")
    with pytest.raises(RuntimeError, match="at most 512 sections"):
        ext().sos_forward(dev(x), None, torch.from_numpy(np.tile(sos, (6, 1))), None, None)


@pytest.mark.parametrize("case", ["cfg2", "butter4@2k", "hp4@300", "notchQ30", "peq100"])
def test_precision_auto_estimate_bounds_the_measured_float32_error(case):
    """precision="auto": the host-side estimate (replay of the float32 kernel arithmetic, x 2.5) must bound the
    error the float32 recursion really makes on the device, and "auto" must pick float32 only below 2e-5."""
    from scipy.signal import butter
    from torchfx_amd import filter as F

    def sos_of(*fs):
        for f in fs:
            f.fs = 48000
            f.compute_coefficients()
        return torch.cat([f._sos for f in fs])
    sos = {"cfg2": lambda: sos_of(F.LoButterworth(2000, order=6), F.ParametricEQ(frequency=1000, q=2.0, gain=3.0)),
           "butter4@2k": lambda: torch.from_numpy(butter(4, 2000 / 24000, output="sos")),
           "hp4@300": lambda: sos_of(F.HiButterworth(300, order=4)),
           "notchQ30": lambda: sos_of(F.Notch(1000, 30.0)),
           "peq100": lambda: sos_of(F.ParametricEQ(frequency=100, q=4.0, gain=12.0))}[case]()
    info = ext().sos_plan_info(sos)
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.rand(8, 1_500_000, device=DEV, generator=g) * 2 - 1
    y64 = ext().sos_forward(x, None, sos, None, None, out_dtype=torch.float64, precision="f64")[0]
    y32 = ext().sos_forward(x, None, sos, None, None, precision="f32")[0]
    ya = ext().sos_forward(x, None, sos, None, None, precision="auto")[0]
    scale = max(1.0, float(y64.abs().max()))
    err32 = float((y32.double() - y64).abs().max()) / scale
    assert err32 <= info["f32_error_bound"], (case, err32, info)
    erra = float((ya.double() - y64).abs().max()) / scale
    if info["auto_precision"] == "f32":
        assert info["f32_error_bound"] <= 2e-5 and torch.equal(ya, y32)
    else:
        assert erra <= 1.5e-7                      # auto stayed in float64: one ulp of the float32 output


@pytest.mark.parametrize("bad", [float("nan"), float("inf")])
def test_non_finite_input_poisons_the_rest_of_the_row_like_the_sequential_recursion(bad, monkeypatch):
    """iir_cpu.cpp:132-147: once a NaN / Inf is in the state it never leaves.  The time-segmented launch must give
    the same picture -- finite before the bad sample, non-finite from it to the end of the row and in the
    returned state, other rows untouched -- for any number of segments."""
    from scipy.signal import butter
    sos = torch.from_numpy(np.vstack([butter(6, 2000 / 24000, output="sos"), [[1.0089, -1.9636, 0.9695, 1, -1.9636, 0.9784]]]))
    x = rnd((3, 400_000), 31)
    pos = 123_457
    x[1, pos] = bad
    ref, _, refs = O.sos_forward(x, sos.numpy())
    for nseg in ("1", "8", "0"):                                   # 0 = the launch's own choice
        monkeypatch.setenv("TFX_SOS_NSEG", nseg)
        y, sx, sy = ext().sos_forward(dev(x), None, sos, None, None)
        y = y.cpu().numpy()
        assert np.isfinite(y[0]).all() and np.isfinite(y[2]).all() and np.isfinite(y[1, :pos]).all(), nseg
        assert not np.isfinite(y[1, pos:]).any(), nseg
        assert not np.isfinite(sy[:, 1].cpu().numpy()).any() and np.isfinite(sy[:, 0].cpu().numpy()).all(), nseg
        for c in (0, 2):
            assert np.abs(y[c] - ref[c].astype(np.float32)).max() <= 1.5e-7 * max(1.0, np.abs(ref[c]).max())
    assert not np.isfinite(ref[1, pos:]).any() and not np.isfinite(refs[:, 1]).any()     # the oracle agrees


@pytest.mark.parametrize("bad", [float("nan"), float("inf")])
def test_non_finite_input_with_epilogue_bank_and_taps_is_segment_independent(bad, monkeypatch):
    """the poisoning of later segments must not depend on what an epilogue did to the stored
    samples (clamp turns an Inf end sample into 1), must reach the statistic a following Normalize reads, and must
    also hold for filter-bank, sum-mode and section-tap launches: every result equals the one-segment launch."""
    from scipy.signal import butter
    E = ext()
    sos = torch.from_numpy(np.vstack([butter(4, 1500 / 24000, output="sos"), [[1.0089, -1.9636, 0.9695, 1, -1.9636, 0.9784]]]))
    banks = torch.stack([sos, torch.from_numpy(np.vstack([butter(6, 3000 / 24000, output="sos")]))])
    x = rnd((3, 300_000), 32)
    x[1, 77_777] = bad
    xd = dev(x)

    def same(a, b, what):
        assert np.array_equal(a.cpu().numpy(), b.cpu().numpy(), equal_nan=True), what

    def run():
        out = {}
        for stat in ("absmax", "sumsq"):
            ep = E.Epilogue(gain=0.5, clamp=True, stat=stat, per_row=True)
            y, _, sy = E.sos_forward(xd, None, sos, None, None, epilogue=ep)
            out["ep_" + stat] = (y, sy, ep.stat_value.clone())
        out["bank"] = E.sos_bank_forward(xd, banks, None, None)
        out["sum"] = E.sos_bank_sum_forward(xd, banks, None, None)
        out["taps"] = E.sos_forward(xd.double(), None, sos, None, None, return_sections=True)
        return out
    monkeypatch.setenv("TFX_SOS_NSEG", "1")
    ref = run()
    # the one-segment fused epilogue equals the staged passes (sequential recursion, then Gain, then the reduction)
    ys = E.gain_forward(E.sos_forward(xd, None, sos, None, None)[0], 0.5, True)
    same(ref["ep_absmax"][0], ys, "fused epilogue vs staged")
    assert not np.isfinite(ref["ep_absmax"][2].cpu().numpy()[1]) and np.isfinite(ref["ep_absmax"][2].cpu().numpy()[[0, 2]]).all()
    for nseg in ("7", "0"):
        monkeypatch.setenv("TFX_SOS_NSEG", nseg)
        got = run()
        for key in ref:
            for i, (a, b) in enumerate(zip(ref[key], got[key])):
                a_, b_ = a.cpu().numpy(), b.cpu().numpy()
                assert np.array_equal(np.isfinite(a_), np.isfinite(b_)), (key, i, nseg)
                fin = np.isfinite(a_)
                scale = max(1.0, float(np.abs(a_[fin]).max())) if fin.any() else 1.0
                assert np.abs(a_[fin] - b_[fin]).max() <= (1e-6 if key.startswith("ep_") and i == 2 else 3e-7) * scale, (key, i, nseg)
