"""GPU parity -- effects between filters and epilogues (8f rank 3).
Tolerances and helpers: tests/gpu_common.py."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.gpu_common import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


GAIN_CASES = {"amp": dict(gain=0.37, gain_type="amplitude"), "db": dict(gain=-4.5, gain_type="db"),
              "db0": dict(gain=0.0, gain_type="db"), "pow": dict(gain=2.5, gain_type="power"),
              "clamp": dict(gain=1.9, gain_type="amplitude", clamp=True)}


def _strategies():
    from torchfx_amd import effect as E
    return {"peak": E.PeakNormalizationStrategy(), "rms": E.RMSNormalizationStrategy(),
            "percentile": E.PercentileNormalizationStrategy(97.0), "per_channel": E.PerChannelNormalizationStrategy()}


def test_gain_and_normalize_golden(golden):
    """Gain is bit-exact (one rounding, same as torch); the normalisations are within 1e-6 of the
    reference (the RMS is accumulated in float64 here, in float32 there)."""
    from torchfx_amd import effect as E
    g = golden("effects")
    x, x64, z = dev(g["x"]), dev(g["x64"]), dev(g["zeros"])
    for tag, kw in GAIN_CASES.items():
        y = E.Gain(**kw)(x)
        assert np.array_equal(y.cpu().numpy(), g["gain_" + tag]), tag
    assert np.array_equal(E.Gain(3.0, "db")(x64).cpu().numpy(), g["gain64_db"])
    for name, st in _strategies().items():
        close(E.Normalize(0.8, st)(x), g["norm_" + name], 1e-6, name)
        close(E.Normalize(1.25, st)(x64), g["norm64_" + name], 1e-14, name + " f64")
        assert np.array_equal(E.Normalize(0.8, st)(z).cpu().numpy(), g["normz_" + name]), name
    close(E.Normalize(0.5, E.PerChannelNormalizationStrategy())(dev(g["x3"])), g["norm3_per_channel"], 1e-6)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("C,T", [(1, 1), (3, 17), (2, 4097), (5, 100003), (2, 1 << 20)])
def test_effect_kernels_ragged_shapes_vs_oracle(C, T, dtype):
    e = ext()
    x = (rnd((C, T), 31 * C + T, dtype) * 3).astype(dtype)
    if C > 1:
        x[1] = 0
    xd = dev(x)
    assert np.array_equal(e.gain_forward(xd, 0.731, True).cpu().numpy(), O.gain(x, 0.731, "amplitude", True))
    st = e.stat_forward(xd, e.STAT_ABSMAX, per_row=True).cpu().numpy()
    assert np.array_equal(st, np.abs(x).max(axis=1).astype(np.float64))
    assert e.stat_forward(xd, e.STAT_ABSMAX).item() == np.abs(x).max()
    rms = e.stat_forward(xd, e.STAT_RMS, per_row=True).cpu().numpy()
    assert np.allclose(rms, np.sqrt((x.astype(np.float64) ** 2).mean(axis=1)), rtol=1e-13)
    tol = 2e-7 if dtype == np.float32 else 1e-15
    for strat, mode, per_row in (("peak", e.STAT_ABSMAX, False), ("per_channel", e.STAT_ABSMAX, True), ("rms", e.STAT_RMS, False)):
        y = e.normalize_forward(xd, 0.9, mode, per_row).cpu().numpy()
        exp = O.normalize(x, 0.9, strat)
        assert np.abs(y - exp).max() <= tol * max(1.0, np.abs(exp).max()), strat
    # rows that are not 16-byte aligned (a view shifted by one sample) take the scalar path
    if T > 8:
        xs = dev(x)[:, 1:]
        assert not xs.is_contiguous()
        y = e.normalize_forward(xs, 0.9, e.STAT_ABSMAX, True).cpu().numpy()
        assert np.abs(y - O.normalize(x[:, 1:], 0.9, "per_channel")).max() <= tol * 3


def test_effect_nan_and_aliasing_rules():
    e = ext()
    x = rnd((2, 5000), 5)
    x[0, 1234] = np.nan
    xd = dev(x)
    st = e.stat_forward(xd, e.STAT_ABSMAX, per_row=True).cpu().numpy()
    assert np.isnan(st[0]) and st[1] == np.abs(x[1]).max()             # NaN wins, like torch.max
    y = e.normalize_forward(xd, 1.0, e.STAT_ABSMAX, False).cpu().numpy()
    assert np.array_equal(np.isnan(y), np.isnan(x)) and np.array_equal(y[1], x[1])   # `nan > 0` is False: unchanged
    g = e.gain_forward(xd, 2.0, True).cpu().numpy()
    assert np.isnan(g[0, 1234]) and np.abs(g[1]).max() <= 1.0
    assert xd.cpu().numpy()[1].tobytes() == x[1].tobytes()              # inputs never written


def test_reverb_on_device(golden):
    from torchfx_amd import effect as E
    g = golden("delay")
    rv = E.Reverb(delay=100, decay=0.5, mix=0.3)
    close(rv(dev(g["x"])), g["y"], 1e-7, "reverb")
    close(rv(dev(g["x"]).reshape(1, 2, -1))[0], g["y"], 1e-7, "reverb 3-D")
    close(rv(dev(g["x"][0])), g["y"][0], 1e-7, "reverb 1-D")
    s = dev(np.zeros((2, 50), np.float32))
    assert rv(s) is s


def _tails():
    import torchfx_amd as fx
    E = fx.effect
    return {
        "gain+clamp": lambda: [fx.Gain(1.7, clamp=True)],
        "gain db": lambda: [fx.Gain(-3.0, "db")],
        "peak": lambda: [fx.Normalize(0.8)],
        "gain+peak": lambda: [fx.Gain(0.3), fx.Normalize(0.8)],
        "rms": lambda: [fx.Normalize(0.5, E.RMSNormalizationStrategy())],
        "clamp+per_channel": lambda: [fx.Gain(2.0, clamp=True), fx.Normalize(0.7, E.PerChannelNormalizationStrategy())],
    }


@pytest.mark.parametrize("producer", ["cascade f32", "cascade f64 io", "lone iir", "fir fft short rows", "fir fft long rows", "cascade unaligned"])
@pytest.mark.parametrize("tail", ["gain+clamp", "gain db", "peak", "gain+peak", "rms", "clamp+per_channel"])
def test_epilogue_equals_staged_passes(producer, tail):
    """filter | Gain | Normalize as ONE kernel with an epilogue (+ one apply pass for Normalize) against the same
    modules staged as separate HIP passes: bit-identical for gain / clamp, 1e-6 for the normalisations (the
    statistic is reduced in another order).  Covers the fused epilogues (float32 cascade kernel, last pass of the
    LDS-resident overlap-save) and the producers that run the epilogue as passes inside the call (float64 I/O,
    unaligned rows, rocFFT path)."""
    import torchfx_amd as fx
    from torchfx_amd import filter as F
    T = {"fir fft long rows": 200_000, "cascade unaligned": 30_001}.get(producer, 30_000)
    dt = np.float64 if producer == "cascade f64 io" else np.float32
    x = dev(rnd((3, T), 21, dt) * 0.9)

    def filt():
        if producer.startswith("cascade"):
            return [F.LoButterworth(3000, order=4), F.HiShelving(2000, q=0.7, gain=2.0)]
        if producer == "lone iir":
            return [F.ParametricEQ(frequency=800, q=3.0, gain=6.0)]
        return [F.FIR(np.hanning(129) / np.hanning(129).sum() * 1.5)]
    outs = []
    for ep in (False, True):
        w = fx.Wave(x, 48000, device=DEV)
        w.fuse_epilogue, w.fuse_fir, w.fuse_spectral = ep, False, False
        for m in filt() + _tails()[tail]():
            w = w | m
        if ep:
            assert [type(m).__name__ for m in w.plan()] == ["Epilogued"]
        outs.append(w.ys)
    staged, fused = outs
    assert fused.dtype == staged.dtype and fused.shape == staged.shape
    if "peak" in tail or "rms" in tail or "per_channel" in tail:
        close(fused, staged.cpu().numpy(), 1e-6 if dt == np.float32 else 1e-13, f"{producer} | {tail}")
    else:
        assert torch.equal(fused, staged), f"{producer} | {tail}"


def test_epilogue_statistics_and_golden_mix(golden):
    """The raw statistic an epilogue leaves on the device == the statistic of the stored output; and the reference's
    mixed pipeline (tests/golden/effects.npz, iir | iir | Gain | iir | iir) under the default plan."""
    import torchfx_amd as fx
    from torchfx_amd import filter as F
    e = ext()
    x = dev(rnd((4, 150_000), 22) * 0.9)
    from scipy.signal import butter
    sos = torch.from_numpy(butter(4, 3000 / 24000, output="sos"))
    for stat, per_row in (("absmax", False), ("absmax", True), ("sumsq", False), ("sumsq", True)):
        ep = e.Epilogue(gain=1.3, clamp=True, stat=stat, per_row=per_row)
        y, _, _ = e.sos_forward(x, None, sos, None, None, epilogue=ep)
        yd = y.double()
        rows = yd if per_row else yd.reshape(1, -1)
        ref = rows.abs().max(dim=1).values if stat == "absmax" else (rows * rows).sum(dim=1)
        assert torch.allclose(ep.stat_value, ref, rtol=1e-12, atol=0), (stat, per_row)
        assert float(y.abs().max()) <= 1.0
        for taps, block in ((301, 4096), (3001, 8192), (6001, 16384)):         # the three one-launch kernels: fused into their stores
            assert e.ols_plan_info(taps, x.shape[1], (taps - 1, 0))["N"] == block
            k = torch.from_numpy((np.hanning(taps) / np.hanning(taps).sum()).astype(np.float32))
            ep2 = e.Epilogue(gain=0.5, stat=stat, per_row=per_row)
            y2 = e.fft_conv_forward(x, k, (taps - 1, 0), epilogue=ep2)
            assert torch.equal(y2, e.gain_forward(e.fft_conv_forward(x, k, (taps - 1, 0)), 0.5))
            rows = y2.double() if per_row else y2.double().reshape(1, -1)
            ref = rows.abs().max(dim=1).values if stat == "absmax" else (rows * rows).sum(dim=1)
            assert torch.allclose(ep2.stat_value, ref, rtol=1e-12, atol=0), ("fft", taps, stat, per_row)
    g = golden("effects")
    w = fx.Wave(g["mix_x"], 48000, device=DEV)
    for m in (F.LoButterworth(4000, order=2), F.HiButterworth(200, order=2), fx.Gain(0.5), F.LoButterworth(6000, order=2),
              F.HiButterworth(100, order=2)):
        w = w | m
    assert [type(m).__name__ for m in w.plan()] == ["Epilogued", "FusedSOSCascade"]
    close(w.ys, g["mix_y"], TOL_IIR_F32OUT * 2, "reference mixed pipeline, gain as an epilogue")


@pytest.mark.parametrize("n,q", [(1, 0.5), (2, 0.3), (7, 0.99), (1000, 0.97), (4097, 0.5), (100_003, 0.999), (1 << 20, 0.97),
                                 (3_000_001, 0.25), (16_000_000, 0.99), (5, 1.0), (5, 0.0)])
def test_quantile_abs_equals_torch_quantile(n, q):
    """The percentile threshold as a three-pass radix select on the device == torch.quantile(|x|, q, "linear") on the same
    float32 data (torch's own CPU implementation; same float32 rank arithmetic and lerp), including heavy ties, denormals,
    zeros and +-inf."""
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, generator=g)
    if n > 100:
        x[::7] = x[3]                      # ties
        x[5] = 0.0
        x[11] = -0.0
        x[13] = 1e-42                      # denormal
        x[17] = float("inf")
        x[19] = -float("inf")
    ref = torch.quantile(x.abs(), q, interpolation="linear")
    got = ext().quantile_abs(x.to(DEV), q).cpu()
    assert got.dtype == torch.float64 and got.shape == (1,)
    assert float(got) == float(ref), (n, q, float(got), float(ref))
    # unaligned base pointer and a 2-D view
    if n > 9:
        xo = torch.zeros(n + 1, device=DEV)
        xo[1:] = x.to(DEV)
        assert float(ext().quantile_abs(xo[1:], q).cpu()) == float(ref)


def test_quantile_abs_nan_and_beyond_torch_limit():
    x = torch.randn(1000)
    x[123] = float("nan")
    assert torch.isnan(ext().quantile_abs(x.to(DEV), 0.5).cpu()).all()
    # 40 M elements: torch.quantile refuses (> 16 M); the select agrees with a sort-based float32 computation done ATen's way
    n = 40_000_000
    g = torch.Generator(device=DEV).manual_seed(3)
    xd = torch.randn(n, device=DEV, generator=g)
    q = 0.97
    s = torch.sort(xd.abs()).values
    ranks = np.float32(q) * np.float32(n - 1)
    lo, hi = int(np.floor(ranks)), int(np.ceil(ranks))
    w = np.float32(ranks - np.float32(np.floor(ranks)))
    a, b = np.float32(s[lo].item()), np.float32(s[hi].item())
    exp = a + w * (b - a) if w < 0.5 else b - (b - a) * (np.float32(1) - w)
    assert float(ext().quantile_abs(xd, q).cpu()) == float(np.float32(exp))


def test_percentile_strategy_runs_on_the_device_without_torch_quantile(monkeypatch):
    """PercentileNormalizationStrategy (effect.py:723-755) on a float32 device signal: select + apply pass, equal to the
    reference formula; an all-zero signal (threshold 0) is returned unchanged; a 3-D batch works."""
    from torchfx_amd import effect as E
    x = torch.from_numpy(rnd((3, 50_001), 91))
    exp = x / torch.quantile(x.abs(), 0.97, interpolation="linear") * 0.8
    monkeypatch.setattr(torch, "quantile", None)                   # the device path must not need it
    y = E.Normalize(0.8, E.PercentileNormalizationStrategy(97.0))(x.to(DEV))
    close(y, exp.numpy(), 1e-6, "percentile normalisation")
    z = torch.zeros(2, 1000, device=DEV)
    assert torch.equal(E.PercentileNormalizationStrategy(99.0)(z, 1.0), z)
    xb = x.reshape(3, 1, -1).to(DEV)
    close(E.PercentileNormalizationStrategy(97.0)(xb, 0.8), exp.reshape(3, 1, -1).numpy(), 1e-6)


def test_percentile_normalisation_is_hip_graph_capturable():
    """The selection is stream-ordered device work only (an init launch, three histogram + pick pairs, the apply pass): it can be
    captured into a HIP graph and replayed on new samples; no host-to-device copy, no host synchronisation."""
    from torchfx_amd import effect as E
    strat = E.PercentileNormalizationStrategy(97.0)
    static_x = dev(rnd((4, 200_000), 5))
    for _ in range(2):
        strat(static_x, 0.8)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        strat(static_x, 0.8)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        out = strat(static_x, 0.8)
    for seed in (6, 7):
        x2 = torch.from_numpy(rnd((4, 200_000), seed))
        static_x.copy_(x2.to(DEV))
        graph.replay()
        torch.cuda.synchronize()
        exp = x2 / torch.quantile(x2.abs(), 0.97, interpolation="linear") * 0.8
        close(out, exp.numpy(), 1e-6, f"replay {seed}")
