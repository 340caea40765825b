"""Host-side logic on CPU (kernels replaced by the oracle through the `oracle_backend`
fixture): coefficient design, the Wave planner / chain fusion, shape / dtype / state rules,
`+`, error behaviour.  Mirrors the reference's own hot-path tests."""
import ast

import numpy as np
import pytest
import torch
from torch import nn

import torchfx_amd as fx
from torchfx_amd import filter as F


def close(a, b, tol):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    assert np.abs(a.astype(np.float64) - np.asarray(b, dtype=np.float64)).max() <= tol * max(1.0, np.abs(b).max())


# ------------------------------------------------------------------ designs (row a8)
def test_every_design_class_matches_reference_sos(golden):
    g = golden("designs")
    specs = [ast.literal_eval(str(s)) for s in g["specs"]]
    keys = [k for k in g.files if k != "specs"]
    assert len(specs) == len(keys) == 68
    per_fs = len(specs) // 2
    for n, (cls, args, kw, fs) in enumerate(specs):
        key = f"{n % per_fs:02d}_{cls}_{fs}"
        f = getattr(F, cls)(*args, fs=fs, **kw)
        f.compute_coefficients()
        exp = g[key]
        assert f._sos.dtype == torch.float64 and tuple(f._sos.shape) == exp.shape, key
        # same SciPy, same cookbook arithmetic order -> identical rows
        assert np.array_equal(f._sos.numpy(), exp), key


def test_designable_fir_taps(golden):
    g = golden("fir_designs")
    for i in range(4):
        cut, nt, kw = ast.literal_eval(str(g[f"spec{i}"]))
        f = F.DesignableFIR(cutoff=cut, num_taps=nt, fs=48000, **kw)
        assert f.kernel.dtype == torch.float32 and tuple(f.kernel.shape) == (1, 1, nt)
        assert np.array_equal(f.kernel.numpy(), g[f"k{i}"])


def test_fir_kernel_is_flipped_float32():
    f = F.FIR([1.0, 2.0, 3.0])
    assert f.kernel.dtype == torch.float32
    assert f.kernel.reshape(-1).tolist() == [3.0, 2.0, 1.0] and f.a == [1.0]
    assert list(f.state_dict()) == ["kernel"]          # IIR modules have an empty state_dict
    assert list(F.LoButterworth(100, fs=8000).state_dict()) == []
    with pytest.raises(ValueError, match="conv_mode"):
        F.FIR([1.0], conv_mode="nope")


def test_biquad_coefficient_views_and_defaults():
    f = F.BiquadLPF(1000, 0.707, fs=48000)
    assert f.b is None and f.a is None
    f.compute_coefficients()
    assert f.b.shape == (3,) and f.a[0] == 1.0 and f._has_computed_coeff
    assert F.LoButterworth(100).order == 5 and F.HiButterworth(100).order == 5
    assert F.Butterworth("lowpass", 100, order=24, order_scale="db").order == 4
    with pytest.raises(ValueError, match="positive even"):
        F.LinkwitzRiley("lowpass", 1000, order=3)
    assert F.Peaking(1000, 1.0, -1.0, "linear").gain_db == 0


# ------------------------------------------------------------------ forward rules (a6, a7)
def test_missing_fs_raises(oracle_backend):
    with pytest.raises(ValueError, match="Sample rate"):
        F.LoButterworth(1000)(torch.zeros(1, 8))
    with pytest.raises(ValueError, match="Sample rate"):
        F.BiquadHPF(1000, 0.7)(torch.zeros(1, 8))


def test_shapes_dtypes_state(oracle_backend, golden):
    g = golden("iir_shapes")
    bq = F.BiquadLPF(cutoff=1500, q=0.9, fs=48000)
    y = bq(torch.from_numpy(g["x1d"]))
    assert y.shape == g["y1d"].shape and y.dtype == torch.float32
    close(y, g["y1d"], 1e-7)
    close(bq._state_x, g["bq_sx"], 1e-12)
    lr = F.LoLinkwitzRiley(1200, order=4, fs=44100)
    y = lr(torch.from_numpy(g["x3d"]))
    assert y.dtype == torch.float64 and tuple(y.shape) == g["y3d"].shape
    close(y, g["y3d"], 1e-12)
    assert tuple(lr._state_x.shape) == (2, 6, 2)            # [K, B*C, 2]
    lr(torch.from_numpy(g["x3d"][0]))                       # C changes -> state re-zeroed
    assert tuple(lr._state_x.shape) == (2, 3, 2)
    lr.reset_state()
    assert lr._sos is None and lr._state_x is None          # IIR.reset_state drops the design too
    bq.reset_state()
    assert bq._sos is not None and bq._state_x is None      # Biquad.reset_state keeps it


def test_state_carries_across_calls(oracle_backend, golden):
    g = golden("iir_chunked")
    fa, fb = F.HiButterworth(300, order=3, fs=44100), F.LoChebyshev1(4000, order=4, ripple=0.5, fs=44100)
    fz = F.FusedSOSCascade(fa, fb)
    assert np.array_equal(fz._sos.numpy(), g["sos"])
    x = torch.from_numpy(g["x"])
    close(fz(x[:, :1024]), g["y1"], 1e-12)
    close(fz(x[:, 1024:]), g["y2"], 1e-12)
    close(fz._state_y, g["state_y"], 1e-12)


# ------------------------------------------------------------------ planner / fusion (a9-a11)
def test_wave_is_lazy_and_fuses_iir_runs(oracle_backend, golden):
    g = golden("chain")
    oracle_backend.calls.clear()
    f1, f2 = F.HiButterworth(100, order=2), F.LoButterworth(8000, order=4)
    f3 = F.ParametricEQ(2000, 1.0, -3.0)
    fir = F.DesignableFIR(cutoff=6000, num_taps=127)
    w = fx.Wave(g["x"], 48000)
    w.fuse_fir = w.fuse_spectral = False                                   # the reference's staging (wave.py:207-239)
    w = w | f1 | f2 | fir | f3
    assert f1.fs == 48000 and fir.fs == 48000 and f1._sos is not None      # fs propagated, eager design
    assert oracle_backend.calls == []                                      # nothing computed yet
    plan = w.plan()
    assert [type(p).__name__ for p in plan] == ["FusedSOSCascade", "DesignableFIR", "ParametricEQ"]
    assert np.array_equal(plan[0]._sos.numpy(), g["sos_run1"])
    close(w.ys, g["y_chain"], 2e-5)
    names = [c[0] for c in oracle_backend.calls]
    assert names == ["sos_forward", "fft_conv_forward", "sos_forward"]
    assert oracle_backend.calls[0][2] == 3                                 # 1 + 2 sections in one call
    assert f1._state_x is None                                             # members of a fused run keep no state
    assert f3._state_x is not None                                         # a lone IIR step does (quirk, wave.py:221-224)


def test_chain_forms_are_equivalent(oracle_backend):
    x = torch.randn(2, 3000, generator=torch.Generator().manual_seed(0), dtype=torch.float64)

    def fs3():
        return F.LoButterworth(3000, order=2), F.HiChebyshev1(200, order=3), F.Notch(1000, 5.0)
    a = (fx.Wave(x, 44100) | fs3()[0] | fs3()[1] | fs3()[2]).ys
    f = fs3()
    b = (fx.Wave(x, 44100) | (f[0] | f[1] | f[2])).ys
    c = (fx.Wave(x, 44100) | nn.Sequential(*fs3())).ys
    seq = x
    for m in fs3():
        m.fs = 44100
        seq = m(seq)
    for other in (b, c, seq):
        close(other, a.numpy(), 1e-12)
    chain = f[0] | f[1] | f[2]
    assert isinstance(chain, fx.FilterChain) and len(chain) == 3
    assert len((f[0] | f[1]) | (f[1] | f[2])) == 4                          # flattening
    assert f[0].__or__(3) is NotImplemented
    with pytest.raises(TypeError, match="nn.Module"):
        fx.Wave(x, 44100) | 3


def test_fused_cascade_validation():
    with pytest.raises(ValueError, match="at least one"):
        F.FusedSOSCascade()
    with pytest.raises(ValueError, match="different sample rates"):
        F.FusedSOSCascade(F.LoButterworth(100, fs=8000), F.LoButterworth(100, fs=16000))
    with pytest.raises(ValueError, match="no sampling frequency"):
        F.FusedSOSCascade(F.LoButterworth(100))
    with pytest.raises(TypeError):
        F.FusedSOSCascade(F.FIR([1.0]))
    fz = F.FusedSOSCascade.from_chain(nn.Sequential(F.LoButterworth(100, order=2, fs=8000), F.FIR([1.0]),
                                                    F.BiquadNotch(50, 5, fs=8000)))
    assert fz._sos.shape == (2, 6) and fz.fs == 8000


def test_fusion_policy_defaults(monkeypatch):
    """auto (default): FIR-run merge and the recursion-inside-the-overlap-save fusion on (both keep the reference's
    arithmetic), spectral folding (float32 FFT arithmetic for the IIR part) and gain folding opt-in; reference: all off;
    the per-feature variables override either way."""
    for k in ("TORCHFX_AMD_FUSION", "TORCHFX_AMD_FUSE_FIR", "TORCHFX_AMD_FUSE_SPECTRAL", "TORCHFX_AMD_FUSE_GAIN",
              "TORCHFX_AMD_FUSE_RECURSIVE"):
        monkeypatch.delenv(k, raising=False)
    w = fx.Wave(torch.zeros(1, 8), 48000)
    assert (w.fuse_fir, w.fuse_spectral, w.fuse_gain, w.fuse_recursive) == (True, False, False, True)
    monkeypatch.setenv("TORCHFX_AMD_FUSE_SPECTRAL", "1")
    assert fx.Wave(torch.zeros(1, 8), 48000).fuse_spectral is True
    monkeypatch.delenv("TORCHFX_AMD_FUSE_SPECTRAL")
    monkeypatch.setenv("TORCHFX_AMD_FUSE_RECURSIVE", "0")
    assert fx.Wave(torch.zeros(1, 8), 48000).fuse_recursive is False
    monkeypatch.delenv("TORCHFX_AMD_FUSE_RECURSIVE")
    monkeypatch.setenv("TORCHFX_AMD_FUSION", "reference")
    w = fx.Wave(torch.zeros(1, 8), 48000)
    assert (w.fuse_fir, w.fuse_spectral, w.fuse_gain, w.fuse_recursive) == (False, False, False, False)
    chain = w | F.LoButterworth(2000, order=2) | F.HiButterworth(100, order=2) | F.FIR([0.5, 0.5]) | F.FIR([1.0, -1.0])
    assert [type(m).__name__ for m in chain.plan()] == ["FusedSOSCascade", "FIR", "FIR"]     # wave.py:207-239 staging
    monkeypatch.setenv("TORCHFX_AMD_FUSE_FIR", "1")
    assert fx.Wave(torch.zeros(1, 8), 48000).fuse_fir is True
    monkeypatch.setenv("TORCHFX_AMD_FUSION", "auto")
    monkeypatch.setenv("TORCHFX_AMD_FUSE_SPECTRAL", "0")
    w = fx.Wave(torch.zeros(1, 8), 48000)
    assert (w.fuse_fir, w.fuse_spectral) == (True, False)


def test_fir_run_merge(oracle_backend, golden):
    g = golden("chain")
    from scipy.signal import firwin
    irs = np.random.default_rng(1).standard_normal(4097) * np.exp(-np.arange(4097) / 500.0)
    irs = irs / np.abs(irs).sum()

    def pipe(fuse):
        w = fx.Wave(g["xc"], 48000)
        w.fuse_fir, w.fuse_spectral = fuse, False
        return (w | F.LoButterworth(2000, order=6) | F.ParametricEQ(frequency=1000, q=2.0, gain=3.0)
                | F.FIR(firwin(1024, 5000, fs=48000)) | F.FIR(irs))
    assert len(pipe(False).plan()) == 3
    close(pipe(False).ys, g["yc"], 2e-5)
    wf = pipe(True)
    plan = wf.plan()
    assert len(plan) == 2 and plan[1].kernel.numel() == 1024 + 4097 - 1
    close(wf.ys, g["yc"], 2e-5)                   # conv associativity: same result as staged


# ------------------------------------------------------------------ parallel sum (a12)
def test_parallel_combination(oracle_backend, golden):
    g = golden("chain")
    p1, p2 = F.LoButterworth(1000, order=2), F.HiButterworth(4000, order=2)
    comb = p1 + p2
    assert isinstance(comb, F.ParallelFilterCombination) and not comb._has_computed_coeff
    w = fx.Wave(g["x"], 48000) | comb
    assert p1.fs == 48000 and comb._has_computed_coeff
    close(w.ys, g["y_par"], 1e-6)
    nested = (F.LoButterworth(1000, order=2, fs=48000) + F.HiButterworth(4000, order=2, fs=48000)) | F.Notch(50, 3, fs=48000)
    assert isinstance(nested, fx.FilterChain)


# ------------------------------------------------------------------ FIR / fft_conv1d (a13, a14)
def test_fir_forward_modes_and_shapes(oracle_backend, golden):
    g = golden("fir")
    x = torch.from_numpy(g["x"])
    from scipy.signal import firwin
    b = firwin(32, 5000, fs=48000)
    for mode, key in (("fft", "fft32"), ("auto", "fft32"), ("direct", "direct32")):
        y = F.FIR(b, conv_mode=mode)(x)
        assert y.shape == x.shape and y.dtype == x.dtype
        close(y, g[key], 2e-5)
    assert F.FIR(b)(x[0]).shape == x[0].shape
    assert F.FIR(b)(x[None]).shape == (1, *x.shape)
    with pytest.raises(ValueError, match="shape"):
        F.FIR(b)(torch.zeros(1, 1, 1, 8))
    close(F.FIR(g["kt"][::-1].copy())(torch.from_numpy(g["xt"])), g["yt_fft"], 1e-12)    # f64, T < K


def test_fft_conv1d_signature_and_errors(oracle_backend, golden):
    from torchfx_amd.filter._fftconv import fft_conv1d
    g = golden("fftconv")
    x = torch.from_numpy(g["x"])[None]
    y = fft_conv1d(x, torch.from_numpy(g["k16"])[None, None], padding=(8, 7))
    assert tuple(y.shape) == (1, 2, 50000)
    close(y[0], g["y16_pad87"], 2e-5)
    with pytest.raises(RuntimeError, match="kernel size"):
        fft_conv1d(x[..., :10], torch.ones(1, 1, 16))
    with pytest.raises(RuntimeError, match="Block ratio"):
        fft_conv1d(x, torch.ones(1, 1, 16), block_ratio=0.5)


def test_ops_module_surface():
    from torchfx_amd import _ops, torchfx_ext
    assert _ops.PARALLEL_SCAN_THRESHOLD == 2048
    for n in ("biquad_forward", "sos_forward", "delay_line_forward"):
        assert hasattr(torchfx_ext, n)
    assert fx.is_native_available() is True


# ------------------------------------------------------------------ filter bank (8f rank 2)
def test_log_filter_bank(oracle_backend):
    from oracle import oracle as O
    fb = F.LogFilterBank(5, f_min=50, f_max=8000, q=2.0)
    assert len(fb.center_frequencies) == 5 and abs(fb.center_frequencies[-1] - 8000) < 1e-6
    x = torch.randn(2, 3000, generator=torch.Generator().manual_seed(3))
    with pytest.raises(ValueError, match="Sample rate"):
        fb(x)
    w = fx.Wave(x, 44100) | fb                  # fs propagates to the bands through the setter
    y = w.ys
    assert tuple(y.shape) == (5, 2, 3000) and y.dtype == torch.float32
    names = [c[0] for c in oracle_backend.calls if c[0].startswith("sos")]
    assert names[-1] == "sos_bank_forward"       # one launch for all bands
    for i, f in enumerate(fb.filters):
        e, _, _ = O.iir_module_forward(x.numpy(), f._sos.numpy())
        close(y[i], e, 1e-7)
        assert tuple(f._state_x.shape) == (1, 2, 2)
    # chunked == contiguous (per-band state carried on the member filters)
    fb2 = F.LogFilterBank(5, f_min=50, f_max=8000, q=2.0, fs=44100)
    ya, yb = fb2(x[:, :1000]), fb2(x[:, 1000:])
    close(torch.cat([ya, yb], dim=-1), y.numpy(), 1e-6)
    assert F.LogFilterBank(3, fs=8000)(torch.zeros(100)).shape == (3, 100)


# ------------------------------------------------------------------ streaming (8f rank 1)
def test_stream_processor_chunked_equals_contiguous(oracle_backend):
    from scipy.signal import firwin
    from torchfx_amd.realtime import StatefulFIR, StreamProcessor
    x = torch.randn(2, 10000, generator=torch.Generator().manual_seed(5), dtype=torch.float64)
    taps = firwin(129, 0.2)

    def effects():
        return [F.HiButterworth(80, order=2), StatefulFIR(taps), F.ParametricEQ(1500, 1.2, 2.0)]
    whole = x
    for e in effects():
        e.fs = 32000
        whole = e(whole)
    sp = StreamProcessor(effects(), chunk_size=1536, overlap=0, device="cpu")
    out = sp.process_tensor(x, 32000)
    assert out.shape == x.shape
    close(out, whole.numpy(), 1e-11)                       # IIR state + FIR history carried: exact
    # the reference's way for a stateless FIR: overlap >= K-1 (IIR stages excluded: they would
    # see the overlapped samples twice)
    sp2 = StreamProcessor([F.FIR(taps)], chunk_size=2048, overlap=128, device="cpu")
    close(sp2.process_tensor(x, 32000), F.FIR(taps)(x).numpy(), 1e-11)
    with pytest.raises(ValueError, match="less than chunk_size"):
        StreamProcessor([F.FIR(taps)], chunk_size=64, overlap=64)
    with pytest.raises(ValueError, match="Nyquist"):
        StreamProcessor([F.LoButterworth(20000, order=2)], device="cpu").process_tensor(x, 32000)
    with pytest.raises(TypeError):
        StreamProcessor([nn.Identity()])


class _MockBackend:
    """The reference's test double (tests/test_realtime.py:46-120): records the stream, lets the test fire callbacks."""

    def __init__(self):
        from torchfx_amd.realtime import AudioBackend
        assert {"open_stream", "start", "stop", "close"} <= AudioBackend.__abstractmethods__
        self.config = self.callback = None
        self.start_count = self.stop_count = 0
        self.closed = False

    def open_stream(self, config, callback=None):
        self.config, self.callback = config, callback

    def start(self):
        self.start_count += 1

    def stop(self):
        self.stop_count += 1

    def close(self):
        self.closed = True

    def simulate_callback(self, x):
        out = torch.zeros(self.config.channels_out, x.shape[-1], dtype=x.dtype)
        self.callback(x, out, x.shape[-1])
        return out


def test_realtime_processor_surface_and_callback(oracle_backend):
    """tests/test_realtime.py:376-575 of the reference, on the host mirror (device='cpu')."""
    from torchfx_amd.effect import Gain
    from torchfx_amd.realtime import RealtimeError, RealtimeProcessor, StreamConfig, StreamDirection
    cfg = StreamConfig(sample_rate=48000, buffer_size=512, channels_in=2, channels_out=2)
    assert cfg.direction is StreamDirection.DUPLEX and abs(cfg.latency_ms - 512 / 48000 * 1000) < 1e-9
    assert StreamConfig(channels_in=0).direction is StreamDirection.OUTPUT
    be = _MockBackend()
    p = RealtimeProcessor([Gain(2.0)], be, cfg, device="cpu")
    assert not p.is_running and len(p.effects) == 1 and p.config is cfg
    assert abs(p.latency_ms - cfg.latency_ms) < 1e-12
    with pytest.raises(RealtimeError, match="not running"):
        p.stop()
    p.start()
    assert p.is_running and be.start_count == 1 and be.callback is not None
    with pytest.raises(RealtimeError, match="already running"):
        p.start()
    x = torch.randn(2, 512)
    close(be.simulate_callback(x), (x * 2.0).numpy(), 1e-7)
    p.stop()
    assert not p.is_running and be.stop_count == 1 and be.closed
    # chains, nn.Sequential, the context manager (stops on exceptions too)
    be = _MockBackend()
    with RealtimeProcessor(nn.Sequential(Gain(2.0), Gain(0.5)), be, cfg, device="cpu") as p:
        assert p.is_running and len(p.effects) == 2
        close(be.simulate_callback(x), x.numpy(), 1e-7)
    assert not p.is_running
    be = _MockBackend()
    with pytest.raises(RuntimeError, match="test error"):
        with RealtimeProcessor([Gain(1.0)], be, cfg, device="cpu"):
            raise RuntimeError("test error")
    assert be.stop_count == 1
    with pytest.raises(TypeError, match="inherit from FX"):
        RealtimeProcessor([nn.Identity()], be, cfg, device="cpu")
    # parameters are staged and applied at the next buffer boundary
    g = Gain(1.0)
    be = _MockBackend()
    with RealtimeProcessor([g], be, cfg, device="cpu") as p:
        p.set_parameter("0.gain", 0.5)
        assert g.gain == 1.0
        close(be.simulate_callback(x), (x * 0.5).numpy(), 1e-7)
        assert g.gain == 0.5
        p.set_parameter("7.gain", 3.0)          # bad index: ignored
        be.simulate_callback(x)
    # channel mismatch: mono to every output channel, truncation
    be = _MockBackend()
    with RealtimeProcessor([Gain(1.0)], be, StreamConfig(48000, 256, channels_in=1, channels_out=2), device="cpu"):
        m = torch.randn(1, 256)
        out = be.simulate_callback(m)
        close(out[0], m[0].numpy(), 1e-7)
        close(out[1], m[0].numpy(), 1e-7)
    be = _MockBackend()
    with RealtimeProcessor([Gain(1.0)], be, StreamConfig(48000, 256, channels_in=3, channels_out=2), device="cpu"):
        m = torch.randn(3, 256)
        close(be.simulate_callback(m), m[:2].numpy(), 1e-7)


def test_realtime_processor_blocks_are_one_continuous_signal(oracle_backend):
    """fs propagation, coefficient design at construction, carried state across callbacks, redesign + reset on a
    filter parameter change (processor.py:96-101, 231-250)."""
    from scipy.signal import firwin
    from torchfx_amd.realtime import RealtimeProcessor, StatefulFIR, StreamConfig
    cfg = StreamConfig(sample_rate=44100, buffer_size=256, channels_in=2, channels_out=2)
    lpf = F.LoButterworth(2000, order=4)
    fir = StatefulFIR(firwin(65, 0.3))
    assert lpf.fs is None
    be = _MockBackend()
    p = RealtimeProcessor([lpf, fir], be, cfg, device="cpu")
    assert lpf.fs == 44100 and lpf._has_computed_coeff
    x = torch.randn(2, 256 * 9 + 100, generator=torch.Generator().manual_seed(9), dtype=torch.float64)
    ref_l = F.LoButterworth(2000, order=4, fs=44100)
    whole = StatefulFIR(firwin(65, 0.3))(ref_l(x))
    with p:
        blocks = [be.simulate_callback(x[:, i:i + 256]) for i in range(0, x.shape[-1], 256)]     # the last one is ragged
        close(torch.cat(blocks, dim=-1), whole.numpy(), 1e-11)
        # a new cutoff: redesigned, state cleared -> equals a fresh filter on the next block
        p.set_parameter("0.cutoff", 500)
        p.reset_state()
        y = be.simulate_callback(x[:, :256])
        fresh = StatefulFIR(firwin(65, 0.3))(F.LoButterworth(500, order=4, fs=44100)(x[:, :256]))
        close(y, fresh.numpy(), 1e-11)
        assert lpf.cutoff == 500


def test_spectral_fusion_matches_staged(oracle_backend, golden):
    g = golden("chain")
    from scipy.signal import firwin
    irs = np.random.default_rng(1).standard_normal(4097) * np.exp(-np.arange(4097) / 500.0)
    irs = irs / np.abs(irs).sum()

    def pipe(spectral):
        w = fx.Wave(g["xc"], 48000)
        w.fuse_fir, w.fuse_spectral = True, spectral
        return (w | F.LoButterworth(2000, order=6) | F.ParametricEQ(frequency=1000, q=2.0, gain=3.0)
                | F.FIR(firwin(1024, 5000, fs=48000)) | F.FIR(irs))
    assert [type(m).__name__ for m in pipe(False).plan()] == ["FusedSOSCascade", "FIR"]
    ws = pipe(True)
    plan = ws.plan()
    assert [type(m).__name__ for m in plan] == ["FIR"]                 # the whole LTI run is one FIR
    oracle_backend.calls.clear()
    close(ws.ys, g["yc"], 2e-5)
    assert [c[0] for c in oracle_backend.calls] == ["fft_conv_forward"]
    # a cascade whose memory never dies out is left alone
    w = fx.Wave(g["xc"], 48000)
    w.fuse_fir = w.fuse_spectral = True
    integ = F.Biquad(100, 1.0, fs=48000)
    integ._set_coefficients(1.0, 0.0, 0.0, -1.0, 0.0)
    w = w | integ | F.Notch(50, 2.0) | F.FIR([0.5, 0.5])
    assert [type(m).__name__ for m in w.plan()] == ["FusedSOSCascade", "FIR"]


def test_fold_bound_refuses(oracle_backend, golden, monkeypatch):
    """The spectral fold's error bound: the estimate is measured per (cascade, FIR) pair and the fold is dropped when it
    exceeds the limit -- the cascade then runs in float64 (its own launch, or inside the column pass)."""
    from torchfx_amd import wave as W
    g = golden("chain")
    from scipy.signal import firwin
    irs = np.random.default_rng(1).standard_normal(4097) * np.exp(-np.arange(4097) / 500.0)

    def pipe():
        w = fx.Wave(g["xc"], 48000)
        w.fuse_fir = w.fuse_spectral = True
        return (w | F.LoButterworth(2000, order=6) | F.ParametricEQ(frequency=1000, q=2.0, gain=3.0)
                | F.FIR(firwin(1024, 5000, fs=48000)) | F.FIR(irs / np.abs(irs).sum()))
    W.plan_cache_clear()
    p = pipe().plan()
    assert [type(m).__name__ for m in p] == ["FIR"] and 0.0 < p[0].fold_error_estimate <= W.FOLD_ERROR_LIMIT
    est = p[0].fold_error_estimate
    W.plan_cache_clear()
    monkeypatch.setattr(W, "FOLD_ERROR_LIMIT", est / 4)
    p = pipe().plan()
    assert [type(m).__name__ for m in p] == ["FusedSOSCascade", "FIR"]
    assert p[0].fold_refused == pytest.approx(est)
    close(pipe().ys, g["yc"], 2e-5)


def test_default_plan_of_the_headline_chain_is_one_cascade_fir_step():
    """BASELINE cfg 5's chain at its real row length: planner-built cascade | merged 66 559-tap FIR = one CascadeFIR
    (the float64 recursion inside the overlap-save pipeline's column pass); the planning queries are host-only."""
    from torchfx_amd import wave as W
    from scipy.signal import firwin
    W.plan_cache_clear()
    ir = np.random.default_rng(0).standard_normal(65536) * np.exp(-np.arange(65536) / 8000.0)
    members = (F.LoButterworth(2000, order=6, fs=48000), F.ParametricEQ(frequency=1000, q=2.0, gain=3.0, fs=48000),
               F.FIR(firwin(1024, 5000, fs=48000)), F.FIR(ir / np.abs(ir).sum()))
    x = torch.empty((1, 28_800_064), dtype=torch.float32, device="meta")

    def plan(length=28_800_000, **flags):
        w = fx.Wave.__new__(fx.Wave)
        w._ys, w.fs, w._device, w.metadata, w._pipeline = x[:, :length], 48000, "meta", {}, list(members)
        w.fuse_fir, w.fuse_spectral, w.fuse_gain, w.fuse_epilogue, w.fuse_recursive = True, False, False, True, True
        for k, v in flags.items():
            setattr(w, k, v)
        return w.plan()
    p = plan()
    assert [type(m).__name__ for m in p] == ["CascadeFIR"] and p[0].fir.kernel.numel() == 66559 and p[0]._sos.shape == (4, 6)
    assert [type(m).__name__ for m in plan(fuse_recursive=False)] == ["FusedSOSCascade", "FIR"]
    assert [type(m).__name__ for m in plan(fuse_spectral=True)] == ["FIR"]             # the fold wins when both are on
    assert [type(m).__name__ for m in plan(length=28_800_001)] == ["CascadeFIR"]       # any row length since round 6 (row_shift)
    p18 = plan(length=1_440_000)
    assert [type(m).__name__ for m in p18] == ["FusedSOSCascade", "FIR"]               # 2^18-point blocks there ...
    assert "2^20" in p18[0].recursive_refused                                          # ... and the plan says so
    assert [type(m).__name__ for m in plan(length=2_880_000)] == ["CascadeFIR"]        # 2^20 blocks from two per row (the module itself
    #                                                                                  takes the staged pair of launches for small batches)
    g = fx.effect.Gain(0.5)
    members = members + (g,)
    p = plan()
    assert type(p[0]).__name__ == "Epilogued" and type(p[0].producer).__name__ == "CascadeFIR"


def test_cascade_fir_module_staged_fallback(oracle_backend, golden):
    """`CascadeFIR` where its kernel does not serve the rows (here: host tensors, via the test backend): the two staged launches,
    fresh state every call, epilogue handed to the FIR -- the reference's staging of `iir... | fir` (wave.py:207-239)."""
    from torchfx_amd.effect import Gain
    from torchfx_amd.filter._sos import CascadeTable
    from torchfx_amd.filter.fused import CascadeFIR
    from torchfx_amd import torchfx_ext as E
    g = golden("chain")
    f1, f2 = F.LoButterworth(2000, order=6, fs=48000), F.ParametricEQ(frequency=1000, q=2.0, gain=3.0, fs=48000)
    fir = F.FIR(np.hanning(129) / np.hanning(129).sum())
    m = CascadeFIR(CascadeTable.gather([f1, f2]), fir)
    x = torch.from_numpy(g["xc"])
    oracle_backend.calls.clear()
    y = m(x)
    assert [c[0] for c in oracle_backend.calls] == ["sos_forward", "fft_conv_forward"]
    ref = fir(F.FusedSOSCascade(f1, f2)(x))
    assert torch.equal(y, ref) and torch.equal(m(x), y)                 # stateless: a second call starts from zero state again
    ep = E.Epilogue(gain=0.5, clamp=True)
    assert torch.equal(m(x, epilogue=ep), torch.clamp(ref * 0.5, -1.0, 1.0))
    assert m(x[0]).shape == x[0].shape and m(x[None]).shape == x[None].shape
    with pytest.raises(RuntimeError, match="section taps"):
        m(x, return_sections=True)
    assert m._sos.shape == (4, 6) and m.fs == 48000


def test_plan_explains_its_routes():
    """`Wave.explain()` / `CascadeFIR.route()`: which route a tensor takes and why -- host-only queries."""
    from scipy.signal import firwin
    ir = np.random.default_rng(0).standard_normal(65536) * np.exp(-np.arange(65536) / 8000.0)
    members = (F.LoButterworth(2000, order=6, fs=48000), F.ParametricEQ(frequency=1000, q=2.0, gain=3.0, fs=48000),
               F.FIR(firwin(1024, 5000, fs=48000)), F.FIR(ir / np.abs(ir).sum()))

    def wave(x):
        w = fx.Wave.__new__(fx.Wave)
        w._ys, w.fs, w._device, w.metadata, w._pipeline = x, 48000, "meta", {}, list(members)
        w.fuse_fir, w.fuse_spectral, w.fuse_gain, w.fuse_epilogue, w.fuse_recursive = True, False, False, True, True
        return w
    lines = wave(torch.empty((64, 28_800_001), dtype=torch.float32, device="meta")).explain()
    assert len(lines) == 1 and lines[0].startswith("CascadeFIR: staged -- host tensor")        # a meta tensor is not a device tensor
    step = wave(torch.empty((64, 28_800_001), dtype=torch.float32, device="meta")).plan()[0]

    class FakeCuda:                                       # what route() reads of a tensor
        def __init__(self, shape, dtype=torch.float32):
            self.shape, self.dtype, self.ndim, self.is_cuda = shape, dtype, len(shape), True
            self.device = torch.device("cuda", 0)

        def reshape(self, *a):
            return FakeCuda((int(np.prod(self.shape[:-1])), self.shape[-1]), self.dtype)
    path, why, info = step.route(FakeCuda((64, 28_800_001)))
    assert path == "fused" and "2097152-point blocks" in why and "480 frame pairs" in why and info["N"] == 1 << 21
    path, why, _ = step.route(FakeCuda((2, 28_800_001)))
    assert path == "staged" and "15 frame pairs < 256" in why
    assert step.route(FakeCuda((2, 28_800_001)), return_sections=True)[0] == "fused"                # section taps come from the fused pass
    path, why, _ = step.route(FakeCuda((64, 28_800_001), torch.float64))
    assert path == "staged" and "float64" in why
    short = wave(torch.empty((64, 1_440_000), dtype=torch.float32, device="meta")).explain()
    assert short[0].startswith("FusedSOSCascade: staged -- ") and "2^20" in short[0] and short[1] == "FIR"


def _gain_chain(wave):
    from scipy.signal import firwin
    irg = np.random.default_rng(2).standard_normal(2049) * np.exp(-np.arange(2049) / 300.0)
    return (wave | F.HiShelving(3000, q=0.7, gain=4.0) | F.ParametricEQ(frequency=500, q=4.0, gain=9.0)
            | F.HiButterworth(80, order=2) | F.FIR(firwin(257, 6000, fs=48000)) | F.FIR(8.0 * irg / np.abs(irg).sum()))


def test_spectral_plan_folds_a_cascade_with_gain_into_the_fir_run(oracle_backend, golden, monkeypatch):
    """Reference output of the staged chain (tests/golden/chain_gain.npz) vs the folded plan, which is a
    single FIR: the 3-filter IIR run as taps, convolved with both FIRs.  The planner prices the fold in HBM bytes
    (`Wave._ols_bytes_per_sample`): the merged 2305-tap FIR alone runs on the one-launch 8192-point kernel at 9.6 B/sample,
    with the cascade's impulse response folded in it would need the three-pass pipeline (~26 B/sample) -- so by default the
    cascade stays its own 8 B/sample pass; without the 8192- and 16 384-point kernels the fold pays."""
    g = golden("chain_gain")
    monkeypatch.setenv("TORCHFX_AMD_FUSE_SPECTRAL", "1")           # the fold is opt-in since round 5
    w = _gain_chain(fx.Wave(g["x"], 48000))
    assert w.fuse_spectral
    assert [type(m).__name__ for m in w.plan()] == ["FusedSOSCascade", "FIR"]
    close(w.ys, g["y"], 2e-5)
    monkeypatch.setenv("TFX_OLS_LDS8K_MINK", "0")
    monkeypatch.setenv("TFX_OLS_LDS16K", "0")
    w = _gain_chain(fx.Wave(g["x"], 48000))
    plan = w.plan()
    assert [type(m).__name__ for m in plan] == ["FIR"] and plan[0].kernel.numel() > 257 + 2049
    assert 0.0 < plan[0].fold_error_estimate <= fx.wave.FOLD_ERROR_LIMIT        # measured on the host, printed by bench.py
    oracle_backend.calls.clear()
    close(w.ys, g["y"], 2e-5)
    assert [c[0] for c in oracle_backend.calls] == ["fft_conv_forward"]
    assert np.allclose(np.vstack([m._sos.numpy() for m in _gain_chain(fx.Wave(g["x"], 48000))._pipeline[:3]]), g["sos"],
                       rtol=0, atol=0)          # the designers reproduce the reference's SOS bit for bit


def test_pad_to_and_unfold_helpers():
    from torchfx_amd.filter._fftconv import pad_to, unfold
    x = torch.ones(5)
    assert torch.equal(pad_to(x, 8), torch.tensor([1., 1, 1, 1, 1, 0, 0, 0]))
    assert pad_to(torch.randn(3, 4, 5), 8).shape == (3, 4, 8)
    y = torch.randn(10)
    assert torch.equal(pad_to(y, 10), y)
    a = torch.arange(20, dtype=torch.float32)
    fr = unfold(a, kernel_size=5, stride=3)
    assert fr.shape == (6, 5)
    for i in range(5):
        assert torch.equal(fr[i], a[3 * i:3 * i + 5])
    assert torch.equal(fr[5], torch.tensor([15., 16, 17, 18, 19]))
    assert unfold(torch.randn(100), 10, 5).shape == (19, 10)
    assert unfold(torch.randn(4, 2, 100), 10, 5).shape == (4, 2, 19, 10)
    assert unfold(torch.randn(17), 5, 3).shape[0] == 5          # tail frame zero-padded


def test_wave_transform_and_merge():
    a = fx.Wave(torch.ones(2, 10), 8000)
    b = fx.Wave(2 * torch.ones(2, 6), 8000)
    m = fx.Wave.merge([a, b])
    assert m.ys.shape == (2, 10) and float(m.ys[0, 0]) == 3.0 and float(m.ys[0, 9]) == 1.0
    assert fx.Wave.merge([a, b], split_channels=False).fs == 8000
    assert fx.Wave.merge([a, fx.Wave(torch.zeros(1, 10), 8000)], split_channels=True).channels() == 3
    with pytest.raises(ValueError, match="mismatch"):
        fx.Wave.merge([a, fx.Wave(torch.ones(2, 4), 16000)])
    with pytest.raises(ValueError, match="No waves"):
        fx.Wave.merge([])
    t = a.transform(lambda y, k: y * k, 4.0)
    assert isinstance(t, fx.Wave) and float(t.ys[1, 3]) == 4.0 and len(t) == 10 and t.duration("ms") == 1.25


def test_parallel_combination_routing(oracle_backend):
    """`f1 + f2 + f3` of SOS filters is one sum-mode launch (left-nested combinations are flattened);
    a branch that is not an SOS filter sends the whole combination down the branch-by-branch path."""
    from torchfx_amd import filter as F
    x = torch.from_numpy(np.random.default_rng(0).standard_normal((2, 500)).astype(np.float32))
    a, b, c = (F.LoButterworth(800, order=4, fs=8000), F.HiButterworth(1500, order=2, fs=8000),
               F.BiquadLPF(cutoff=1000, q=0.7, fs=8000))
    comb = a + b + c
    oracle_backend.calls.clear()
    y = comb(x)
    assert [k[0] for k in oracle_backend.calls] == ["sos_bank_sum_forward"] and oracle_backend.calls[0][2] == 3
    ref = [F.LoButterworth(800, order=4, fs=8000), F.HiButterworth(1500, order=2, fs=8000), F.BiquadLPF(cutoff=1000, q=0.7, fs=8000)]
    exp = torch.zeros_like(x)
    for f in ref:
        exp += f(x)
    assert torch.equal(y, exp)
    assert a._state_x.shape == (2, 2, 2) and b._state_x.shape == (1, 2, 2) and c._state_x.shape == (1, 2, 2)
    y2 = comb(x)                                          # stateful: continues where the first call stopped
    exp2 = torch.zeros_like(x)
    for f in ref:
        exp2 += f(x)
    assert torch.equal(y2, exp2) and not torch.equal(y2, y)
    mixed = F.LoButterworth(800, order=2, fs=8000) + F.FIR([0.5, 0.5])
    oracle_backend.calls.clear()
    mixed(x)
    assert [k[0] for k in oracle_backend.calls][-1] == "sum_forward"


def test_precision_auto_estimate_is_a_replay_of_the_float32_kernel():
    """tfx_sos_plan_info's float32 error estimate (host replay of the kernel's float32 arithmetic): tiny for
    well-conditioned cascades, enormous for poles next to z = 1, and 'auto' draws the line at 2e-5."""
    import scipy.signal as sg
    from torchfx_amd import torchfx_ext as E
    easy = E.sos_plan_info(sg.butter(4, 2000 / 24000, output="sos"))
    assert easy["auto_precision"] == "f32" and 1e-7 < easy["f32_error_bound"] < 5e-6
    dc = E.sos_plan_info(sg.butter(2, 20 / 24000, "highpass", output="sos"))          # poles at radius 0.998, 20 Hz
    assert dc["auto_precision"] == "f64" and dc["f32_error_bound"] > 1e-2
    f1, f2 = F.LoButterworth(2000, order=6, fs=48000), F.ParametricEQ(frequency=1000, q=2.0, gain=3.0, fs=48000)
    f1.compute_coefficients(), f2.compute_coefficients()
    cfg2 = E.sos_plan_info(torch.cat([f1._sos, f2._sos]))
    assert cfg2["auto_precision"] == "f32" and 5e-6 < cfg2["f32_error_bound"] < 2e-5   # measured on the device: 6.2e-6
    unstable = E.sos_plan_info(np.array([[1.0, 0, 0, 1, -2.1, 1.2]]))
    assert unstable["auto_precision"] == "f64"


def test_host_taps_are_converted_once_per_module_buffer():
    """FIR.forward hands `self.kernel.reshape(-1)` (a new view object per call) to the backend; the host copy in the
    signal's dtype must be made once per buffer, not once per call (a fresh quarter-megabyte host allocation per step
    stalls the GPU queues of the process, docs/HISTORY.md section 6.2)."""
    from torchfx_amd.torchfx_ext import _kernel_host
    buf = torch.randn(1, 1, 70000, dtype=torch.float64)
    a = _kernel_host(buf.reshape(-1), torch.float32)
    b = _kernel_host(buf.reshape(-1), torch.float32)
    assert a is b and a.dtype == torch.float32 and a.data_ptr() != buf.data_ptr()
    assert _kernel_host(buf.reshape(-1), torch.float64).data_ptr() == buf.data_ptr()      # nothing to convert: no copy at all
    buf.mul_(2.0)                                                                          # in-place edit: new copy
    c = _kernel_host(buf.reshape(-1), torch.float32)
    assert c is not a and torch.equal(c, buf.reshape(-1).float())
    half = _kernel_host(buf.reshape(-1)[:100], torch.float32)                              # another window of the same buffer
    assert half.numel() == 100 and torch.equal(half, buf.reshape(-1)[:100].float())


# ------------------------------------------------------------------ plan cache and the ADVICE r2 planner fixes
def _cfg5_members():
    from scipy.signal import firwin
    ir = np.random.default_rng(5).standard_normal(4097) * np.exp(-np.arange(4097) / 500.0)
    return (F.LoButterworth(2000, order=4, fs=48000), F.ParametricEQ(frequency=1000, q=2.0, gain=3.0, fs=48000),
            F.FIR(firwin(1024, 5000, fs=48000)), F.FIR(ir / np.abs(ir).sum()))


def test_plan_cache_hits_and_invalidates():
    from torchfx_amd import wave as W
    W.plan_cache_clear()
    f1, f2, fir, rev = _cfg5_members()
    x = torch.zeros(2, 100_000)

    def plan(members=(f1, f2, fir, rev), length=100_000):
        w = fx.Wave(x[:, :length], 48000)
        w.fuse_spectral = True                                        # the fold: the plan that derives the most (opt-in)
        for m in members:
            w = w | m
        return w.plan()
    p0 = plan()
    assert [type(m).__name__ for m in p0] == ["FIR"] and len(W._PLANS) == 1
    p1 = plan()
    assert p1[0] is p0[0] and len(W._PLANS) == 1                  # same merged module, nothing re-derived
    # another row length is another plan entry but re-uses the merged taps (content caches)
    p2 = plan(length=90_000)
    assert len(W._PLANS) == 2 and p2[0] is p0[0]
    # re-design: compute_coefficients assigns a new SOS tensor -> miss, new taps
    f1.cutoff = 3000
    f1.compute_coefficients()
    p3 = plan()
    assert p3[0] is not p0[0] and not torch.equal(p3[0].kernel, p0[0].kernel)
    # in-place edit of FIR taps bumps the version counter -> miss
    rev.kernel.mul_(0.5)
    p4 = plan()
    assert p4[0] is not p3[0]
    assert torch.allclose(p4[0].kernel, 0.5 * p3[0].kernel, rtol=1e-12, atol=1e-18)
    # a Gain's settings are part of the key (fuse_gain bakes them into coefficients)
    g = fx.effect.Gain(2.0)
    w = fx.Wave(x, 48000); w.fuse_gain = True
    a = (w | f1 | g | f2).plan()
    g.gain = 4.0
    w = fx.Wave(x, 48000); w.fuse_gain = True
    b = (w | f1 | g | f2).plan()
    assert not torch.equal(a[0]._sos, b[0]._sos)
    # flags are part of the key
    w = fx.Wave(x, 48000); w.fuse_spectral = False
    assert [type(m).__name__ for m in (w | f1 | f2 | fir | rev).plan()] == ["FusedSOSCascade", "FIR"]


def test_cached_plans_hand_out_fresh_cascades(oracle_backend):
    """The reference builds a new FusedSOSCascade per materialisation (wave.py:216-233): a cached plan must not
    carry the previous wave's state into the next one."""
    from torchfx_amd import wave as W
    W.plan_cache_clear()
    f1, f2, _, _ = _cfg5_members()
    x = torch.from_numpy(np.random.default_rng(0).standard_normal((2, 5000))).float()
    w = fx.Wave(x, 48000); w.fuse_spectral = False
    pa = (w | f1 | f2).plan()
    ya = pa[0](x)
    assert pa[0]._state_x is not None
    w = fx.Wave(x, 48000); w.fuse_spectral = False
    pb = (w | f1 | f2).plan()
    assert pb[0] is not pa[0] and pb[0]._state_x is None and pb[0]._stream.table is pa[0]._stream.table
    assert torch.equal(pb[0](x), ya)
    y1 = (fx.Wave(x, 48000) | f1 | f2).ys
    y2 = (fx.Wave(x, 48000) | f1 | f2).ys
    assert torch.equal(y1, y2) and torch.equal(y1, ya)


def test_steady_state_planning_is_cheap():
    import time
    from torchfx_amd import wave as W
    W.plan_cache_clear()
    f1, f2, fir, rev = _cfg5_members()
    x = torch.zeros(2, 200_000)
    (fx.Wave(x, 48000) | f1 | f2 | fir | rev).plan()
    t0 = time.perf_counter()
    for _ in range(20):
        (fx.Wave(x, 48000) | f1 | f2 | fir | rev).plan()
    assert (time.perf_counter() - t0) / 20 < 1e-3            # <= 1 ms host planning in steady state


def test_user_held_cascade_and_stateful_fir_stay_staged(oracle_backend):
    """(1) a FusedSOSCascade the user holds is stateful across waves and must never be folded into the
    FIR behind it -- chunked use carries its state; (2) StatefulFIR (own forward, history, no epilogue kwarg) is
    neither merged, nor folded into, nor given an epilogue."""
    from torchfx_amd.effect import Gain, Normalize
    from torchfx_amd.realtime import StatefulFIR
    f1, f2, fir, rev = _cfg5_members()
    held = F.FusedSOSCascade(f1, f2)
    x = torch.from_numpy(np.random.default_rng(1).standard_normal((2, 40_000))).float()
    w = fx.Wave(x, 48000)
    assert w.fuse_recursive and w.fuse_fir and w.fuse_epilogue and not w.fuse_spectral      # the default policy
    w.fuse_spectral = True                                              # ... and with the fold switched on as well
    plan = (w | held | fir).plan()
    assert plan[0] is held and [type(m).__name__ for m in plan] == ["FusedSOSCascade", "FIR"]
    # chunked == one shot through the default plan (state carried on the user's object)
    whole = (fx.Wave(x, 48000) | F.FusedSOSCascade(f1, f2) | fir).ys
    held.reset_state()
    a = (fx.Wave(x[:, :25_000], 48000) | held).ys
    b = (fx.Wave(x[:, 25_000:], 48000) | held).ys
    y_iir = torch.cat([a, b], dim=1)
    close(fir(y_iir), whole.numpy(), 1e-5)
    s1, s2 = StatefulFIR(np.ones(8) / 8), StatefulFIR([0.5, 0.5])
    plan = (fx.Wave(x, 48000) | s1 | s2 | Gain(0.5) | Normalize()).plan()
    assert plan[0] is s1 and plan[1] is s2 and [type(m).__name__ for m in plan[2:]] == ["Gain", "Normalize"]
    plan = (fx.Wave(x, 48000) | f1 | f2 | s1).plan()
    assert [type(m).__name__ for m in plan] == ["FusedSOSCascade", "StatefulFIR"]
    # ... and they run: history carried over two waves equals one shot
    s1.reset_state()
    ya = (fx.Wave(x[:, :10_000], 48000) | s1 | Gain(0.5)).ys
    yb = (fx.Wave(x[:, 10_000:], 48000) | s1 | Gain(0.5)).ys
    s1.reset_state()
    close(torch.cat([ya, yb], dim=1), (fx.Wave(x, 48000) | s1 | Gain(0.5)).ys.numpy(), 1e-6)


# ------------------------------------------------------------------ fused per-chunk run of the stream processors
def test_stream_processor_fuses_small_chunks_and_keeps_member_state(oracle_backend):
    from torchfx_amd.effect import Gain
    from torchfx_amd.realtime import StatefulFIR, StreamProcessor, _ChunkRun
    fs = 48000
    f1, f2 = F.LoButterworth(3000, order=4, fs=fs), F.ParametricEQ(frequency=800, q=2.0, gain=4.0, fs=fs)
    fir, g = StatefulFIR(np.hanning(65) / np.hanning(65).sum(), conv_mode="fft"), Gain(1.7, clamp=True)
    x = torch.from_numpy(np.random.default_rng(3).standard_normal((2, 6000))).float()
    ref = Gain(1.7, clamp=True)(F.FIR(fir.kernel.flip(-1).reshape(-1).numpy())(
        F.ParametricEQ(frequency=800, q=2.0, gain=4.0, fs=fs)(F.LoButterworth(3000, order=4, fs=fs)(x))))
    sp = StreamProcessor([f1, f2, fir, g], chunk_size=512, device="cpu")
    assert len(sp._segments) == 1 and isinstance(sp._segments[0], _ChunkRun)
    oracle_backend.calls.clear()
    y = sp.process_tensor(x, fs)
    assert {c[0] for c in oracle_backend.calls} == {"chunk_forward"} and len(oracle_backend.calls) == 12   # one launch per chunk
    close(y, ref.numpy(), 2e-6)
    # the members still own their state: [K, C, 2] views of the combined tensors, the FIR's history
    run = sp._segments[0]
    assert f1._state_y.shape == (2, 2, 2) and f2._state_y.shape == (1, 2, 2) and fir._hist.shape == (2, 64)
    assert f1._state_y.data_ptr() == run._sy.data_ptr() and f2._state_y.data_ptr() == run._sy[2:].data_ptr()
    # a member reset in between is honoured: the next chunk starts that member from silence, the others continue
    sp2 = StreamProcessor([F.LoButterworth(3000, order=4, fs=fs), F.ParametricEQ(frequency=800, q=2.0, gain=4.0, fs=fs)],
                          chunk_size=1000, device="cpu")
    a = sp2._run(x[:, :1000])
    sp2.effects[1].reset_state()
    b = sp2._run(x[:, 1000:2000])
    m1, m2 = F.LoButterworth(3000, order=4, fs=fs), F.ParametricEQ(frequency=800, q=2.0, gain=4.0, fs=fs)
    a_ref = m2(m1(x[:, :1000]))
    m2.reset_state()
    b_ref = m2(m1(x[:, 1000:2000]))
    # several IIR members run as ONE float64 cascade (rounded once, like the Wave planner's fusion); the staged modules
    # round to float32 between members: equal to a float32 ulp, not bit for bit
    close(a, a_ref.numpy(), 3e-7)
    close(b, b_ref.numpy(), 3e-7)
    # chunks the fused kernel does not take run member by member; lone effects are not wrapped
    big = StreamProcessor([F.LoButterworth(3000, order=4, fs=fs), Gain(0.5)], chunk_size=8192, device="cpu")
    oracle_backend.calls.clear()
    big.process_tensor(torch.zeros(2, 8192), fs)
    assert [c[0] for c in oracle_backend.calls] == ["sos_forward", "gain_forward"]
    assert not any(isinstance(s, _ChunkRun) for s in StreamProcessor([F.LoButterworth(3000, order=4, fs=fs)], 512, device="cpu")._segments)
