"""The reference's own hot-path test grids, restated for the HIP backend: same seeds,
shapes, parameter grids and tolerances as

    tests/test_fftconv.py:64-122      fft_conv1d == F.conv1d           (K x T grid, asymmetric padding, errors)
    tests/test_fir.py:79-131          FIR fft mode == direct mode      (taps 5 ... 1024 on [2, 44100], shapes)
    tests/test_fused.py:116-259       FusedSOSCascade == scipy sosfilt (seeds 0, 1, 2, 42; orders 6 / 12 / 20; chunks)
    tests/test_ops_dispatch.py:29-167 the dispatch layer               (pass-through, states, SciPy, delay line)
    tests/test_chain_fusion.py:51-226 deferred fusion == sequential    (Wave | ..., FilterChain, nn.Sequential)

of the reference repository -- our code, our modules, device tensors; expected values come from SciPy or
from torch's CPU ``conv1d`` (third-party arithmetic the reference itself compares against), never from the
reference package.  Where the reference only checks 1e-4 we additionally pin the result against a float64
SciPy computation at our own tolerance.
"""
import numpy as np
import pytest
import scipy.signal as sps
import torch
import torch.nn.functional as TF
from scipy.signal import firwin
from torch import nn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SAMPLE_RATE = 44100
ATOL = RTOL = 1e-4            # the reference's bar in all five files


def dev(t):
    return t.to(DEV)


def allclose(a, b, atol=ATOL, rtol=RTOL):
    return torch.allclose(a.detach().cpu().double(), b.detach().cpu().double(), atol=atol, rtol=rtol)


# ------------------------------------------------------------------ tests/test_fftconv.py:64-122
class TestFftConv1d:
    @pytest.mark.parametrize("K", [3, 5, 16, 32, 64, 128, 256])
    def test_matches_conv1d(self, K):
        from torchfx_amd.filter._fftconv import fft_conv1d
        torch.manual_seed(K)
        T = 4410
        x, w = torch.randn(1, 1, T), torch.randn(1, 1, K)
        expected = TF.conv1d(TF.pad(x, (K - 1, 0)), w)
        result = fft_conv1d(dev(x), dev(w), padding=(K - 1, 0))
        assert result.shape == expected.shape and result.is_cuda
        assert allclose(result, expected)
        ref64 = sps.correlate(np.pad(x[0, 0].double().numpy(), (K - 1, 0)), w[0, 0].double().numpy(), mode="valid")
        assert np.abs(result[0, 0].cpu().numpy() - ref64).max() <= 1e-5 * max(1.0, np.abs(ref64).max())

    def test_multichannel(self):
        from torchfx_amd.filter._fftconv import fft_conv1d
        torch.manual_seed(0)
        x, w = torch.randn(2, 4, 1000), torch.randn(1, 1, 32)
        result = fft_conv1d(dev(x), dev(w), padding=(31, 0))
        assert result.shape == (2, 4, 1000)
        for b in range(2):
            for c in range(4):
                expected = TF.conv1d(TF.pad(x[b:b + 1, c:c + 1], (31, 0)), w)
                assert allclose(result[b, c], expected[0, 0], rtol=0)

    def test_raises_on_short_input(self):
        from torchfx_amd.filter._fftconv import fft_conv1d
        with pytest.raises(RuntimeError, match="kernel size"):
            fft_conv1d(dev(torch.randn(1, 1, 5)), dev(torch.randn(1, 1, 10)))

    def test_raises_on_bad_block_ratio(self):
        from torchfx_amd.filter._fftconv import fft_conv1d
        with pytest.raises(RuntimeError, match="Block ratio"):
            fft_conv1d(dev(torch.randn(1, 1, 100)), dev(torch.randn(1, 1, 5)), block_ratio=0.5)

    @pytest.mark.parametrize("T", [100, 1000, 44100])
    def test_various_lengths(self, T):
        from torchfx_amd.filter._fftconv import fft_conv1d
        torch.manual_seed(T)
        x, w = torch.randn(1, 1, T), torch.randn(1, 1, 64)
        expected = TF.conv1d(TF.pad(x, (63, 0)), w)
        result = fft_conv1d(dev(x), dev(w), padding=(63, 0))
        assert result.shape == expected.shape and allclose(result, expected)

    def test_symmetric_padding(self):
        from torchfx_amd.filter._fftconv import fft_conv1d
        torch.manual_seed(1)
        x, w = torch.randn(1, 1, 200), torch.randn(1, 1, 16)
        result = fft_conv1d(dev(x), dev(w), padding=(8, 7))
        expected = TF.conv1d(TF.pad(x, (8, 7)), w)
        assert result.shape == expected.shape and allclose(result, expected)


# ------------------------------------------------------------------ tests/test_fir.py:64-131
def test_conv_mode_default_and_validation():
    from torchfx_amd.filter import FIR
    assert FIR([0.2, 0.2, 0.2, 0.2, 0.2])._conv_mode == "fft"
    with pytest.raises(ValueError, match="conv_mode"):
        FIR([0.2, 0.2, 0.2], conv_mode="invalid")


@pytest.mark.parametrize("num_taps", [5, 32, 64, 128, 256, 512, 1024])
def test_fft_matches_direct(num_taps):
    from torchfx_amd.filter import FIR
    torch.manual_seed(num_taps)
    b = firwin(num_taps, 5000, fs=44100, window="hamming")
    signal = torch.randn(2, 44100)
    out_fft = FIR(b, conv_mode="fft")(dev(signal))
    out_direct = FIR(b, conv_mode="direct")(dev(signal))
    assert allclose(out_fft, out_direct)
    # no reference test pins FIR against an independent oracle: float64 lfilter with the float32 taps
    ref = sps.lfilter(b.astype(np.float32).astype(np.float64), [1.0], signal.double().numpy(), axis=-1)
    for out in (out_fft, out_direct):
        assert np.abs(out.cpu().numpy() - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("shape", [(44100,), (2, 44100), (4, 2, 44100)], ids=["mono", "stereo", "batch"])
def test_fft_conv_shapes(shape):
    from torchfx_amd.filter import FIR
    signal = dev(torch.randn(*shape))
    result = FIR([0.1, 0.15, 0.5, 0.15, 0.1], conv_mode="fft")(signal)
    assert result.shape == signal.shape and result.dtype == signal.dtype


@pytest.mark.parametrize("mode", ["fft", "direct"])
def test_designable_fir_conv_mode(mode):
    from torchfx_amd.filter import DesignableFIR
    fir = DesignableFIR(cutoff=5000, num_taps=101, fs=44100, conv_mode=mode)
    assert fir._conv_mode == mode
    signal = dev(torch.randn(44100))
    assert fir(signal).shape == signal.shape


def test_fft_conv_short_signal():
    from torchfx_amd.filter import FIR
    b = firwin(32, 5000, fs=44100, window="hamming")
    signal = dev(torch.randn(1, 100))
    out_fft, out_direct = FIR(b, conv_mode="fft")(signal), FIR(b, conv_mode="direct")(signal)
    assert out_fft.shape == signal.shape and allclose(out_fft, out_direct)


# ------------------------------------------------------------------ tests/test_fused.py:100-259
def _reference_sosfilt(filters, x):
    y = x.copy()
    for f in filters:
        y = sps.sosfilt(f._sos.detach().cpu().numpy().astype(np.float64), y, axis=-1)
    return y


class TestFusedConstruction:
    def test_from_chain_single_and_errors(self):
        from torchfx_amd.filter import FusedSOSCascade, LoButterworth
        f = LoButterworth(cutoff=2000, order=4, fs=SAMPLE_RATE)
        fused = FusedSOSCascade.from_chain(f)
        assert fused._num_sections == f._sos.shape[0]
        with pytest.raises(TypeError, match="Expected nn.Sequential or IIR/Biquad"):
            FusedSOSCascade.from_chain(nn.ReLU())
        with pytest.raises(ValueError, match="No IIR/Biquad filters"):
            FusedSOSCascade.from_chain(nn.Sequential(nn.Identity()))


class TestFusedNumerics:
    @pytest.mark.parametrize("seed", [0, 1, 2])
    def test_single_filter_matches_sosfilt(self, seed):
        from torchfx_amd.filter import FusedSOSCascade, LoButterworth
        torch.manual_seed(seed)
        f = LoButterworth(cutoff=2000, order=4, fs=SAMPLE_RATE)
        x = torch.randn(2, SAMPLE_RATE, dtype=torch.float64)
        y = FusedSOSCascade(f)(dev(x))
        ref = _reference_sosfilt([f], x.numpy())
        np.testing.assert_allclose(y.cpu().numpy(), ref, atol=ATOL, rtol=RTOL)
        assert np.abs(y.cpu().numpy() - ref).max() <= 2e-11 * max(1.0, np.abs(ref).max())     # our float64 bar

    def test_multi_filter_matches_sequential_sosfilt(self):
        from torchfx_amd.filter import FusedSOSCascade, HiButterworth, LoButterworth
        torch.manual_seed(42)
        f1 = LoButterworth(cutoff=4000, order=4, fs=SAMPLE_RATE)
        f2 = HiButterworth(cutoff=200, order=2, fs=SAMPLE_RATE)
        x = torch.randn(2, SAMPLE_RATE, dtype=torch.float64)
        y = FusedSOSCascade(f1, f2)(dev(x))
        ref = _reference_sosfilt([f1, f2], x.numpy())
        np.testing.assert_allclose(y.cpu().numpy(), ref, atol=ATOL, rtol=RTOL)
        assert np.abs(y.cpu().numpy() - ref).max() <= 2e-11 * max(1.0, np.abs(ref).max())


class TestFusedShapesStateDtype:
    def _make(self):
        from torchfx_amd.filter import FusedSOSCascade, LoButterworth
        return FusedSOSCascade(LoButterworth(cutoff=2000, order=4, fs=SAMPLE_RATE))

    @pytest.mark.parametrize("shape", [(SAMPLE_RATE,), (2, SAMPLE_RATE), (4, 2, SAMPLE_RATE)])
    def test_input_shapes(self, shape):
        x = dev(torch.randn(*shape))
        assert self._make()(x).shape == x.shape

    def test_reset_state(self):
        fused = self._make()
        _ = fused(dev(torch.randn(2, 1024)))
        assert fused._state_x is not None and fused._stateful is True
        fused.reset_state()
        assert fused._state_x is None and fused._state_y is None and fused._stateful is False

    def test_chunked_equals_contiguous(self):
        torch.manual_seed(7)
        x = dev(torch.randn(2, 2048, dtype=torch.float64))
        y_all = self._make()(x)
        chunked = self._make()
        y_chunked = torch.cat([chunked(x[:, :1024]), chunked(x[:, 1024:])], dim=-1)
        torch.testing.assert_close(y_chunked, y_all, atol=ATOL, rtol=RTOL)
        assert float((y_chunked - y_all).abs().max()) <= 1e-12

    def test_channel_resize_reallocates_state(self):
        fused = self._make()
        y1 = fused(dev(torch.randn(2, 512)))
        assert fused._state_x.shape[1] == 2
        y2 = fused(dev(torch.randn(4, 512)))
        assert fused._state_x.shape[1] == 4 and y1.shape == (2, 512) and y2.shape == (4, 512)

    def test_move_coeff_keeps_host_table(self):
        fused = self._make()
        fused.move_coeff("cpu")
        assert fused._sos.device.type == "cpu"

    @pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
    def test_preserves_input_dtype(self, dtype):
        assert self._make()(dev(torch.randn(2, 512, dtype=dtype))).dtype == dtype

    @pytest.mark.parametrize("order", [6, 12, 20])
    def test_high_order_stable(self, order):
        from torchfx_amd.filter import FusedSOSCascade, LoButterworth
        torch.manual_seed(0)
        f = LoButterworth(cutoff=2000, order=order, fs=SAMPLE_RATE)
        x = torch.randn(2, SAMPLE_RATE, dtype=torch.float64)
        y = FusedSOSCascade(f)(dev(x))
        assert torch.isfinite(y).all()
        np.testing.assert_allclose(y.cpu().numpy(), _reference_sosfilt([f], x.numpy()), atol=ATOL, rtol=RTOL)


# ------------------------------------------------------------------ tests/test_ops_dispatch.py:20-167
class TestOpsDispatch:
    def test_availability(self):
        from torchfx_amd import _ops, native
        assert _ops.PARALLEL_SCAN_THRESHOLD == 2048 and _ops.is_native_available() is True
        ext = native.load()
        assert all(hasattr(ext, n) for n in ("biquad_forward", "sos_forward", "delay_line_forward"))

    def test_biquad_passthrough_states_roundtrip(self):
        from torchfx_amd import _ops
        b = torch.tensor([1.0, 0.0, 0.0], dtype=torch.float64)
        a = torch.tensor([1.0, 0.0, 0.0], dtype=torch.float64)
        x = dev(torch.randn(2, 256, dtype=torch.float64))
        y, sx, sy = _ops.biquad_forward(x, b, a, None, None)
        torch.testing.assert_close(y, x)
        assert y.shape == x.shape and sx.shape == (2, 2) and sy.shape == (2, 2)
        y2, sx2, sy2 = _ops.biquad_forward(dev(torch.randn(2, 128, dtype=torch.float64)), b, a, sx, sy)
        assert y2.shape == (2, 128) and sx2.shape == sx.shape and sy2.shape == sy.shape

    def test_biquad_matches_scipy(self):
        from torchfx_amd import _ops
        torch.manual_seed(42)
        sos = sps.butter(2, 2000 / (0.5 * 44100), btype="lowpass", output="sos")
        x = torch.randn(2, 1024, dtype=torch.float64)
        y, _, _ = _ops.biquad_forward(dev(x), torch.tensor(sos[0, :3]), torch.tensor(sos[0, 3:]), None, None)
        np.testing.assert_allclose(y.cpu().numpy(), sps.sosfilt(sos, x.numpy(), axis=-1), atol=1e-6, rtol=1e-6)

    def test_sos_passthrough_states_and_scipy(self):
        from torchfx_amd import _ops
        ident = torch.tensor([[1.0, 0, 0, 1, 0, 0], [1.0, 0, 0, 1, 0, 0]], dtype=torch.float64)
        x = dev(torch.randn(2, 256, dtype=torch.float64))
        y, sx, sy = _ops.parallel_iir_forward(x, ident, None, None)
        torch.testing.assert_close(y, x)
        assert sx.shape == (2, 2, 2) and sy.shape == (2, 2, 2)
        torch.manual_seed(42)
        sos_np = sps.butter(4, 2000 / (0.5 * 44100), btype="lowpass", output="sos")
        x = torch.randn(2, 1024, dtype=torch.float64)
        y, _, _ = _ops.parallel_iir_forward(dev(x), torch.tensor(sos_np), None, None)
        np.testing.assert_allclose(y.cpu().numpy(), sps.sosfilt(sos_np, x.numpy(), axis=-1), atol=1e-4, rtol=1e-4)

    def test_delay_line(self):
        from torchfx_amd import _ops
        x = dev(torch.randn(2, 512, dtype=torch.float64))
        y = _ops.delay_line_forward(x, delay_samples=100, decay=0.5, mix=0.8)
        assert y.shape == x.shape
        torch.testing.assert_close(y[:, :100], x[:, :100])
        assert not torch.allclose(y[:, 100:], x[:, 100:])
        x = dev(torch.arange(10, dtype=torch.float64).unsqueeze(0))
        y = _ops.delay_line_forward(x, delay_samples=3, decay=0.5, mix=1.0)
        expected = x.clone()
        expected[0, 3:] = x[0, 3:] + 0.5 * x[0, :-3]
        torch.testing.assert_close(y, expected)
        x = dev(torch.randn(2, 50, dtype=torch.float64))
        torch.testing.assert_close(_ops.delay_line_forward(x, delay_samples=100, decay=0.5, mix=0.5), x)
        x = dev(torch.randn(512, dtype=torch.float64))
        assert _ops.delay_line_forward(x, delay_samples=50, decay=0.5, mix=0.5).shape == x.shape


# ------------------------------------------------------------------ tests/test_chain_fusion.py:51-226
FS16, N16 = 16000, 1600


def _wave():
    from torchfx_amd import Wave
    torch.manual_seed(42)
    return Wave(torch.randn(1, N16, dtype=torch.float64), FS16, device=DEV)


def _prepare(fs, *filters):
    from torchfx_amd.effect import FX
    from torchfx_amd.filter._base import AbstractFilter
    for f in filters:
        if isinstance(f, FX):
            if hasattr(f, "fs") and f.fs is None:
                f.fs = fs
            if isinstance(f, AbstractFilter) and not f._has_computed_coeff:
                f.compute_coefficients()


def _sequentially(wave, *filters):
    data = wave.ys.clone()
    for f in filters:
        data = f(data)
    return data


def _three():
    from torchfx_amd.filter import HiButterworth, LoButterworth
    return LoButterworth(cutoff=4000, order=2), HiButterworth(cutoff=200, order=2), LoButterworth(cutoff=6000, order=2)


class TestDeferredFusion:
    @pytest.mark.parametrize("form", ["pipe", "filterchain", "sequential"])
    def test_fused_equals_sequential(self, form):
        wave = _wave()
        a = _three()
        if form == "pipe":
            result = wave | a[0] | a[1] | a[2]
        elif form == "filterchain":
            result = wave | (a[0] | a[1] | a[2])
        else:
            result = wave | nn.Sequential(*a)
        b = _three()
        _prepare(FS16, *b)
        torch.testing.assert_close(result.ys, _sequentially(wave, *b), atol=1e-6, rtol=1e-6)

    def test_mixed_chain_fuses_iir_runs_separately(self):
        from torchfx_amd import Gain
        from torchfx_amd.filter import HiButterworth, LoButterworth

        def mods():
            return (LoButterworth(cutoff=4000, order=2), HiButterworth(cutoff=200, order=2), Gain(0.5),
                    LoButterworth(cutoff=6000, order=2), HiButterworth(cutoff=100, order=2))
        wave = _wave()
        a = mods()
        result = wave | a[0] | a[1] | a[2] | a[3] | a[4]
        # two IIR runs, fused independently; the Gain between them rides on the first run's kernel (default plan)
        plan = result.plan()
        assert [type(m).__name__ for m in plan] == ["Epilogued", "FusedSOSCascade"]
        assert type(plan[0].producer).__name__ == "FusedSOSCascade" and plan[0].gain is a[2]
        b = mods()
        _prepare(FS16, *b)
        torch.testing.assert_close(result.ys, _sequentially(wave, *b), atol=1e-6, rtol=1e-6)

    def test_single_iir_no_fusion_and_no_mutation(self):
        from torchfx_amd import Gain
        from torchfx_amd.filter import LoButterworth
        wave = _wave()
        lone = LoButterworth(cutoff=4000, order=2)
        result = wave | lone | Gain(0.8)
        plan = result.plan()
        assert [type(m).__name__ for m in plan] == ["Epilogued"] and plan[0].producer is lone     # not wrapped in a FusedSOSCascade
        f1b, gb = LoButterworth(cutoff=4000, order=2), Gain(0.8)
        _prepare(FS16, f1b, gb)
        torch.testing.assert_close(result.ys, _sequentially(wave, f1b, gb), atol=1e-6, rtol=1e-6)
        f1 = LoButterworth(cutoff=4000, order=2, fs=FS16)
        f1.compute_coefficients()
        before = f1._sos.clone()
        _ = wave | f1
        torch.testing.assert_close(f1._sos, before)


class TestLazyAndChains:
    def test_pipeline_deferred_until_ys_access(self):
        from torchfx_amd.filter import LoButterworth
        w2 = _wave() | LoButterworth(cutoff=4000, order=2)
        assert len(w2._pipeline) == 1
        _ = w2.ys
        assert len(w2._pipeline) == 0
        w3 = _wave() | LoButterworth(cutoff=4000, order=2)
        new = dev(torch.randn(1, N16, dtype=torch.float64))
        w3.ys = new
        assert w3._pipeline == []
        torch.testing.assert_close(w3.ys, new)

    def test_filterchain_forms(self):
        from torchfx_amd import FilterChain, Gain
        from torchfx_amd.filter import BiquadLPF, HiButterworth, LoButterworth
        f1, f2, f3 = (LoButterworth(cutoff=4000, order=2, fs=FS16), HiButterworth(cutoff=200, order=2, fs=FS16),
                      LoButterworth(cutoff=6000, order=2, fs=FS16))
        chain = (f1 | f2) | f3
        assert isinstance(chain, FilterChain) and len(list(chain.children())) == 3
        mixed = Gain(0.5) | f1
        assert isinstance(mixed, FilterChain) and len(list(mixed.children())) == 2
        assert Gain(0.5).__or__(42) is NotImplemented
        assert (f1 | f2).__ror__("anything") is NotImplemented
        wave = _wave()
        result = wave | BiquadLPF(cutoff=3000, q=0.707) | LoButterworth(cutoff=4000, order=2)
        bq, lo = BiquadLPF(cutoff=3000, q=0.707), LoButterworth(cutoff=4000, order=2)
        _prepare(FS16, bq, lo)
        torch.testing.assert_close(result.ys, _sequentially(wave, bq, lo), atol=1e-6, rtol=1e-6)
