"""Gain / Normalize without a GPU: the oracle against the reference's golden
outputs (tests/golden/effects.npz, written by oracle/make_golden.py from the real reference), the
module layer on the oracle backend, the reference's error behaviour, and the planner's gain folding."""
import numpy as np
import pytest
import torch

import torchfx_amd as fx
from oracle import oracle as O
from torchfx_amd import effect as E
from torchfx_amd import filter as F

GAINS = {"amp": dict(gain=0.37, gain_type="amplitude"), "db": dict(gain=-4.5, gain_type="db"),
         "db0": dict(gain=0.0, gain_type="db"), "pow": dict(gain=2.5, gain_type="power"),
         "clamp": dict(gain=1.9, gain_type="amplitude", clamp=True)}
STRATS = ("peak", "rms", "percentile", "per_channel")


def test_oracle_effects_match_reference_bit_for_bit(golden):
    g = golden("effects")
    for tag, kw in GAINS.items():
        assert np.array_equal(O.gain(g["x"], **kw), g["gain_" + tag]), tag
    assert np.array_equal(O.gain(g["x64"], 3.0, "db"), g["gain64_db"])
    for s in STRATS:
        assert np.allclose(O.normalize(g["x"], 0.8, s, 97.0), g["norm_" + s], rtol=0, atol=1e-6), s
        assert np.allclose(O.normalize(g["x64"], 1.25, s, 97.0), g["norm64_" + s], rtol=0, atol=1e-14), s
        assert np.array_equal(O.normalize(g["zeros"], 0.8, s, 97.0), g["normz_" + s]), s
    assert np.allclose(O.normalize(g["x3"], 0.5, "per_channel"), g["norm3_per_channel"], atol=1e-6)


def _strategy(name):
    return {"peak": E.PeakNormalizationStrategy(), "rms": E.RMSNormalizationStrategy(),
            "percentile": E.PercentileNormalizationStrategy(97.0), "per_channel": E.PerChannelNormalizationStrategy()}[name]


def test_modules_on_oracle_backend(golden, oracle_backend):
    g = golden("effects")
    x, x64 = torch.from_numpy(g["x"]), torch.from_numpy(g["x64"])
    for tag, kw in GAINS.items():
        y = E.Gain(**kw)(x)
        assert y.dtype == x.dtype and np.array_equal(y.numpy(), g["gain_" + tag]), tag
    assert E.Gain(0.0, "db")(x) is x                                  # 0 dB returns the input itself
    assert np.array_equal(E.Gain(3.0, "db")(x64).numpy(), g["gain64_db"])
    for s in STRATS:
        y = E.Normalize(0.8, _strategy(s))(x)
        assert np.allclose(y.numpy(), g["norm_" + s], rtol=0, atol=1e-6), s
        z = E.Normalize(0.8, _strategy(s))(torch.from_numpy(g["zeros"]))
        assert np.array_equal(z.numpy(), g["normz_" + s]), s
    y3 = E.Normalize(0.5, E.PerChannelNormalizationStrategy())(torch.from_numpy(g["x3"]))
    assert y3.shape == (2, 3, 700) and np.allclose(y3.numpy(), g["norm3_per_channel"], atol=1e-6)
    # 1-D input, callable strategy, custom strategy object (tests/test_effects.py:73-143 of the reference)
    w = torch.tensor([0.2, -0.5, 0.4])
    assert torch.allclose(E.Normalize(peak=1.0)(w), w / 0.5)
    assert torch.equal(E.Normalize(5.0, strategy=lambda a, p: a + p)(w), w + 5.0)

    class Dummy(E.NormalizationStrategy):
        def __call__(self, waveform, peak):
            return waveform * 0 + peak

    assert torch.equal(E.Normalize(2.0, strategy=Dummy())(w), torch.full_like(w, 2.0))


def test_error_behaviour_matches_the_reference():
    with pytest.raises(ValueError, match="must be positive"):
        E.Gain(-1.0, "amplitude")
    with pytest.raises(ValueError):
        E.Gain(-1.0, "power")
    E.Gain(-6.0, "db")                                                # negative dB is fine
    with pytest.raises(AssertionError):
        E.Normalize(peak=0)
    with pytest.raises(TypeError, match="NormalizationStrategy"):
        E.Normalize(1.0, strategy="not_a_strategy")
    for bad in (0, 101):
        with pytest.raises(AssertionError):
            E.PercentileNormalizationStrategy(percentile=bad)
    with pytest.raises(AssertionError):
        E.PerChannelNormalizationStrategy()(torch.tensor([1.0, -2.0]), 1.0)
    with pytest.raises(ValueError, match=r"\(C, T\) or \(B, C, T\)"):
        E.PerChannelNormalizationStrategy()(torch.zeros(1, 2, 3, 4), 1.0)
    assert E.Gain(0.5).__or__(42) is NotImplemented
    assert isinstance(E.Gain(0.5) | F.LoButterworth(4000, order=2, fs=48000), fx.FilterChain)


def _mixed(fuse):
    g = [F.LoButterworth(4000, order=2), F.HiButterworth(200, order=2), E.Gain(0.5),
         F.LoButterworth(6000, order=2), F.HiButterworth(100, order=2)]
    return g, fuse


@pytest.mark.parametrize("fuse", ["reference", "epilogue", "coefficients"])
def test_mixed_pipeline_golden_and_plan(golden, oracle_backend, fuse):
    """wave | iir | iir | Gain | iir | iir (tests/test_chain_fusion.py:102-121 of the reference) against the
    reference's output: staged like the reference (two cascades around a gain pass), default (the gain rides on
    the first cascade's kernel as an epilogue: two launches), and with fuse_gain (folded into the coefficients:
    one cascade of four sections)."""
    g = golden("effects")
    mods, _ = _mixed(fuse)
    w = fx.Wave(torch.from_numpy(g["mix_x"]), 48000)
    assert w.fuse_epilogue is True and w.fuse_gain is False            # default policy
    w.fuse_gain, w.fuse_epilogue = fuse == "coefficients", fuse == "epilogue"
    for m in mods:
        w = w | m
    names = [type(m).__name__ for m in w.plan()]
    assert names == {"coefficients": ["FusedSOSCascade"], "reference": ["FusedSOSCascade", "Gain", "FusedSOSCascade"],
                     "epilogue": ["Epilogued", "FusedSOSCascade"]}[fuse]
    oracle_backend.calls.clear()
    y = w.ys.numpy()
    assert np.abs(y - g["mix_y"]).max() <= 1e-6
    launches = [c[0] for c in oracle_backend.calls]
    assert launches == {"coefficients": ["sos_forward"], "reference": ["sos_forward", "gain_forward", "sos_forward"],
                        "epilogue": ["sos_forward", "sos_forward"]}[fuse]
    if fuse == "coefficients":
        assert oracle_backend.calls[0][2] == 4                       # all four sections in one launch
    if fuse == "epilogue":
        assert oracle_backend.calls[0][-1] == "ep" and oracle_backend.calls[1][-1] != "ep"


def test_epilogue_plan_rules(oracle_backend):
    """Which `filter | Gain | Normalize` runs become one kernel with an epilogue (effect.Epilogued)."""
    w = fx.Wave(torch.zeros(2, 64), 48000)
    w.fuse_fir = w.fuse_spectral = False

    def names(wave):
        return [type(m).__name__ for m in wave.plan()]
    lo, hi = F.LoButterworth(4000, order=2), F.HiButterworth(100, order=2)
    assert names(w | lo | E.Gain(2.0, clamp=True)) == ["Epilogued"]                                   # lone stateful IIR + clamping gain
    assert names(w | lo | hi | E.Gain(0.5) | E.Normalize(0.9)) == ["Epilogued"]                       # cascade + gain + peak normalise
    assert names(w | lo | E.Normalize(0.9, E.RMSNormalizationStrategy())) == ["Epilogued"]
    assert names(w | lo | E.Normalize(0.9, E.PerChannelNormalizationStrategy())) == ["Epilogued"]
    assert names(w | lo | E.Normalize(0.9, E.PercentileNormalizationStrategy(99.0))) == ["LoButterworth", "Normalize"]   # selection, not a stream
    assert names(w | lo | E.Normalize(0.9, lambda x, p: x)) == ["LoButterworth", "Normalize"]
    assert names(w | F.FIR([0.5, 0.5]) | E.Gain(0.5)) == ["Epilogued"]                                # FFT-mode FIR
    assert names(w | F.FIR([0.5, 0.5], conv_mode="direct") | E.Gain(0.5)) == ["FIR", "Gain"]          # direct mode: no epilogue
    assert names(w | E.Gain(0.5) | lo) == ["Gain", "LoButterworth"]                                   # a gain BEFORE the filter is not an epilogue
    assert names(w | lo | E.Gain(0.5) | E.Gain(0.5)) == ["Epilogued", "Gain"]                         # one gain per epilogue (rounding order)
    w.fuse_epilogue = False
    assert names(w | lo | E.Gain(0.5)) == ["LoButterworth", "Gain"]
    # results: identical to the staged pipeline (bit for bit for gain + clamp)
    x = torch.randn(3, 4000, generator=torch.Generator().manual_seed(3))
    for tail in ([E.Gain(1.7, clamp=True)], [E.Gain(0.3), E.Normalize(0.8)], [E.Normalize(0.5, E.RMSNormalizationStrategy())],
                 [E.Gain(2.0, "db", clamp=True), E.Normalize(0.7, E.PerChannelNormalizationStrategy())]):
        outs = []
        for ep in (False, True):
            wave = fx.Wave(x, 48000)
            wave.fuse_epilogue, wave.fuse_fir, wave.fuse_spectral = ep, False, False
            wave = wave | F.LoButterworth(3000, order=4) | F.HiShelving(2000, q=0.7, gain=2.0)
            for t in tail:
                wave = wave | t
            outs.append(wave.ys)
        if not any(isinstance(t, E.Normalize) for t in tail):
            assert torch.equal(outs[0], outs[1])
        else:
            assert float((outs[0] - outs[1]).abs().max()) <= 1e-6 * max(1.0, float(outs[0].abs().max()))


def test_gain_folding_rules(oracle_backend):
    w = fx.Wave(torch.zeros(2, 64), 48000)
    w.fuse_gain, w.fuse_fir, w.fuse_spectral, w.fuse_epilogue = True, False, False, False
    lone = w | E.Gain(0.5) | F.LoButterworth(4000, order=2) | E.Gain(0.1)
    assert [type(m).__name__ for m in lone.plan()] == ["Gain", "LoButterworth", "Gain"]   # stateful lone IIR: staged
    clamp = w | F.LoButterworth(4000, order=2) | F.HiButterworth(100, order=2) | E.Gain(2.0, clamp=True)
    assert [type(m).__name__ for m in clamp.plan()] == ["FusedSOSCascade", "Gain"]        # clamp is not linear
    fir = w | E.Gain(2.0) | F.FIR([0.25, 0.5]) | E.Gain(3.0, "db")
    p = fir.plan()
    assert [type(m).__name__ for m in p] == ["FIR"]
    taps = p[0].kernel.reshape(-1).flip(0).double().numpy()
    assert np.allclose(taps, np.array([0.25, 0.5]) * 2.0 * 10 ** (3 / 20), rtol=1e-12)
    two = w | F.FIR([1.0, 1.0]) | E.Gain(0.5) | F.FIR([1.0, -1.0])                      # fold, but do not merge FIRs
    assert [type(m).__name__ for m in two.plan()] == ["FIR", "FIR"]
    tail = w | E.Gain(0.5)
    assert [type(m).__name__ for m in tail.plan()] == ["Gain"]


def test_reverb_is_the_delay_line(golden, oracle_backend):
    g = golden("delay")
    x = torch.from_numpy(g["x"])
    y = E.Reverb(delay=100, decay=0.5, mix=0.3)(x)
    assert np.abs(y.numpy() - g["y"]).max() <= 1e-7
    short = torch.zeros(2, 50)
    assert E.Reverb(delay=100)(short) is short                       # not longer than the delay: same tensor
    y3 = E.Reverb(delay=100, decay=0.5, mix=0.3)(x.reshape(1, 2, -1))
    assert y3.shape == (1, 2, 1000) and np.abs(y3.numpy()[0] - g["y"]).max() <= 1e-7
    for kw in (dict(delay=0), dict(decay=1.0), dict(decay=0.0), dict(mix=1.5)):
        with pytest.raises(AssertionError):
            E.Reverb(**kw)
