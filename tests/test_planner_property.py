"""Property test of the Wave planner (host logic, oracle backend): for random pipelines of IIR /
Biquad / FIR / Gain steps and any combination of the opt-in fusions, the planned execution equals
the step-by-step execution of fresh copies of the same modules (the reference's definition of a
pipeline, tests/test_chain_fusion.py:60-99) to float32 round-off, and never issues more launches."""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

import torchfx_amd as fx
from torchfx_amd import effect as E
from torchfx_amd import filter as F

FS = 48000

step = st.one_of(
    st.tuples(st.just("lo"), st.integers(200, 8000), st.integers(1, 4)),
    st.tuples(st.just("hi"), st.integers(50, 2000), st.integers(1, 3)),
    st.tuples(st.just("peq"), st.integers(100, 10000), st.floats(-6, 6)),
    st.tuples(st.just("bq"), st.integers(200, 8000), st.floats(0.3, 4.0)),
    st.tuples(st.just("fir"), st.integers(2, 40), st.integers(0, 1000)),
    st.tuples(st.just("gain"), st.floats(0.1, 2.0), st.booleans()),
    st.tuples(st.just("gaindb"), st.floats(-12, 12), st.just(False)),
)


def build(spec):
    kind, a, b = spec
    if kind == "lo":
        return F.LoButterworth(a, order=b, fs=FS)
    if kind == "hi":
        return F.HiButterworth(a, order=b, fs=FS)
    if kind == "peq":
        return F.ParametricEQ(frequency=a, q=1.5, gain=b, fs=FS)
    if kind == "bq":
        return F.BiquadLPF(cutoff=a, q=b, fs=FS)
    if kind == "fir":
        taps = np.random.default_rng(b).standard_normal(a) / a
        return F.FIR(taps.tolist())
    if kind == "gain":
        return E.Gain(a, clamp=b)
    return E.Gain(a, "db")


@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(specs=st.lists(step, min_size=1, max_size=7), fuse_fir=st.booleans(), fuse_gain=st.booleans(),
       fuse_spectral=st.booleans(), fuse_epilogue=st.booleans(), seed=st.integers(0, 1000))
def test_planned_pipeline_equals_stepwise(oracle_backend, specs, fuse_fir, fuse_gain, fuse_spectral, fuse_epilogue, seed):
    x = torch.from_numpy(np.random.default_rng(seed).standard_normal((2, 600)).astype(np.float32))
    # reference semantics: every module applied in order, fresh state
    cur = x
    staged_calls = 0
    for s in specs:
        oracle_backend.calls.clear()
        m = build(s)
        if hasattr(m, "compute_coefficients") and getattr(m, "_sos", 1) is None:
            m.compute_coefficients()
        cur = m(cur)
        staged_calls += len(oracle_backend.calls)
    w = fx.Wave(x, FS)
    w.fuse_fir, w.fuse_gain, w.fuse_spectral, w.fuse_epilogue = fuse_fir, fuse_gain, fuse_spectral, fuse_epilogue
    for s in specs:
        w = w | build(s)
    plan = w.plan()
    oracle_backend.calls.clear()
    y = w.ys
    scale = max(1.0, float(cur.abs().max()))
    assert y.shape == cur.shape and y.dtype == cur.dtype
    assert float((y - cur).abs().max()) <= 3e-6 * scale, [type(m).__name__ for m in plan]
    assert len(oracle_backend.calls) <= staged_calls
    if not fuse_fir and not fuse_gain and not fuse_spectral and not fuse_epilogue:
        # default plan: only IIR runs are fused, exactly like the reference's _materialize
        kinds = ["i" if s[0] in ("lo", "hi", "peq", "bq") else "o" for s in specs]
        runs = sum(1 for i, k in enumerate(kinds) if k == "i" and (i == 0 or kinds[i - 1] != "i"))
        assert len(plan) == kinds.count("o") + runs
