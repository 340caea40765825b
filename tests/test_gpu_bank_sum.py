"""GPU parity -- filter bank and `+` of IIR branches in one launch (8f rank 2, row a12).
Tolerances and helpers: tests/gpu_common.py."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.gpu_common import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("C,T,NB,K", [(2, 3000, 5, 1), (3, 70001, 4, 2), (64, 100000, 8, 1), (1, 17, 3, 1)])
def test_filter_bank_vs_oracle(C, T, NB, K, sos_variant):
    from scipy.signal import butter
    rng = np.random.default_rng(NB * 100 + K)
    banks = np.stack([np.vstack([butter(2, [f, min(0.95, f * 1.5)], "bandpass", output="sos")[:K]])
                      for f in rng.uniform(0.01, 0.5, NB)])
    x = rnd((C, T), T + NB)
    sx0, sy0 = rng.standard_normal((K, NB * C, 2)), rng.standard_normal((K, NB * C, 2))
    y, sx, sy = ext().sos_bank_forward(dev(x), banks, dev(sx0), dev(sy0))
    assert y.shape == (NB, C, T)
    for b in range(NB):
        ey, esx, esy = O.sos_forward(x, banks[b], sx0[:, b * C:(b + 1) * C], sy0[:, b * C:(b + 1) * C])
        close(y[b], ey.astype(np.float32), 2.5e-7, f"band {b}")
        close(sx[:, b * C:(b + 1) * C], esx, TOL_STATE, f"band {b} sx")
        close(sy[:, b * C:(b + 1) * C], esy, TOL_STATE, f"band {b} sy")


def test_log_filter_bank_module_on_device():
    from torchfx_amd import filter as F
    fb = F.LogFilterBank(6, f_min=40, f_max=12000, q=1.414, fs=48000)
    x = rnd((2, 50000), 9)
    y = fb(dev(x))
    assert y.shape == (6, 2, 50000)
    for i, f in enumerate(fb.filters):
        e, _, _ = O.iir_module_forward(x, f._sos.numpy())
        close(y[i], e, TOL_IIR_F32OUT, f"band {i}")
    y2 = torch.cat([fb(dev(x[:, :20000])), fb(dev(x[:, 20000:]))], dim=-1)    # state carried per band
    fb.filters[0].reset_state()


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("n,shape", [(1, (3, 1001)), (2, (1, 5)), (5, (4, 100003)), (16, (2, 4096)), (17, (2, 5000)), (33, (1, 777))])
def test_branch_sum_bit_exact(n, shape, dtype):
    """`+` accumulates zeros_like + in-place adds in branch order (__base.py:1022-1026): the one-pass
    kernel (groups of <= 16 inputs) must give exactly that, also for odd sizes and unaligned views."""
    g = torch.Generator().manual_seed(n * 131 + shape[1])
    ts = [torch.randn(shape, generator=g, dtype=dtype) for _ in range(n)]
    exp = torch.zeros(shape, dtype=dtype)
    for t in ts:
        exp += t
    got = ext().sum_forward([t.to(DEV) for t in ts])
    assert torch.equal(got.cpu(), exp)
    if shape[1] > 16:                                     # views starting one element in: scalar path
        wide = [torch.randn((shape[0], shape[1] + 1), generator=g, dtype=dtype) for _ in range(n)]
        exp = torch.zeros(shape, dtype=dtype)
        for t in wide:
            exp += t[:, 1:]
        got = ext().sum_forward([t.to(DEV)[:, 1:] for t in wide])
        assert torch.equal(got.cpu(), exp)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("C,T,NB,K", [(1, 1, 2, 1), (2, 1000, 2, 2), (3, 70001, 3, 3), (5, 200000, 5, 1), (2, 4099, 4, 4)])
def test_branch_sum_in_one_launch_vs_oracle(C, T, NB, K, dtype):
    """`f1 + f2 + ...` of IIR branches as ONE launch (sum mode of the cascade kernel): equals the
    branch-by-branch oracle accumulated in the signal dtype in branch order, states included, and a
    chunked run with carried state equals the contiguous one."""
    from scipy.signal import butter
    rng = np.random.default_rng(C * 7 + T + NB)
    banks = np.stack([np.vstack([butter(2, rng.uniform(0.02, 0.8), btype=rng.choice(["low", "high"]), output="sos")
                                 for _ in range(K)]) for _ in range(NB)])
    x = rnd((C, T), 3 * T + NB, dtype)
    y, sx, sy = ext().sos_bank_sum_forward(dev(x), banks, None, None)
    exp = np.zeros_like(x)
    esy = []
    for b in range(NB):
        eb, _, sb = O.sos_forward(x, banks[b])
        exp += eb.astype(dtype)
        esy.append(sb)
    tol = 3e-7 if dtype == np.float32 else 1e-13
    close(y, exp, tol, "sum of branches")
    close(sy, np.concatenate(esy, axis=1), TOL_STATE, "branch states (band-major rows)")
    if T > 10:
        cut = T // 3 + 1
        y1, s1x, s1y = ext().sos_bank_sum_forward(dev(x[:, :cut].copy()), banks, None, None)
        y2, _, s2y = ext().sos_bank_sum_forward(dev(x[:, cut:].copy()), banks, s1x, s1y)
        close(torch.cat([y1, y2], dim=1), y.cpu().numpy(), tol, "chunked == contiguous")
        close(s2y, sy.cpu().numpy(), TOL_STATE)


def test_parallel_combination_runs_as_one_launch_and_keeps_branch_state():
    import torchfx_amd as fx
    from torchfx_amd import filter as F
    e = ext()
    lib = __import__("torchfx_amd._lib", fromlist=["load"]).load()
    lo, hi, pk = F.LoButterworth(800, order=4, fs=48000), F.HiButterworth(3000, order=2, fs=48000), \
        F.ParametricEQ(frequency=1000, q=2.0, gain=4.0, fs=48000)
    comb = lo + hi + pk
    x = dev(rnd((3, 30000), 12))
    lib.tfx_prof_enable(1)
    lib.tfx_prof_collect()
    y = comb(x)
    prof = __import__("json").loads(lib.tfx_prof_collect().decode())
    lib.tfx_prof_enable(0)
    assert sum(v["calls"] for v in prof.values()) == 1, prof               # one kernel launch in total
    ref = [F.LoButterworth(800, order=4, fs=48000), F.HiButterworth(3000, order=2, fs=48000),
           F.ParametricEQ(frequency=1000, q=2.0, gain=4.0, fs=48000)]
    exp = e.sum_forward([f(x) for f in ref])
    close(y, exp.cpu().numpy(), 3e-7, "combination == staged branches")
    for f, r in zip((lo, hi, pk), ref):                                     # every branch kept its own state
        assert f._state_y.shape == r._state_y.shape
        close(f._state_y, r._state_y.cpu().numpy(), TOL_STATE)
    y2 = comb(x)                                                            # second call continues from it
    exp2 = e.sum_forward([f(x) for f in ref])
    close(y2, exp2.cpu().numpy(), 3e-7, "stateful second call")
