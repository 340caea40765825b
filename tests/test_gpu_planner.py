"""GPU parity -- modules and the Wave planner (rows a9-a12): chain fusion, spectral folding, custom ops.
Tolerances and helpers: tests/gpu_common.py."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.gpu_common import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


def test_wave_chain_golden(golden):
    from torchfx_amd import Wave
    from torchfx_amd import filter as F
    g = golden("chain")
    f1, f2 = F.HiButterworth(100, order=2), F.LoButterworth(8000, order=4)
    f3 = F.ParametricEQ(2000, 1.0, -3.0)
    fir = F.DesignableFIR(cutoff=6000, num_taps=127, fs=48000)
    w = Wave(g["x"], 48000, device=DEV) | f1 | f2 | fir | f3
    close(w.ys, g["y_chain"], TOL_CONV_F32, "wave chain")
    p1, p2 = F.LoButterworth(1000, order=2, fs=48000), F.HiButterworth(4000, order=2, fs=48000)
    close((p1 + p2)(dev(g["x"])), g["y_par"], 3e-7, "parallel sum")


def test_cfg5_small_chain_golden_staged_and_fused(golden):
    from scipy.signal import firwin
    from torchfx_amd import Wave
    from torchfx_amd import filter as F
    g = golden("chain")
    irs = np.random.default_rng(1).standard_normal(4097) * np.exp(-np.arange(4097) / 500.0)
    irs = irs / np.abs(irs).sum()

    def pipe(fuse, spectral=False):
        w = Wave(g["xc"], 48000, device=DEV)
        w.fuse_fir, w.fuse_spectral = fuse, spectral
        return (w | F.LoButterworth(2000, order=6) | F.ParametricEQ(frequency=1000, q=2.0, gain=3.0)
                | F.FIR(firwin(1024, 5000, fs=48000)) | F.FIR(irs))
    ws = pipe(False)
    assert [type(m).__name__ for m in ws.plan()] == ["FusedSOSCascade", "FIR", "FIR"]   # the reference's staging
    close(ws.ys, g["yc"], TOL_CONV_F32, "staged chain")
    wf = pipe(True)
    assert len(wf.plan()) == 2          # one SOS cascade + one merged FIR
    close(wf.ys, g["yc"], TOL_CONV_F32, "fused chain (merged FIR)")
    wd = (Wave(g["xc"], 48000, device=DEV) | F.LoButterworth(2000, order=6) | F.ParametricEQ(frequency=1000, q=2.0, gain=3.0)
          | F.FIR(firwin(1024, 5000, fs=48000)) | F.FIR(irs))
    # default policy: FIR run merged, the cascade in the reference's float64 arithmetic (rows this short: its own launch)
    assert (wd.fuse_fir, wd.fuse_spectral, wd.fuse_recursive) == (True, False, True)
    assert [type(m).__name__ for m in wd.plan()] == ["FusedSOSCascade", "FIR"]
    close(wd.ys, g["yc"], TOL_CONV_F32, "default plan")
    wo = Wave(g["xc"], 48000, device=DEV)
    wo.fuse_spectral = True                                      # opt-in: the whole LTI run as one overlap-save pass
    wo = (wo | F.LoButterworth(2000, order=6) | F.ParametricEQ(frequency=1000, q=2.0, gain=3.0)
          | F.FIR(firwin(1024, 5000, fs=48000)) | F.FIR(irs))
    assert [type(m).__name__ for m in wo.plan()] == ["FIR"]


@pytest.mark.parametrize("policy", ["auto", "auto_fold", "fir_only", "reference"])
def test_chain_with_iir_gain_golden(golden, policy, monkeypatch):
    """tests/golden/chain_gain.npz (reference staged output; IIR run with +12 dB shelf, +9 dB Q=4 peak, 80 Hz
    high-pass, then FIR-257 | FIR-2049) under the plans: default, one overlap-save pass for the whole chain, cascade
    kernel + merged FIR, and the reference's staging.  The merged 2305-tap FIR runs on the one-launch 8192-point kernel
    (9.6 B/sample), so by the planner's byte model the cascade stays its own 8 B/sample pass by default: folding its
    impulse response in would push the run past 4096 taps onto the three-pass pipeline (~26 B/sample).  Without the
    8192- and 16 384-point kernels (`auto_fold`) the fold pays and the whole chain is one pass."""
    from scipy.signal import firwin
    from torchfx_amd import Wave
    from torchfx_amd import filter as F
    g = golden("chain_gain")
    w = Wave(g["x"], 48000, device=DEV)
    if policy == "auto_fold":
        monkeypatch.setenv("TFX_OLS_LDS8K_MINK", "0")
        monkeypatch.setenv("TFX_OLS_LDS16K", "0")
        w.fuse_spectral = True
    elif policy != "auto":
        w.fuse_spectral = False
        w.fuse_fir = policy == "fir_only"
    irg = np.random.default_rng(2).standard_normal(2049) * np.exp(-np.arange(2049) / 300.0)
    w = (w | F.HiShelving(3000, q=0.7, gain=4.0) | F.ParametricEQ(frequency=500, q=4.0, gain=9.0)
         | F.HiButterworth(80, order=2) | F.FIR(firwin(257, 6000, fs=48000)) | F.FIR(8.0 * irg / np.abs(irg).sum()))
    assert len(w.plan()) == {"auto": 2, "auto_fold": 1, "fir_only": 2, "reference": 3}[policy]
    close(w.ys, g["y"], TOL_CONV_F32, f"chain with IIR gain, plan {policy}")


@pytest.mark.parametrize("case", ["hp20", "shelf40", "hp20_shelf40", "lone_stateful", "held_cascade"])
def test_spectral_fold_targeted_cases(case):
    """the default plan against the STAGED oracle on the filters where folding an IIR run into the
    FFT FIR behind it is least comfortable -- a 20 Hz high-pass (pole radius 0.998, 22 000-tap impulse response,
    exact DC null the float32 FFT has to reproduce), a +40 dB shelf (gain 100) -- and the two runs the planner must
    refuse: a lone IIR and a user-held FusedSOSCascade, both stateful across waves (chunked == one shot)."""
    from scipy.signal import firwin
    from torchfx_amd import Wave
    from torchfx_amd import filter as F
    x = rnd((3, 400_000), 77)
    x += 0.25                                                      # a DC offset for the high-pass to remove
    fir = F.FIR(firwin(513, 7000, fs=48000))
    kf = fir.kernel.numpy().reshape(-1)
    mk = {"hp20": lambda: [F.HiButterworth(20, order=2, fs=48000), F.LoButterworth(9000, order=2, fs=48000)],
          "shelf40": lambda: [F.LoShelving(200, q=0.7, gain=40.0, gain_scale="db", fs=48000), F.HiButterworth(300, order=2, fs=48000)],
          "hp20_shelf40": lambda: [F.HiButterworth(20, order=4, fs=48000), F.LoShelving(100, q=0.7, gain=40.0, gain_scale="db", fs=48000)],
          "lone_stateful": lambda: [F.HiButterworth(20, order=2, fs=48000)],
          "held_cascade": lambda: [F.FusedSOSCascade(F.HiButterworth(20, order=2, fs=48000), F.LoButterworth(9000, order=2, fs=48000))]}[case]
    members = mk()
    for m in members:
        if hasattr(m, "compute_coefficients") and getattr(m, "_sos", None) is None:
            m.compute_coefficients()
    sos = np.vstack([m._sos.numpy() for m in members])
    ref = O.chain_forward(x, sos, [kf])
    scale = max(1.0, float(np.abs(ref).max()))

    def pipe(xs):
        w = Wave(xs, 48000, device=DEV)
        w.fuse_spectral = True                                     # the fold is opt-in since round 5; its bound decides here
        for m in members:
            w = w | m
        return w | fir
    w = pipe(x)
    names = [type(m).__name__ for m in w.plan()]
    if case in ("lone_stateful", "held_cascade"):
        assert len(names) == 2 and names[1] == "FIR" and w.plan()[0] is members[0], names      # staged: the module itself runs
    else:
        assert names in (["FIR"], ["FusedSOSCascade", "FIR"]), names                                # folded only if it pays
    y = w.ys
    err = float(np.abs(y.cpu().numpy() - ref).max())
    assert err <= 1e-5 * scale, (case, names, err, scale)
    if case in ("lone_stateful", "held_cascade"):
        # the state is carried on the user's object from wave to wave: two chunks == one shot (IIR part exactly;
        # the stateless FIR is applied to the concatenated IIR output)
        members[0].reset_state()
        a = (Wave(x[:, :150_000], 48000, device=DEV) | members[0]).ys
        b = (Wave(x[:, 150_000:], 48000, device=DEV) | members[0]).ys
        yi = torch.cat([a, b], dim=1)
        close(fir(yi), ref, 1e-5, "chunked " + case)


def _raw_sos_class():
    from torchfx_amd import filter as F

    class RawSOS(F.IIR):
        """An IIR step around given SOS rows (the fixtures' ill-conditioned designs)."""

        def __init__(self, sos, fs=48000):
            super().__init__(fs)
            self._set_sos(sos)

        def compute_coefficients(self):
            pass
    return RawSOS


def _RawSOS(sos):
    return _raw_sos_class()(sos)


HARD = ["hicheby1_20", "hibutter_20_o5", "lobutter_40_o8", "ellip_o12", "notch_q30", "butter_o20"]


@pytest.mark.parametrize("taps", [513, 65536])
def test_fold_error_bound_on_the_hard_cascades(golden, taps):
    """every cascade of iir_hard.npz (pole radius up to 0.999, order up to 20) in front of FIR-513 and
    FIR-65536 through the plan WITH the spectral fold switched on: within 1e-5 of the staged oracle on the device.  Every
    fold that is taken carries its host-side error estimate (`wave._fold_error_estimate`, limit 2e-6; these cascades
    measure 2e-10 ... 3e-8: the float32 FFT's error is relative to the output scale, whatever the poles); the ones that
    are refused here are refused by the impulse-response length guard (notch, Q = 30: 387 028 samples of memory) or by
    the byte model (a 513-tap FIR does not pay for a 66 574-tap impulse response) and run staged, in the reference's
    float64 arithmetic.  `test_fold_bound_refuses` checks the bound's own refusal path."""
    from torchfx_amd import Wave, wave as W
    from torchfx_amd import filter as F
    g = golden("iir_hard")
    T = 400_000 if taps == 513 else 1_200_000
    x = rnd((2, T), 31)
    k = (np.random.default_rng(9).standard_normal(taps) * np.exp(-np.arange(taps) / (taps / 8.0)))
    k = (k / np.abs(k).sum()).astype(np.float32)
    fir = F.FIR(k)
    refused, folded = [], []
    for name in HARD:
        sos = g[name + "_sos"]
        half = max(1, sos.shape[0] // 2)
        if sos.shape[0] < 2:
            sos = np.vstack([sos, [[1.0, 0, 0, 1, 0, 0]]])
            half = 1
        casc = [_RawSOS(sos[:half]), _RawSOS(sos[half:])]         # two IIR steps -> the planner builds (and may fold) the cascade
        w = Wave(x, 48000, device=DEV)
        w.fuse_spectral = True
        w = w | casc[0] | casc[1] | fir
        plan = w.plan()
        names = [type(m).__name__ for m in plan]
        (folded if names == ["FIR"] else refused).append((name, names))
        if names == ["FIR"]:
            assert plan[0].fold_error_estimate <= W.FOLD_ERROR_LIMIT
        ref = O.chain_forward(x[:1], sos, [fir.kernel.numpy().reshape(-1)])
        scale = max(1.0, float(np.abs(ref).max()))
        err = float(np.abs(w.ys[:1].cpu().numpy() - ref).max())
        assert err <= 1e-5 * scale, (name, taps, names, err, scale)
    assert refused, f"the bound refused nothing at {taps} taps: {folded}"


def test_default_plan_runs_the_recursion_inside_the_overlap_save_pass():
    """`fuse_recursive` (default): planner-built cascade | long FFT FIR on float32 rows of a multiple of 32 samples =
    ONE CascadeFIR step (tfx_sos_fft_conv_forward); its section taps are the float64 recursion's, its result the staged
    chain's.  Rows the kernel does not serve keep the two launches."""
    from torchfx_amd import Wave
    from torchfx_amd import filter as F
    T = (1 << 20) + 320_000
    x = rnd((2, T), 3)
    f1, f2 = F.LoButterworth(2000, order=6, fs=48000), F.ParametricEQ(frequency=1000, q=2.0, gain=3.0, fs=48000)
    rev = F.FIR(reverb_ir(65536))
    fir = F.DesignableFIR(cutoff=5000, num_taps=1024, fs=48000)
    w = Wave(x, 48000, device=DEV) | f1 | f2 | fir | rev
    # at this length the pipeline takes the 2^18-point block: two launches
    assert [type(m).__name__ for m in w.plan()] == ["FusedSOSCascade", "FIR"]
    xl = rnd((1, 4_200_000 // 32 * 32), 4)
    wl = Wave(xl, 48000, device=DEV) | f1 | f2 | fir | rev
    plan = wl.plan()
    assert [type(m).__name__ for m in plan] == ["CascadeFIR"] and plan[0].fir.kernel.numel() == 1024 + 65536 - 1
    y = wl.ys
    sos = np.vstack([f1._sos.numpy(), f2._sos.numpy()])
    ref = O.chain_forward(xl, sos, [fir.kernel.numpy().reshape(-1), rev.kernel.numpy().reshape(-1)])
    close(y, ref, TOL_CONV_F32, "default plan, recursion inside pass A")
    y2, sec = plan[0](dev(xl), return_sections=True)                # one row = a handful of frame pairs: `.ys` ran the two staged launches,
    close(y2, y.cpu().numpy(), 2e-6, "fused pass vs the staged pair")  # the section taps come from the fused pass (same arithmetic)
    _, _, _, rs = O.sos_forward(xl.astype(np.float64), sos, sections=True)
    for s in range(4):
        close(sec[s], rs[s], TOL_IIR_F64OUT, f"section {s}")
    # odd length: the same single step since round 6 (rows shift their frame grid), and the plan says which route a tensor takes
    xo = xl[:, :-7].copy()
    wo = Wave(xo, 48000, device=DEV) | f1 | f2 | fir | rev
    assert [type(m).__name__ for m in wo.plan()] == ["CascadeFIR"]
    step_odd = wo.plan()[0]
    lines = wo.explain()
    assert len(lines) == 1 and lines[0].startswith("CascadeFIR: staged -- ") and "frame pairs <" in lines[0]
    close(wo.ys, ref[:, :-7], TOL_CONV_F32, "an odd length")
    y3, _ = step_odd(dev(xo), return_sections=True)
    close(y3, ref[:, :-7], TOL_CONV_F32, "an odd length, fused pass")


def test_gain_and_normalize_ride_on_the_cascade_fir_step():
    """`iir | iir | fir | Gain | Normalize` on long rows: ONE producing step (Epilogued around CascadeFIR) + the apply pass;
    equal to the same pipeline with every fusion off (the reference's staging) to the FIR tolerance."""
    import torchfx_amd as fx
    from torchfx_amd import effect as E
    from torchfx_amd import filter as F
    T = 4_200_000 // 32 * 32
    x = rnd((2, T), 21)
    f1, f2 = F.LoButterworth(2000, order=6, fs=48000), F.ParametricEQ(frequency=1000, q=2.0, gain=3.0, fs=48000)
    rev = F.FIR(reverb_ir(65536))

    def pipe(reference):
        w = fx.Wave(x, 48000, device=DEV)
        if reference:
            w.fuse_fir = w.fuse_spectral = w.fuse_recursive = w.fuse_epilogue = w.fuse_gain = False
        return w | f1 | f2 | rev | E.Gain(2.0, gain_type="amplitude", clamp=True) | E.Normalize(peak=0.5)
    wp = pipe(False)
    plan = wp.plan()
    assert [type(m).__name__ for m in plan] == ["Epilogued"] and type(plan[0].producer).__name__ == "CascadeFIR"
    plan[0].producer.MIN_PAIRS = 1                                   # two rows: make the fused pass run (small batches take the staged pair)
    wr = pipe(True)
    assert len(wr.plan()) == 4
    close(wp.ys, wr.ys.cpu().numpy(), TOL_CONV_F32, "epilogue on the fused step vs staged")


def test_module_shapes_dtype_and_state_rules(golden):
    from torchfx_amd import filter as F
    g = golden("iir_shapes")
    bq = F.BiquadLPF(cutoff=1500, q=0.9, fs=48000)
    y = bq(dev(g["x1d"]))
    assert y.shape == g["y1d"].shape and y.dtype == torch.float32
    close(y, g["y1d"], TOL_IIR_F32OUT, "1-D")
    close(bq._state_y, g["bq_sy"], TOL_STATE)
    lr = F.LoLinkwitzRiley(1200, order=4, fs=44100)
    y = lr(dev(g["x3d"]))
    assert y.dtype == torch.float64
    close(y, g["y3d"], TOL_IIR_F64OUT, "3-D")
    assert lr._state_x.shape == (2, 6, 2)
    close(lr._state_y, g["lr_sy"], TOL_STATE)
    lr(dev(g["x3d"][0]))                    # channel count changes: state silently re-zeroed
    assert lr._state_x.shape == (2, 3, 2)


def test_delay_and_passthrough(golden):
    g = golden("delay")
    close(ext().delay_line_forward(dev(g["x"]), 100, 0.5, 0.3), g["y"], 1e-7)
    x = dev(g["x"])
    assert ext().delay_line_forward(x, 5000, 0.5, 0.3) is x
    y, sx, sy = ext().sos_forward(x, None, torch.tensor([[1., 0, 0, 1, 0, 0]]).double(), None, None)
    assert torch.equal(y, x)                # pass-through section is exact (test_ops_dispatch.py:45-52)
    assert sx.shape == (1, 2, 2)


def test_custom_ops_on_device(golden):
    import torchfx_amd.ops  # noqa: F401
    g = golden("iir_cfg1")
    y, sx, sy = torch.ops.torchfx_hip.sos_forward(dev(g["x"]), torch.from_numpy(g["sos"]), None, None)
    close(y, g["y"], TOL_IIR_F32OUT, "custom op sos")
    f = golden("fir")
    close(torch.ops.torchfx_hip.fir_direct_forward(dev(f["x"]), torch.from_numpy(f["k32"])), f["direct32"], TOL_CONV_F32)
    close(torch.ops.torchfx_hip.fft_conv_forward(dev(f["x"]), torch.from_numpy(f["k32"]), 31, 0), f["fft32"], TOL_CONV_F32)
    e = golden("effects")
    x = dev(e["x"])
    assert np.array_equal(torch.ops.torchfx_hip.gain_forward(x, 1.9, True).cpu().numpy(), e["gain_clamp"])
    close(torch.ops.torchfx_hip.normalize_forward(x, 0.8, 0, True), e["norm_per_channel"], 1e-6)
    banks = torch.from_numpy(np.stack([g["sos"], g["sos"][::-1].copy()]))
    yb, _, _ = torch.ops.torchfx_hip.sos_bank_forward(dev(g["x"]), banks, None, None)
    ys, _, _ = torch.ops.torchfx_hip.sos_bank_sum_forward(dev(g["x"]), banks, None, None)
    assert yb.shape == (2, *g["x"].shape) and torch.equal(ys, yb[0] + yb[1])
    m = torch.ops.torchfx_hip.sos_bank_sum_forward(torch.empty(3, 50, device="meta"), banks, None, None)
    assert m[0].shape == (3, 50) and m[1].shape == (banks.shape[1], 6, 2)


def test_spectral_fusion_on_device(golden):
    """Opt-in: the whole LTI chain as one overlap-save pass == staged reference output."""
    from scipy.signal import firwin
    from torchfx_amd import Wave
    from torchfx_amd import filter as F
    g = golden("chain")
    irs = np.random.default_rng(1).standard_normal(4097) * np.exp(-np.arange(4097) / 500.0)
    irs = irs / np.abs(irs).sum()
    w = Wave(g["xc"], 48000, device=DEV)
    w.fuse_fir = w.fuse_spectral = True
    w = (w | F.LoButterworth(2000, order=6) | F.ParametricEQ(frequency=1000, q=2.0, gain=3.0)
         | F.FIR(firwin(1024, 5000, fs=48000)) | F.FIR(irs))
    assert len(w.plan()) == 1
    close(w.ys, g["yc"], TOL_CONV_F32, "spectral chain")


@pytest.mark.parametrize("fuse", [False, True])
def test_wave_pipeline_with_gain_on_device(golden, fuse):
    import torchfx_amd as fx
    from torchfx_amd import effect as E
    from torchfx_amd import filter as F
    g = golden("effects")
    w = fx.Wave(dev(g["mix_x"]), 48000, device=DEV)
    w.fuse_gain, w.fuse_epilogue = fuse, False
    for m in (F.LoButterworth(4000, order=2), F.HiButterworth(200, order=2), E.Gain(0.5),
              F.LoButterworth(6000, order=2), F.HiButterworth(100, order=2)):
        w = w | m
    assert len(w.plan()) == (1 if fuse else 3)
    close(w.ys, g["mix_y"], 2e-7, "iir | iir | gain | iir | iir")


def test_planned_chain_steps_have_no_periodic_host_stall():
    """Regression: the planner's merged FIR taps are a float64 host buffer; converting them per call (a fresh 276 KB
    host allocation per step) made the driver hold the GPU queues for ~70 ms on every third synchronised step.  The
    host copy is cached per buffer now: 30 synchronised steps must all take about the same time."""
    import time
    import bench
    x = dev(rnd((16, 1_500_000), 3))
    plan, names = bench.plan_chain(x, fuse_fir=True, fuse_spectral=True)       # the plan with the longest merged, float64 tap buffer
    assert "68977 taps" in names
    for _ in range(3):
        bench.run_plan(plan, x)
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        t0 = time.perf_counter()
        y = bench.run_plan(plan, x)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
        del y
    ts = np.array(ts) * 1e3
    assert ts.max() < 10 * np.median(ts) + 5.0, f"step times (ms): {np.round(ts, 2).tolist()}"


@pytest.mark.parametrize("seed", range(64))
def test_random_pipelines_planned_equals_staged_on_device(seed):
    """Planner + kernels together: random pipelines of IIR / Biquad / FIR / Gain / Normalize / `+` steps under random
    fusion flags, planned execution on the device == the same modules applied one after the other on the device with
    every fusion off (the reference's definition of a pipeline) == the oracle applied step by step on the host."""
    import random
    import torchfx_amd as fx
    from torchfx_amd import effect as E
    from torchfx_amd import filter as F
    rnd_ = random.Random(seed)
    FS_ = 48000

    def make(spec):
        kind, a, b = spec
        if kind == "lo":
            return F.LoButterworth(a, order=b, fs=FS_)
        if kind == "hi":
            return F.HiButterworth(a, order=b, fs=FS_)
        if kind == "peq":
            return F.ParametricEQ(frequency=a, q=1.5, gain=b, fs=FS_)
        if kind == "bq":
            return F.BiquadLPF(cutoff=a, q=b, fs=FS_)
        if kind == "fir":
            return F.FIR((np.random.default_rng(b).standard_normal(a) / np.sqrt(a)).tolist())
        if kind == "par":
            return F.LoButterworth(a, order=2, fs=FS_) + F.HiButterworth(b, order=2, fs=FS_)
        if kind == "gain":
            return E.Gain(a, clamp=b)
        return E.Normalize(peak=a)

    def one():
        k = rnd_.choice(["lo", "hi", "peq", "bq", "fir", "fir", "par", "gain", "norm"])
        return {"lo": lambda: ("lo", rnd_.randint(200, 8000), rnd_.randint(1, 4)),
                "hi": lambda: ("hi", rnd_.randint(50, 2000), rnd_.randint(1, 3)),
                "peq": lambda: ("peq", rnd_.randint(100, 10000), rnd_.uniform(-6, 6)),
                "bq": lambda: ("bq", rnd_.randint(200, 8000), rnd_.uniform(0.3, 4.0)),
                "fir": lambda: ("fir", rnd_.choice([2, 17, 64, 300, 1500, 5000]), rnd_.randint(0, 1000)),
                "par": lambda: ("par", rnd_.randint(1000, 6000), rnd_.randint(60, 900)),
                "gain": lambda: ("gain", rnd_.uniform(0.2, 1.8), rnd_.random() < 0.3),
                "norm": lambda: ("norm", rnd_.uniform(0.3, 1.0), None)}[k]()

    specs = [one() for _ in range(rnd_.randint(1, 6))]
    C, T = rnd_.choice([(1, 30011), (3, 70000), (2, 200000)])
    x = rnd((C, T), 900 + seed)
    # staged on the host through the oracle-backed module implementations (tests/_fake_backend.py)
    from tests import _fake_backend as FB
    from torchfx_amd import torchfx_ext as TE
    names = ("sos_forward", "sos_bank_forward", "sos_bank_sum_forward", "biquad_forward", "fir_direct_forward", "fft_conv_forward",
             "fir_stream_forward", "normalize_apply", "sum_forward", "delay_line_forward", "gain_forward", "stat_forward", "normalize_forward")
    saved = {n: getattr(TE, n) for n in names}
    try:
        for n in names:
            setattr(TE, n, getattr(FB, n))
        ref = torch.from_numpy(x)
        for sp in specs:
            ref = make(sp)(ref)
    finally:
        for n in names:
            setattr(TE, n, saved[n])
    ref = ref.numpy()
    # staged on the device, every fusion off
    cur = dev(x)
    for sp in specs:
        cur = make(sp)(cur)
    scale = max(1.0, float(np.abs(ref).max()))
    assert float(np.abs(cur.cpu().numpy() - ref).max()) <= 1e-5 * scale, specs
    # planned, random flags
    w = fx.Wave(dev(x), FS_, device=DEV)
    w.fuse_fir, w.fuse_gain, w.fuse_spectral, w.fuse_epilogue = (rnd_.random() < 0.5 for _ in range(4))
    flags = (w.fuse_fir, w.fuse_gain, w.fuse_spectral, w.fuse_epilogue)
    for sp in specs:
        w = w | make(sp)
    y = w.ys
    assert y.shape == cur.shape and y.dtype == cur.dtype
    assert float((y - cur).abs().max()) <= 1e-5 * scale, (specs, flags, [type(m).__name__ for m in w.plan()] if hasattr(w, "plan") else None)
