#!/usr/bin/env python3
"""oracle/make_golden.py -- generate tests/golden/*.npz FROM THE REAL REFERENCE.

Runs ONLY in the build container (needs /root/reference + oracle/_ref, built by
``make -C oracle ref``).  It

  1. imports the reference Python package from /root/reference/src with its own
     CPU extension (oracle/_ref/torchfx_ext.so, compiled from the reference's
     sources where they lie) and a stub ``soundfile`` module (the reference
     imports it unconditionally at realtime/stream.py:28; file I/O is not on
     the hot path);
  2. runs the reference on seeded inputs and stores inputs + expected outputs as
     small ``.npz`` fixtures (data only -- no reference source or bytecode);
  3. checks our CPU restatement (oracle/oracle.py) against every vector and
     fails loudly if it deviates (IIR: 1e-12 in f64 / bit-exact after the f32
     cast up to 1 ulp; FIR/FFT: 2e-5).

Nothing here runs on the GPU box; there the committed fixtures pin the oracle.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF_SRC = "/root/reference/src"
OUT = os.path.join(ROOT, "tests", "golden")


def import_reference():
    sys.modules.setdefault("soundfile", types.ModuleType("soundfile"))
    import torch  # noqa: F401  (must be loaded before the extension)

    so = os.path.join(HERE, "_ref", "torchfx_ext.so")
    spec = importlib.util.spec_from_file_location("torchfx.torchfx_ext", so)
    ext = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ext)
    sys.modules["torchfx.torchfx_ext"] = ext
    sys.path.insert(0, REF_SRC)
    import torchfx  # noqa: F401

    return torchfx


def main() -> None:
    import torch

    torchfx = import_reference()
    from torchfx import Wave, _ops
    from torchfx import filter as F
    from torchfx.filter._fftconv import fft_conv1d

    sys.path.insert(0, ROOT)
    from oracle import oracle as O

    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)

    def rnd(shape, seed, dtype=torch.float32):
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(*shape, generator=g, dtype=torch.float64)
        x = x / x.abs().max()
        return x.to(dtype)

    def sos_of(f):
        if f._sos is None:
            f.compute_coefficients()
        return f._sos.detach().cpu().numpy().astype(np.float64)

    report = []

    def check(name, got, exp, tol):
        err = float(np.max(np.abs(np.asarray(got, dtype=np.float64) - np.asarray(exp, dtype=np.float64)))) if np.size(exp) else 0.0
        tol = tol * max(1.0, float(np.max(np.abs(exp))) if np.size(exp) else 1.0)
        report.append((name, err, tol))
        assert err <= tol, f"oracle deviates from reference on {name}: {err} > {tol}"

    # ------------------------------------------------------------------ cfg 1
    x = rnd((1, 48000), 0)
    f = F.LoButterworth(1000, order=4, fs=48000)
    y = f(x)
    sos = sos_of(f)
    np.savez(os.path.join(OUT, "iir_cfg1.npz"), x=x.numpy(), sos=sos, y=y.numpy(),
             state_x=f._state_x.numpy(), state_y=f._state_y.numpy())
    oy, osx, osy = O.iir_module_forward(x.numpy(), sos)
    check("cfg1.y", oy, y.numpy(), 1e-7)
    check("cfg1.sx", osx, f._state_x.numpy(), 4e-12)
    check("cfg1.sy", osy, f._state_y.numpy(), 4e-12)

    # ----------------------------------------------- cfg 2 chain, per section
    x = rnd((4, 8192), 1)
    f1 = F.LoButterworth(2000, order=6, fs=48000)
    f2 = F.ParametricEQ(frequency=1000, q=2.0, gain=3.0, fs=48000)
    w = Wave(x, 48000) | f1 | f2
    y_fused = w.ys.numpy()
    sos = np.vstack([sos_of(f1), sos_of(f2)])
    # section by section through the public op (f64 intermediates)
    cur = x.to(torch.float64)
    secs = []
    for k in range(sos.shape[0]):
        cur, _, _ = _ops.parallel_iir_forward(cur, torch.from_numpy(sos[k:k + 1]), None, None)
        secs.append(cur.numpy().copy())
    secs = np.stack(secs)
    yy, sx, sy = _ops.parallel_iir_forward(x, torch.from_numpy(sos), None, None)
    assert np.array_equal(yy.numpy(), secs[-1])
    np.savez(os.path.join(OUT, "iir_cfg2_sections.npz"), x=x.numpy(), sos=sos,
             y=y_fused, y_sections=secs, state_x=sx.numpy(), state_y=sy.numpy())
    oy, osx, osy, osec = O.sos_forward(x.numpy(), sos, sections=True)
    check("cfg2.sections", osec, secs, 4e-12)
    check("cfg2.y", oy.astype(np.float32), y_fused, 1e-7)
    check("cfg2.sx", osx, sx.numpy(), 4e-12)
    check("cfg2.sy", osy, sy.numpy(), 4e-12)

    # ------------------------------------------- chunked == contiguous, state
    x = rnd((2, 2048), 2, torch.float64)
    fa = F.HiButterworth(300, order=3, fs=44100)
    fb = F.LoChebyshev1(4000, order=4, ripple=0.5, fs=44100)
    fz = F.FusedSOSCascade(fa, fb)
    c1 = fz(x[:, :1024]).numpy().copy()
    s1x, s1y = fz._state_x.numpy().copy(), fz._state_y.numpy().copy()
    c2 = fz(x[:, 1024:]).numpy().copy()
    sos = fz._sos.numpy()
    np.savez(os.path.join(OUT, "iir_chunked.npz"), x=x.numpy(), sos=sos, y1=c1, y2=c2,
             mid_state_x=s1x, mid_state_y=s1y, state_x=fz._state_x.numpy(), state_y=fz._state_y.numpy())
    o1, ox, oy_ = O.sos_forward(x.numpy()[:, :1024], sos)
    check("chunk.y1", o1, c1, 4e-12)
    check("chunk.midx", ox, s1x, 4e-12)
    o2, ox2, oy2 = O.sos_forward(x.numpy()[:, 1024:], sos, ox, oy_)
    check("chunk.y2", o2, c2, 4e-12)
    check("chunk.sx", ox2, fz._state_x.numpy(), 4e-12)
    check("chunk.sy", oy2, fz._state_y.numpy(), 4e-12)

    # --------------------------------------------- ill-conditioned / long memory
    x = rnd((2, 16384), 3)
    hard = {
        "hicheby1_20": F.HiChebyshev1(20, order=4, ripple=0.1, fs=48000),
        "hibutter_20_o5": F.HiButterworth(20, order=5, fs=48000),
        "lobutter_40_o8": F.LoButterworth(40, order=8, fs=48000),
        "ellip_o12": F.LoElliptic(3000, order=12, fs=44100),
        "notch_q30": F.Notch(60, q=30, fs=48000),
        "butter_o20": F.LoButterworth(5000, order=20, fs=44100),
    }
    d = {"x": x.numpy()}
    for k, flt in hard.items():
        y = flt(x).numpy()
        d[k + "_sos"] = sos_of(flt)
        d[k + "_y"] = y
        d[k + "_sx"] = flt._state_x.numpy()
        d[k + "_sy"] = flt._state_y.numpy()
        oy, osx, osy = O.iir_module_forward(x.numpy(), d[k + "_sos"])
        check(f"hard.{k}.y", oy, y, 1e-6 * max(1.0, float(np.abs(y).max())))
        check(f"hard.{k}.sy", osy, d[k + "_sy"], 1e-9 * max(1.0, float(np.abs(d[k + '_sy']).max())))
    np.savez(os.path.join(OUT, "iir_hard.npz"), **d)

    # ---------------------------------------------------- shapes / dtypes / K=1
    d = {}
    bq = F.BiquadLPF(cutoff=1500, q=0.9, fs=48000)
    x1 = rnd((3000,), 4)
    d["x1d"], d["y1d"] = x1.numpy(), bq(x1).numpy()
    d["bq_sos"] = sos_of(bq)
    d["bq_sx"], d["bq_sy"] = bq._state_x.numpy(), bq._state_y.numpy()
    lr = F.LoLinkwitzRiley(1200, order=4, fs=44100)
    x3 = rnd((2, 3, 1000), 5, torch.float64)
    d["x3d"], d["y3d"] = x3.numpy(), lr(x3).numpy()
    d["lr_sos"] = sos_of(lr)
    d["lr_sx"], d["lr_sy"] = lr._state_x.numpy(), lr._state_y.numpy()
    # direct ext call with non-zero, non-consistent initial states
    g = torch.Generator().manual_seed(6)
    xs = rnd((3, 500), 7, torch.float64)
    sosr = sos_of(F.LoButterworth(3000, order=6, fs=48000))
    isx = torch.randn(3, 3, 2, generator=g, dtype=torch.float64)
    isy = torch.randn(3, 3, 2, generator=g, dtype=torch.float64)
    ys, nsx, nsy = _ops.parallel_iir_forward(xs, torch.from_numpy(sosr), isx, isy)
    d.update(xs=xs.numpy(), s_sos=sosr, isx=isx.numpy(), isy=isy.numpy(), ys=ys.numpy(),
             nsx=nsx.numpy(), nsy=nsy.numpy())
    # T = 1 and T = 2 edge cases with state
    for Tn in (1, 2, 3):
        yt, tx, ty = _ops.parallel_iir_forward(xs[:, :Tn], torch.from_numpy(sosr), isx, isy)
        d[f"t{Tn}_y"], d[f"t{Tn}_sx"], d[f"t{Tn}_sy"] = yt.numpy(), tx.numpy(), ty.numpy()
        oy, ox, oyy = O.sos_forward(xs.numpy()[:, :Tn], sosr, isx.numpy(), isy.numpy())
        check(f"edge.T{Tn}.y", oy, yt.numpy(), 4e-12)
        check(f"edge.T{Tn}.sx", ox, tx.numpy(), 4e-12)
        check(f"edge.T{Tn}.sy", oyy, ty.numpy(), 4e-12)
    np.savez(os.path.join(OUT, "iir_shapes.npz"), **d)
    oy, ox, oyy = O.sos_forward(xs.numpy(), sosr, isx.numpy(), isy.numpy())
    check("state.y", oy, ys.numpy(), 4e-12)
    check("state.sx", ox, nsx.numpy(), 4e-12)
    check("state.sy", oyy, nsy.numpy(), 4e-12)
    oy, _, _ = O.iir_module_forward(x1.numpy()[None], d["bq_sos"])
    check("shape.1d", oy[0], d["y1d"], 1e-7)
    oy, _, _ = O.iir_module_forward(x3.numpy().reshape(6, 1000), d["lr_sos"])
    check("shape.3d", oy.reshape(2, 3, 1000), d["y3d"], 4e-12)

    # --------------------------------------------------------- designs (all classes)
    specs = [
        ("Butterworth", ("bandpass" if False else "lowpass", 1000), dict(order=4)),
        ("Butterworth", ("highpass", 500), dict(order=3, order_scale="linear")),
        ("Butterworth", ("lowpass", 2000), dict(order=24, order_scale="db")),
        ("HiButterworth", (800,), dict()),
        ("LoButterworth", (800,), dict()),
        ("LoButterworth", (2000,), dict(order=6)),
        ("Chebyshev1", ("lowpass", 1500), dict(order=5, ripple=0.5)),
        ("HiChebyshev1", (200,), dict(order=4, ripple=0.1)),
        ("LoChebyshev1", (3000,), dict(order=6, ripple=1.0)),
        ("Chebyshev2", ("highpass", 700), dict(order=4, ripple=30)),
        ("HiChebyshev2", (400,), dict(order=3, ripple=40)),
        ("LoChebyshev2", (5000,), dict(order=8, ripple=50)),
        ("Elliptic", ("lowpass", 2500), dict(order=5)),
        ("HiElliptic", (300,), dict(order=4, passband_ripple=0.5, stopband_attenuation=60)),
        ("LoElliptic", (6000,), dict(order=7)),
        ("LinkwitzRiley", ("lowpass", 1000), dict(order=4)),
        ("HiLinkwitzRiley", (2000,), dict(order=8)),
        ("LoLinkwitzRiley", (2000,), dict(order=2)),
        ("HiShelving", (4000, 0.707, 2.0), dict()),
        ("HiShelving", (4000, 0.707, 6.0), dict(gain_scale="db")),
        ("LoShelving", (200, 0.707, 0.5), dict()),
        ("LoShelving", (200, 1.0, -4.0), dict(gain_scale="db")),
        ("ParametricEQ", (1000, 2.0, 3.0), dict()),
        ("ParametricEQ", (250, 0.7, -6.0), dict()),
        ("Peaking", (3000, 1.5, 2.0, "linear"), dict()),
        ("Peaking", (3000, 1.5, 4.0, "db"), dict()),
        ("Notch", (60, 10.0), dict()),
        ("AllPass", (1000, 0.5), dict()),
        ("BiquadLPF", (1000, 0.707), dict()),
        ("BiquadHPF", (1000, 0.707), dict()),
        ("BiquadBPF", (1000, 2.0), dict()),
        ("BiquadBPFPeak", (1000, 2.0), dict()),
        ("BiquadNotch", (1000, 5.0), dict()),
        ("BiquadAllPass", (1000, 0.9), dict()),
    ]
    names, soses = [], {}
    for fs in (44100, 48000):
        for i, (cls, args, kw) in enumerate(specs):
            flt = getattr(F, cls)(*args, fs=fs, **kw)
            key = f"{i:02d}_{cls}_{fs}"
            names.append(repr((cls, list(args), kw, fs)))
            soses[key] = sos_of(flt)
    np.savez(os.path.join(OUT, "designs.npz"), specs=np.array(names), **soses)
    # DesignableFIR taps
    dd = {}
    for i, (cut, nt, kw) in enumerate([(5000, 1024, {}), (1000, 101, {}), ([300, 3000], 255, dict(pass_zero=False)),
                                       (8000, 64, dict(window="blackman"))]):
        flt = F.DesignableFIR(cutoff=cut, num_taps=nt, fs=48000, **kw)
        dd[f"k{i}"] = flt.kernel.numpy()
        dd[f"spec{i}"] = np.array(repr((cut, nt, kw)))
    np.savez(os.path.join(OUT, "fir_designs.npz"), **dd)

    # --------------------------------------------------------------------- FIR
    x = rnd((2, 20000), 8)
    d = {"x": x.numpy()}
    from scipy.signal import firwin
    for K in (5, 32, 1024):
        b = firwin(K, 5000, fs=48000) if K > 8 else np.array([0.1, 0.2, 0.4, 0.2, 0.1])
        fd = F.FIR(b, conv_mode="direct")
        ff = F.FIR(b, conv_mode="fft")
        d[f"k{K}"] = fd.kernel.numpy().reshape(-1)
        d[f"direct{K}"] = fd(x).numpy()
        d[f"fft{K}"] = ff(x).numpy()
        check(f"fir.direct{K}", O.fir_direct(x.numpy(), d[f"k{K}"]), d[f"direct{K}"], 2e-5)
        check(f"fir.fft{K}", O.fir_forward(x.numpy(), d[f"k{K}"], "fft"), d[f"fft{K}"], 2e-5)
    # short signal T < K  (tests/test_fir.py:122-131) and f64 input
    xs = rnd((1, 100), 9)
    b = firwin(32, 0.3)
    d["xs"], d["ks"] = xs.numpy(), F.FIR(b).kernel.numpy().reshape(-1)
    d["ys_fft"], d["ys_direct"] = F.FIR(b)(xs).numpy(), F.FIR(b, conv_mode="direct")(xs).numpy()
    xt = rnd((2, 20), 10, torch.float64)
    bt = firwin(64, 0.2)
    d["xt"], d["kt"] = xt.numpy(), F.FIR(bt).kernel.numpy().reshape(-1)
    d["yt_fft"], d["yt_direct"] = F.FIR(bt)(xt).numpy(), F.FIR(bt, conv_mode="direct")(xt).numpy()
    check("fir.short.fft", O.fir_forward(xs.numpy(), d["ks"], "fft"), d["ys_fft"], 2e-5)
    check("fir.TltK.fft", O.fir_forward(xt.numpy(), d["kt"], "fft"), d["yt_fft"], 4e-12)
    check("fir.TltK.direct", O.fir_forward(xt.numpy(), d["kt"], "direct"), d["yt_direct"], 4e-12)
    np.savez(os.path.join(OUT, "fir.npz"), **d)

    # ---------------------------------------------------------------- fft_conv1d
    d = {}
    x = rnd((2, 50000), 11)
    d["x"] = x.numpy()
    g = np.random.default_rng(0)
    for K in (64, 4097):
        k = (g.standard_normal(K) * np.exp(-np.arange(K) / (K / 8))).astype(np.float32)
        k /= np.abs(k).sum()
        d[f"k{K}"] = k
        d[f"y{K}"] = fft_conv1d(x[None], torch.from_numpy(k)[None, None], padding=(K - 1, 0))[0].numpy()
        check(f"fftconv.{K}", O.fft_conv1d(x.numpy(), k, (K - 1, 0)), d[f"y{K}"], 2e-5)
    # asymmetric padding (tests/test_fftconv.py)
    k16 = g.standard_normal(16).astype(np.float32)
    d["k16"] = k16
    d["y16_pad87"] = fft_conv1d(x[None], torch.from_numpy(k16)[None, None], padding=(8, 7))[0].numpy()
    check("fftconv.pad87", O.fft_conv1d(x.numpy(), k16, (8, 7)), d["y16_pad87"], 2e-5)
    # the 65536-tap reverb IR of cfg 4, one channel
    K = 65536
    ir = (np.random.default_rng(0).standard_normal(K) * np.exp(-np.arange(K) / 8000.0))
    ir = (ir / np.abs(ir).sum()).astype(np.float32)
    xl = rnd((1, 400000), 12)
    kf = ir[::-1].copy()
    yl = fft_conv1d(xl[None], torch.from_numpy(kf)[None, None], padding=(K - 1, 0))[0].numpy()
    d["x_long"], d["y_long"] = xl.numpy(), yl
    d["ir_seed"] = np.array(0)
    check("fftconv.65536", O.fft_conv1d(xl.numpy(), kf, (K - 1, 0)), yl, 2e-5)
    np.savez(os.path.join(OUT, "fftconv.npz"), **d)

    # -------------------------------------------- planner / chain / parallel (+)
    x = rnd((2, 6000), 13)
    f1 = F.HiButterworth(100, order=2, fs=None)
    f2 = F.LoButterworth(8000, order=4, fs=None)
    f3 = F.ParametricEQ(2000, 1.0, -3.0)
    fir = F.DesignableFIR(cutoff=6000, num_taps=127, fs=48000)
    w = Wave(x, 48000) | f1 | f2 | fir | f3
    y_chain = w.ys.numpy()
    d = dict(x=x.numpy(), y_chain=y_chain, sos_run1=np.vstack([sos_of(f1), sos_of(f2)]),
             sos_run2=sos_of(f3), fir_k=fir.kernel.numpy().reshape(-1))
    oy = O.chain_forward(x.numpy(), d["sos_run1"], [d["fir_k"]])
    oy, _, _ = O.iir_module_forward(oy, d["sos_run2"])
    check("planner.chain", oy, y_chain, 2e-5)
    # parallel combination
    p1 = F.LoButterworth(1000, order=2, fs=48000)
    p2 = F.HiButterworth(4000, order=2, fs=48000)
    ysum = (p1 + p2)(x).numpy()
    d.update(p1_sos=sos_of(p1), p2_sos=sos_of(p2), y_par=ysum)
    a, _, _ = O.iir_module_forward(x.numpy(), d["p1_sos"])
    b_, _, _ = O.iir_module_forward(x.numpy(), d["p2_sos"])
    check("parallel.sum", a + b_, ysum, 1e-6)
    # cfg-5 style chain, small: fused SOS -> FIR(1024, fft) -> FIR(4097-tap IR, fft)
    xc = rnd((2, 30000), 14)
    c1 = F.LoButterworth(2000, order=6, fs=48000)
    c2 = F.ParametricEQ(frequency=1000, q=2.0, gain=3.0, fs=48000)
    cf = F.FIR(firwin(1024, 5000, fs=48000))
    irs = (np.random.default_rng(1).standard_normal(4097) * np.exp(-np.arange(4097) / 500.0))
    irs = irs / np.abs(irs).sum()
    cr = F.FIR(irs)
    yc = (Wave(xc, 48000) | c1 | c2 | cf | cr).ys.numpy()
    d.update(xc=xc.numpy(), yc=yc, c_sos=np.vstack([sos_of(c1), sos_of(c2)]),
             c_fir=cf.kernel.numpy().reshape(-1), c_ir=cr.kernel.numpy().reshape(-1))
    check("chain.cfg5small", O.chain_forward(xc.numpy(), d["c_sos"], [d["c_fir"], d["c_ir"]]), yc, 2e-5)
    np.savez(os.path.join(OUT, "chain.npz"), **d)

    # chain whose IIR run has a non-trivial gain (+12 dB shelf, +9 dB high-Q peak, 80 Hz high-pass) in front
    # of two FFT-mode FIRs: the case the planner's spectral folding (IIR run -> taps of the FIR run) must
    # reproduce; the reference stages it as cascade -> FIR -> FIR
    xg = rnd((2, 100000), 31)          # long enough for the LDS-resident overlap-save path (one 65536 block + tail)
    xg = (0.5 * xg + 0.5 * torch.sin(2 * np.pi * 500.0 / 48000.0 * torch.arange(100000, dtype=torch.float64))[None, :]
          .to(torch.float32) * torch.tensor([[1.0], [-0.7]])).to(torch.float32)
    g1 = F.HiShelving(3000, q=0.7, gain=4.0, fs=48000)
    g2 = F.ParametricEQ(frequency=500, q=4.0, gain=9.0, fs=48000)
    g3 = F.HiButterworth(80, order=2, fs=48000)
    gf = F.FIR(firwin(257, 6000, fs=48000))
    irg = (np.random.default_rng(2).standard_normal(2049) * np.exp(-np.arange(2049) / 300.0))
    irg = 8.0 * irg / np.abs(irg).sum()
    gr = F.FIR(irg)
    yg = (Wave(xg, 48000) | g1 | g2 | g3 | gf | gr).ys.numpy()
    dg = dict(x=xg.numpy(), y=yg, sos=np.vstack([sos_of(g1), sos_of(g2), sos_of(g3)]),
              fir=gf.kernel.numpy().reshape(-1), ir=gr.kernel.numpy().reshape(-1))
    check("chain.gain", O.chain_forward(xg.numpy(), dg["sos"], [dg["fir"], dg["ir"]]), yg, 2e-5)
    assert float(np.abs(yg).max()) > 1.5, float(np.abs(yg).max())     # the fixture really has gain
    np.savez(os.path.join(OUT, "chain_gain.npz"), **dg)

    # delay line (export kept for API compat)
    xd = rnd((2, 1000), 15)
    yd = _ops.delay_line_forward(xd, 100, 0.5, 0.3).numpy()
    np.savez(os.path.join(OUT, "delay.npz"), x=xd.numpy(), y=yd)
    check("delay", O.delay_line(xd.numpy(), 100, 0.5, 0.3), yd, 1e-7)

    # ---------------------------------------------- effects between filters
    from torchfx import effect as E
    d = {}
    xe = rnd((4, 5000), 21) * 1.7                      # peaks above 1 so that clamp does something
    xe[3] = 0.0                                         # a silent channel (per-channel zero rule)
    xe64 = rnd((2, 3001), 22, torch.float64)
    d["x"], d["x64"], d["zeros"] = xe.numpy(), xe64.numpy(), np.zeros((2, 64), np.float32)
    for tag, kw in {"amp": dict(gain=0.37, gain_type="amplitude"), "db": dict(gain=-4.5, gain_type="db"),
                    "db0": dict(gain=0.0, gain_type="db"), "pow": dict(gain=2.5, gain_type="power"),
                    "clamp": dict(gain=1.9, gain_type="amplitude", clamp=True)}.items():
        d["gain_" + tag] = E.Gain(**kw)(xe).numpy()
        check("gain." + tag, O.gain(xe.numpy(), **kw), d["gain_" + tag], 1e-7)
    d["gain64_db"] = E.Gain(3.0, "db")(xe64).numpy()
    check("gain.f64", O.gain(xe64.numpy(), 3.0, "db"), d["gain64_db"], 1e-15)
    strat = {"peak": E.PeakNormalizationStrategy(), "rms": E.RMSNormalizationStrategy(),
             "percentile": E.PercentileNormalizationStrategy(97.0), "per_channel": E.PerChannelNormalizationStrategy()}
    for tag, st in strat.items():
        d["norm_" + tag] = E.Normalize(peak=0.8, strategy=st)(xe).numpy()
        d["norm64_" + tag] = E.Normalize(peak=1.25, strategy=st)(xe64).numpy()
        d["normz_" + tag] = E.Normalize(peak=0.8, strategy=st)(torch.from_numpy(d["zeros"])).numpy()
        check("norm." + tag, O.normalize(xe.numpy(), 0.8, tag, 97.0), d["norm_" + tag], 1e-6)
        check("norm64." + tag, O.normalize(xe64.numpy(), 1.25, tag, 97.0), d["norm64_" + tag], 1e-14)
        check("normz." + tag, O.normalize(d["zeros"], 0.8, tag, 97.0), d["normz_" + tag], 0.0)
    x3 = rnd((2, 3, 700), 23)
    d["x3"], d["norm3_per_channel"] = x3.numpy(), E.Normalize(0.5, E.PerChannelNormalizationStrategy())(x3).numpy()
    check("norm3.per_channel", O.normalize(x3.numpy(), 0.5, "per_channel"), d["norm3_per_channel"], 1e-6)
    # the mixed pipeline of tests/test_chain_fusion.py:102-121: IIR runs around a Gain
    xm = rnd((2, 6000), 24)
    fl = [F.LoButterworth(4000, order=2), F.HiButterworth(200, order=2), E.Gain(0.5),
          F.LoButterworth(6000, order=2), F.HiButterworth(100, order=2)]
    wv = Wave(xm, 48000)
    for m in fl:
        wv = wv | m
    d["mix_x"], d["mix_y"] = xm.numpy(), wv.ys.numpy()
    d["mix_sos_a"] = np.vstack([sos_of(fl[0]), sos_of(fl[1])])
    d["mix_sos_b"] = np.vstack([sos_of(fl[3]), sos_of(fl[4])])
    ya = O.iir_module_forward(xm.numpy(), d["mix_sos_a"])[0]
    yb = O.iir_module_forward(O.gain(ya, 0.5), d["mix_sos_b"])[0]
    check("mix.iir-gain-iir", yb, d["mix_y"], 1e-7)
    np.savez(os.path.join(OUT, "effects.npz"), **d)

    print(f"{'vector':34s} {'max|oracle-ref|':>16s} {'tol':>9s}")
    for n, e, t in report:
        print(f"{n:34s} {e:16.3e} {t:9.1e}")
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print(f"wrote {len(os.listdir(OUT))} fixtures, {tot / 1e6:.2f} MB -> {OUT}")


if __name__ == "__main__":
    main()
