"""Experiment (round 6): `iir-cascade | FIR-1024` on 64 x 2.88 M as the two existing kernels (sos_stream_kernel, ols_lds8192_kernel)
run over CHANNEL slabs on two streams -- the cascade of slab i+1 beside the overlap-save of slab i, the intermediate of a slab
small enough for the Infinity Cache.  usage: python tools/experiments/iir_fir_slab_pipe.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torchfx_amd import torchfx_ext as E  # noqa: E402
from torchfx_amd import filter as F  # noqa: E402

C, T = 64, 2_880_000
f1 = F.LoButterworth(2000, order=6, fs=48000)
f2 = F.ParametricEQ(frequency=1000, q=2.0, gain=3.0, fs=48000)
f1.compute_coefficients(); f2.compute_coefficients()
sos = torch.cat([f1._sos, f2._sos])
K = 1024
k = np.random.default_rng(0).standard_normal(K) * np.exp(-np.arange(K) / 200.0)
k = torch.from_numpy((k / np.abs(k).sum()).astype(np.float32)[::-1].copy())
x = torch.rand((C, T), device="cuda") * 2 - 1
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def staged():
    y, _, _ = E.sos_forward(x, None, sos, None, None)
    return E.fft_conv_forward(y, k, (K - 1, 0))


def piped(slab):
    outs = []
    cur = torch.cuda.current_stream()
    ev0 = torch.cuda.Event(); ev0.record(cur)
    s1.wait_event(ev0); s2.wait_event(ev0)
    for c0 in range(0, C, slab):
        with torch.cuda.stream(s1):
            y, _, _ = E.sos_forward(x[c0:c0 + slab], None, sos, None, None)
            ev = torch.cuda.Event(); ev.record(s1)
        with torch.cuda.stream(s2):
            s2.wait_event(ev)
            outs.append(E.fft_conv_forward(y, k, (K - 1, 0)))
            y.record_stream(s2)
    e1 = torch.cuda.Event(); e1.record(s1); e2 = torch.cuda.Event(); e2.record(s2)
    cur.wait_event(e1); cur.wait_event(e2)
    return outs


def timed(fn, name, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / reps * 1e3)
    ts.sort()
    print(f"{name:32s} min {ts[0]:.4f} med {ts[len(ts) // 2]:.4f} ms  frac {8 * C * T / ts[len(ts) // 2] / 1e9 / 8:.3f}", flush=True)


timed(staged, "staged (two launches)")
timed(lambda: E.sos_forward(x, None, sos, None, None), "cascade alone")
y0, _, _ = E.sos_forward(x, None, sos, None, None)
timed(lambda: E.fft_conv_forward(y0, k, (K - 1, 0)), "overlap-save alone")
for slab in (4, 8, 16, 32):
    timed(lambda: piped(slab), f"piped, {slab}-channel slabs")
ref = staged()
got = torch.cat(piped(8))
print("equal:", torch.equal(ref, got))
