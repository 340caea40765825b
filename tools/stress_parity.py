#!/usr/bin/env python3
"""Randomised parity stress (GPU box): many more geometries than the test suite draws -- fft_conv_forward (float32 / float64, every
path the dispatch picks: one-launch 4096 / 8192 / 16 384 points, three passes 2^16 ... 2^21, float64 three passes, rocFFT) and
sos_fft_conv_forward (cascade inside pass A, both block sizes) on rows of ANY length at ANY float alignment, against a float64
SciPy correlation / the staged HIP pair.  Prints one line per failure and a summary; exit code 1 on any failure.

usage: stress_parity.py [n_conv, default 150] [n_fused, default 40] [seed0, default 0]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import scipy.signal as sg
import torch

from torchfx_amd import torchfx_ext as E

DEV = "cuda:0"
n_conv = int(sys.argv[1]) if len(sys.argv) > 1 else 150
n_fused = int(sys.argv[2]) if len(sys.argv) > 2 else 40
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 0
fails = 0
paths: dict = {}


def placed(x: np.ndarray, off: int) -> torch.Tensor:
    """x on the device, its first element `off` elements into a 128-byte line."""
    C, T = x.shape
    buf = torch.zeros(C * T + 64, dtype=torch.from_numpy(x).dtype, device=DEV)
    v = buf[off: off + C * T].view(C, T)
    v.copy_(torch.from_numpy(x))
    return v


def err(got, exp):
    return float(np.abs(got.astype(np.float64) - exp.astype(np.float64)).max()) / max(1.0, float(np.abs(exp).max()))


for i in range(n_conv):
    rng = np.random.default_rng(seed0 + i)
    f64 = rng.random() < 0.3
    dt = np.float64 if f64 else np.float32
    K = int(np.exp(rng.uniform(np.log(2), np.log(150_000 if not f64 else 40_000))))
    C = int(rng.integers(1, 6))
    T = int(np.exp(rng.uniform(np.log(max(K, 64)), np.log(3_500_000))))
    pl = int(rng.choice([K - 1, 0, K // 2, K - 1 + int(rng.integers(0, 40))]))
    pr = int(rng.choice([0, 0, K // 2, int(rng.integers(0, 40))]))
    if T + pl + pr < K:
        pl = K - 1
    off = int(rng.integers(0, 32 if not f64 else 16)) if rng.random() < 0.5 else 0
    x = rng.standard_normal((C, T)).astype(dt)
    x /= np.abs(x).max()
    k = rng.standard_normal(K) * np.exp(-np.arange(K) / max(4.0, K / 6.0))
    kf = (k / np.abs(k).sum()).astype(dt)[::-1].copy()
    info = E.ols_plan_info(K, T, (pl, pr), torch.float64 if f64 else torch.float32)
    key = (info["path"], info["N"], "f64" if f64 else "f32")
    paths[key] = paths.get(key, 0) + 1
    y = E.fft_conv_forward(placed(x, off), torch.from_numpy(kf), (pl, pr)).cpu().numpy()
    xp = np.pad(x.astype(np.float64), ((0, 0), (pl, pr)))
    exp = sg.fftconvolve(xp, kf[::-1].astype(np.float64)[None], mode="valid", axes=-1)
    e = err(y, exp)
    tol = 1e-11 if f64 else 1e-5
    if y.shape != exp.shape or not np.isfinite(e) or e > tol:
        fails += 1
        print(f"FAIL conv seed {seed0 + i}: {key} C={C} T={T} K={K} pad=({pl},{pr}) off={off}: err {e:.3e} shape {y.shape} vs {exp.shape}", flush=True)

for i in range(n_fused):
    rng = np.random.default_rng(10_000 + seed0 + i)
    block = int(rng.integers(1, 3))
    C = int(rng.integers(1, 5))
    T = int(rng.integers(1 << 16, (3 << 20) * block))
    K = int(rng.integers(33, 120_000))
    if (1 << (19 + block)) < 2 * (K + 32):
        K = int(rng.integers(33, 200_000 if block == 1 else 400_000)) % ((1 << (18 + block)) - 64) + 33
    nsec = int(rng.integers(1, 9))
    sos = np.ascontiguousarray(np.vstack([sg.butter(2, float(rng.uniform(0.03, 0.6)), ["lowpass", "highpass"][int(rng.integers(0, 2))], output="sos")[0]
                                          for _ in range(nsec)]))
    pad = (K - 1 + int(rng.integers(0, 50)) * int(rng.integers(0, 2)), 0)
    if T + pad[0] < K or not E.sos_fft_conv_supported(T, sos, K, pad, force_block=block):
        continue
    off = int(rng.integers(0, 32)) if rng.random() < 0.5 else 0
    x = rng.standard_normal((C, T)).astype(np.float32)
    x /= np.abs(x).max()
    k = rng.standard_normal(K) * np.exp(-np.arange(K) / max(8.0, K / 8.0))
    kf = torch.from_numpy((k / np.abs(k).sum()).astype(np.float32)[::-1].copy())
    xv = placed(x, off)
    y = E.sos_fft_conv_forward(xv, sos, kf, pad, force_block=block)
    ys, _, _ = E.sos_forward(xv, None, torch.from_numpy(sos), None, None)
    ys = E.fft_conv_forward(ys, kf, pad)
    e = err(y.cpu().numpy(), ys.cpu().numpy())
    paths[("fused", 1 << (19 + block), "f32")] = paths.get(("fused", 1 << (19 + block), "f32"), 0) + 1
    if tuple(y.shape) != tuple(ys.shape) or not np.isfinite(e) or e > 2e-6:
        fails += 1
        print(f"FAIL fused seed {seed0 + i}: block 2^{19 + block} C={C} T={T} K={K} sections={nsec} pad={pad} off={off}: err {e:.3e}", flush=True)

print("paths exercised:", {f"{p[0]}/{p[1]}/{p[2]}": n for p, n in sorted(paths.items())})
print(f"stress_parity: {n_conv} convolutions + {n_fused} fused draws, {fails} failures")
sys.exit(1 if fails else 0)
