"""Probe (round 6, VERDICT r5 #4b): what would rows of 16 384 samples (N = 2^22 = 256 x 16384) buy the recursion pass A'?
A row pays its warm-up once per 512 blocks instead of once per 256, a block wastes 1.6 % instead of 3.3 % on the overlap.
TFX_OLS_SOS_PROBE_A: 1 = pass A' alone at the default 2^21-point block, 22 = alone at 2^22 (passes B / C skipped: the output is
garbage, only the time counts).  usage: python tools/experiments/sos_rows16k_probe.py"""
import os
os.environ["TFX_ENV_DYNAMIC"] = "1"
import sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torchfx_amd import torchfx_ext as E
from torchfx_amd import filter as F

C, T, K = 64, 28_800_000, 66559
f1 = F.LoButterworth(2000, order=6, fs=48000); f2 = F.ParametricEQ(frequency=1000, q=2.0, gain=3.0, fs=48000)
f1.compute_coefficients(); f2.compute_coefficients()
sos = torch.cat([f1._sos, f2._sos])
k = np.random.default_rng(0).standard_normal(K) * np.exp(-np.arange(K) / 8000.0)
k = torch.from_numpy((k / np.abs(k).sum()).astype(np.float32)[::-1].copy())
x = torch.rand((C, T), device="cuda") * 2 - 1


def timed(name, reps=7):
    fn = lambda: E.sos_fft_conv_forward(x, sos, k, (K - 1, 0))
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    info = E.sos_fft_conv_plan_info(T, sos, K, (K - 1, 0))
    print(f"{name:44s} N=2^{info['N'].bit_length() - 1} F={info['F']:3d}  min {ts[0]:7.3f} med {ts[len(ts) // 2]:7.3f} ms", flush=True)


print("warm-up samples:", E.sos_fft_conv_warmup(sos))
timed("full step (A' + B + C), default")
for lanes in ("3", "1"):
    os.environ["TFX_OLS_SOS_STREAMS"] = lanes
    os.environ["TFX_OLS_SOS_PROBE_A"] = "1"
    timed(f"A' alone, 8192-sample rows, {lanes} lane(s)")
    os.environ["TFX_OLS_SOS_PROBE_A"] = "22"
    timed(f"A' alone, 16384-sample rows, {lanes} lane(s)")
    os.environ["TFX_OLS_SOS_PROBE_A"] = "0"
