"""Second sweep of the cascade-in-pass-A chain step: staggered lanes on / off x lanes x pairs per slab (wall clock)."""
import os, sys, time
os.environ.setdefault("TFX_ENV_DYNAMIC", "1")      # this tool flips TFX_* knobs inside one process
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchfx_amd import torchfx_ext as E
from torchfx_amd import filter as F
C, T = 64, 28_800_000
f1 = F.LoButterworth(2000, order=6, fs=48000); f2 = F.ParametricEQ(frequency=1000, q=2.0, gain=3.0, fs=48000)
f1.compute_coefficients(); f2.compute_coefficients()
sos = torch.cat([f1._sos, f2._sos])
K = 66559
k = np.random.default_rng(0).standard_normal(K) * np.exp(-np.arange(K) / 8000.0)
k = torch.from_numpy((k / np.abs(k).sum()).astype(np.float32)[::-1].copy())
x = torch.rand((C, T), device="cuda") * 2 - 1
def timed(name, reps=5):
    fn = lambda: E.sos_fft_conv_forward(x, sos, k, (K - 1, 0))
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort(); print(f"{name:44s} min {ts[0]:7.3f} med {ts[len(ts)//2]:7.3f} ms", flush=True)
for stag in (0, 1):
    for streams in (2, 3, 4, 6):
        for pairs in (96, 160, 240, 320, 480):
            os.environ.update(TFX_OLS_SOS_STAGGER=str(stag), TFX_OLS_SOS_STREAMS=str(streams), TFX_OLS_SOS_PAIRS=str(pairs))
            timed(f"stagger={stag} streams={streams} pairs={pairs}")
