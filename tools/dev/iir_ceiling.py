"""Where is the ceiling for the cascade kernel?  copy (torch) vs cascade with K = 1, 2, 4, 8 sections, f32 and f64
arithmetic, at the cfg-2 size and at 10 x that."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torchfx_amd import torchfx_ext as E
from tools.quick_bench import timed
from scipy.signal import butter

def ev(fn, reps=10, warm=3):
    for _ in range(warm): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return float(np.median(ts)), float(np.min(ts))

for T in (2_880_000, 28_800_000):
    C = 64
    x = torch.randn(C, T, device="cuda:0")
    y = torch.empty_like(x)
    med, mn = ev(lambda: y.copy_(x))
    print(f"T={T}: torch copy  median {med:.3f} min {mn:.3f} ms  = {8*C*T/med/1e9:.2f} TB/s", flush=True)
    med, mn = ev(lambda: torch.mul(x, 1.5, out=y))
    print(f"T={T}: torch mul   median {med:.3f} min {mn:.3f} ms  = {8*C*T/med/1e9:.2f} TB/s", flush=True)
    for K in (1, 2, 4, 8):
        sos = butter(2 * K, 2000 / 24000, output="sos")
        st = torch.from_numpy(sos)
        for prec in ("f64", "f32"):
            wall, prof = timed(lambda: E.sos_forward(x, None, st, None, None, precision=prec), reps=10, warm=3)
            ks = {k: v for k, v in prof.items()}
            med, mn = ev(lambda: E.sos_forward(x, None, st, None, None, precision=prec))
            print(f"T={T}: K={K} {prec}: kernels {ks}  event median {med:.3f} min {mn:.3f} ms = {8*C*T/med/1e9:.2f} TB/s ({8*C*T/med/1e9/8*100:.1f} %)", flush=True)
    del x, y
