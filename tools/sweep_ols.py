import os, sys, json, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchfx_amd import _lib, torchfx_ext as E
from tools.quick_bench import timed
C, T, K = 64, 28_800_000, 65536
x = torch.randn(C, T, device="cuda:0")
ir = np.random.default_rng(0).standard_normal(K) * np.exp(-np.arange(K) / 8000.0)
k = (ir / np.abs(ir).sum()).astype(np.float32)[::-1].copy()
for slab in (4, 8, 16, 32, 64, 128, 256, 1024):
    os.environ["TFX_OLS_PAIRS_PER_SLAB"] = str(slab)
    wall, prof = timed(lambda: E.fft_conv_forward(x, k, (K - 1, 0)), reps=2, warm=1)
    print(f"slab={slab:5d} pairs ({slab*2} MB): wall {wall:7.3f} ms  {C*T/wall/1e3:9.1f} Msamp/s  " + " ".join(f"{n.replace('_kernel','')}={v:.2f}" for n, v in prof.items()), flush=True)
