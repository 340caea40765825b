import os, sys, json, time
os.environ.setdefault("TFX_ENV_DYNAMIC", "1")      # this tool flips TFX_* knobs inside one process
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchfx_amd import _lib, torchfx_ext as E
from tools.quick_bench import timed
C, T, K = 64, 28_800_000, 65536
x = torch.randn(C, T, device="cuda:0")
ir = np.random.default_rng(0).standard_normal(K) * np.exp(-np.arange(K) / 8000.0)
k = (ir / np.abs(ir).sum()).astype(np.float32)[::-1].copy()
for lg in (18, 20):
    for r4 in (0, 1):
        os.environ["TFX_FFT_LOG2N"] = str(lg)
        os.environ["TFX_OLS_ROW_R4"] = str(r4)
        wall, prof = timed(lambda: E.fft_conv_forward(x, k, (K - 1, 0)), reps=3, warm=1)
        print(f"log2N={lg} col_r4={r4}: wall {wall:7.3f} ms  {C*T/wall/1e3:9.1f} Msamp/s  " + " ".join(f"{n.replace('_kernel','')}={v:.2f}" for n, v in prof.items()), flush=True)
