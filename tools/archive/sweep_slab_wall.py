import os, sys, time
os.environ.setdefault("TFX_ENV_DYNAMIC", "1")      # this tool flips TFX_* knobs inside one process
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchfx_amd import _lib, torchfx_ext as E
C, T, K = 64, 28_800_000, 65536
x = torch.randn(C, T, device="cuda:0")
ir = np.random.default_rng(0).standard_normal(K) * np.exp(-np.arange(K) / 8000.0)
k = (ir / np.abs(ir).sum()).astype(np.float32)[::-1].copy()
for lg in (20,):
    for mb in (24, 32, 48, 64, 96, 128):
        os.environ["TFX_FFT_LOG2N"] = str(lg); os.environ["TFX_OLS_SLAB_MB"] = str(mb)
        for _ in range(2): E.fft_conv_forward(x, k, (K - 1, 0))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): E.fft_conv_forward(x, k, (K - 1, 0))
        torch.cuda.synchronize(); w = (time.perf_counter() - t0) / 3 * 1e3
        print(f"log2N={lg} slab={mb:4d} MB: wall {w:7.3f} ms (no event profiling)", flush=True)
