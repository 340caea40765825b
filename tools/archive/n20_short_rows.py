"""65 536 / 32 768 / 16 385-tap FFT convolution on 64 x 2.88 M: the 2^18-point block the policy picks for rows < 4 M samples against 2^20 / 2^16."""
import os, sys, time
os.environ.setdefault("TFX_ENV_DYNAMIC", "1")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchfx_amd import torchfx_ext as E
x = torch.rand((64, 2_880_000), device="cuda") * 2 - 1
def timed(fn, n=40):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n
for K in (65536, 40000, 32768, 20000, 16385, 9000):
    k = np.random.default_rng(0).standard_normal(K) * np.exp(-np.arange(K) / (K / 8))
    k = torch.from_numpy((k / np.abs(k).sum()).astype(np.float32)[::-1].copy())
    row = []
    for lg in ("0", "16", "18", "20"):
        os.environ["TFX_FFT_LOG2N"] = lg
        try:
            info = E.ols_plan_info(K, 2_880_000, (K - 1, 0))
            row.append(f"log2N={lg}: N={info['N']} {timed(lambda: E.fft_conv_forward(x, k, (K - 1, 0))):.3f} ms")
        except Exception as e:
            row.append(f"log2N={lg}: n/a")
    print(K, " | ".join(row), flush=True)
