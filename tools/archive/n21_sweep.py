"""N = 2^21 (256 x 8192) blocks: slab / lane sweep of the plain overlap-save pipeline and of the recursion-in-pass-A chain."""
import os, sys, time
os.environ.setdefault("TFX_ENV_DYNAMIC", "1")
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
def timed(fn, name, n=12):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print(f"{name:60s} {(time.perf_counter() - t0) * 1e3 / n:7.3f} ms / step", flush=True)
what = sys.argv[1] if len(sys.argv) > 1 else "ols,fused"
for n21 in (0, 1):
    os.environ["TFX_OLS_N21"] = str(n21)
    if "ols" in what:
        for streams in (2, 3):
            for mb in (64, 128, 256, 512):
                os.environ.update(TFX_OLS_STREAMS=str(streams), TFX_OLS_SLAB_MB=str(mb))
                timed(lambda: E.fft_conv_forward(x, k, (K - 1, 0)), f"N21={n21} plain OLS streams={streams} slab={mb} MB")
    os.environ.pop("TFX_OLS_STREAMS", None); os.environ.pop("TFX_OLS_SLAB_MB", None)
    if "fused" in what:
        for streams in (2, 3, 4):
            for pairs in (80, 120, 160, 240):
                os.environ.update(TFX_OLS_SOS_STREAMS=str(streams), TFX_OLS_SOS_PAIRS=str(pairs * (1 if n21 else 2)))
                timed(lambda: E.sos_fft_conv_forward(x, sos, k, (K - 1, 0)), f"N21={n21} fused streams={streams} pairs={pairs * (1 if n21 else 2)}")
