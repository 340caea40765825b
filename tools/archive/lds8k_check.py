"""8192-point LDS overlap-save (TFX_FFT_LOG2N=13) against a float64 FFT convolution (development)."""
import os, sys
os.environ.setdefault("TFX_ENV_DYNAMIC", "1")      # this tool flips TFX_* knobs inside one process
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy.signal import fftconvolve
from torchfx_amd import torchfx_ext as E

os.environ["TFX_FFT_LOG2N"] = "13"
worst = 0.0
for C, T, K in [(2, 44100, 1024), (3, 100_003, 4096), (1, 9000, 5), (5, 250_003, 3000), (2, 12_289, 2049), (64, 40_000, 384), (1, 1, 1), (4, 98_304, 4000)]:
    rng = np.random.default_rng(K + T)
    kf = (rng.standard_normal(K) / np.sqrt(K)).astype(np.float32)
    x = rng.uniform(-1, 1, (C, T)).astype(np.float32)
    for pad in ((K - 1, 0), (100, 77), (0, K)):
        if T + pad[0] + pad[1] < K:
            continue
        info = E.ols_plan_info(K, T, pad)
        y = E.fft_conv_forward(torch.from_numpy(x).cuda(), kf, pad).cpu().numpy()
        xp = np.pad(x.astype(np.float64), ((0, 0), pad))
        ref = fftconvolve(xp, kf[::-1].astype(np.float64)[None], mode="valid", axes=-1)
        err = np.abs(y - ref).max() / max(1.0, np.abs(ref).max())
        worst = max(worst, err)
        print(C, T, K, pad, info["path"], info["N"], info["S"], f"{err:.2e}", flush=True)
print("worst", worst)
assert worst < 4e-6
