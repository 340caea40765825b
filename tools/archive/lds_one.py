"""FIR through the FFT mode on cfg-3's shape, a few calls (for rocprofv3 passes).  usage: lds_one.py [K] [dtype] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy.signal import firwin
from torchfx_amd import torchfx_ext as E

K = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dt = torch.float64 if len(sys.argv) > 2 and sys.argv[2] == "f64" else torch.float32
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
x = torch.randn(64, 2_880_000, device="cuda:0", dtype=dt)
k = firwin(K, 5000, fs=48000).astype(np.float32).astype(np.float64 if dt == torch.float64 else np.float32)[::-1].copy()
for _ in range(reps):
    y = E.fft_conv_forward(x, k, (K - 1, 0))
torch.cuda.synchronize()
print("ok", float(y[0, 1000]))
