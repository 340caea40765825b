"""Cascade kernel: time against the number of sections (1 ... 4 of cfg 2's), old structure vs LDS-DMA prefetch (development)."""
import os, sys
os.environ.setdefault("TFX_ENV_DYNAMIC", "1")      # this tool flips TFX_* knobs inside one process
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy.signal import butter
from tools.quick_bench import timed, E
x = torch.randn(64, 2_880_000, device="cuda:0")
full = np.vstack([butter(6, 2000 / 24000, output="sos"), np.array([[1.0089, -1.9636, 0.9695, 1, -1.9636, 0.9784]])])
for K in (1, 2, 3, 4):
    sos = torch.from_numpy(full[:K].copy())
    for rep in range(2):
        for dma in ("0", "1"):
            os.environ["TFX_SOS_DMA"] = dma
            wall, prof = timed(lambda: E.sos_forward(x, None, sos, None, None), reps=30, warm=3)
            print(f"K={K} DMA={dma}: kernel {sum(prof.values()):.4f} ms  {list(prof)}", flush=True)
