"""Cascade kernel time against row length (cfg 2's filter, 64 rows): fixed cost per launch vs streaming rate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy.signal import butter
from tools.quick_bench import timed, E
sos = torch.from_numpy(np.vstack([butter(6, 2000 / 24000, output="sos"), np.array([[1.0089, -1.9636, 0.9695, 1, -1.9636, 0.9784]])]))
for prec in ("f64", "f32"):
    for T in (360_000, 720_000, 1_440_000, 2_880_000, 5_760_000, 11_520_000, 28_800_000):
        x = torch.randn(64, T, device="cuda:0")
        best = 1e9
        for rep in range(3):
            wall, prof = timed(lambda: E.sos_forward(x, None, sos, None, None, precision=prec), reps=max(4, int(60e6 / T)), warm=3)
            best = min(best, sum(prof.values()))
        print(f"{prec} T={T:9d}: kernel {best:.4f} ms  {best / (64 * T) * 1e9:.3f} ps/sample  {8 * 64 * T / best / 1e9 / 8 * 100:.1f}% of 8 TB/s", flush=True)
        del x
