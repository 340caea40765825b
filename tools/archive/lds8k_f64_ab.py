"""float64: 4096- vs 8192-point block of the one-launch overlap-save kernel (development)."""
import os, sys
os.environ.setdefault("TFX_ENV_DYNAMIC", "1")      # this tool flips TFX_* knobs inside one process
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tools.quick_bench import timed, E
x = torch.randn(32, 2_880_000, device="cuda:0", dtype=torch.float64)
for K in [int(a) for a in sys.argv[1:]] or [256, 512, 1024, 1500, 2048]:
    k = np.random.default_rng(0).standard_normal(K)
    for rep in range(3):
        for lg in ("12", "13"):
            os.environ["TFX_FFT_LOG2N"] = lg
            wall, prof = timed(lambda: E.fft_conv_forward(x, k, (K - 1, 0)), reps=20, warm=3)
            print("f64 K", K, "log2N", lg, f"wall {wall:.4f} ms kernel {sum(prof.values()):.4f}", flush=True)
