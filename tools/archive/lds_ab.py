"""A/B of the LDS overlap-save kernel's knobs (development).  usage: lds_ab.py ENV=v1,v2,... [ENV2=...]  (cartesian product)"""
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy.signal import firwin
from tools.quick_bench import timed, E

knobs = [(a.split("=")[0], a.split("=")[1].split(",")) for a in sys.argv[1:] if "=" in a]
Ks = [int(a) for a in sys.argv[1:] if "=" not in a] or [1024]
C, T = 64, 2_880_000
x = torch.randn(C, T, device="cuda:0")
for K in Ks:
    k = firwin(K, 5000, fs=48000).astype(np.float32)[::-1].copy()
    for rep in range(3):
        for combo in itertools.product(*[v for _, v in knobs]):
            for (name, _), val in zip(knobs, combo):
                os.environ[name] = val
            wall, prof = timed(lambda: E.fft_conv_forward(x, k, (K - 1, 0)), reps=30, warm=3)
            print(f"K={K:5d} " + " ".join(f"{n}={v}" for (n, _), v in zip(knobs, combo)) + f": wall {wall:.4f} ms kernel {sum(prof.values()):.4f} ms", flush=True)
