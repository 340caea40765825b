"""A/B of the cascade kernel's knobs on cfg 2 (64 x 2.88 M, 4 sections).  usage: sos_ab.py ENV=v1,v2 ... [prec]"""
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy.signal import butter
from tools.quick_bench import timed, E

knobs = [(a.split("=")[0], a.split("=")[1].split(",")) for a in sys.argv[1:] if "=" in a]
precs = [a for a in sys.argv[1:] if "=" not in a] or ["f64"]
for T in (2_880_000, 28_800_000):
    x = torch.randn(64, T, device="cuda:0")
    sos = torch.from_numpy(np.vstack([butter(6, 2000 / 24000, output="sos"), np.array([[1.0089, -1.9636, 0.9695, 1, -1.9636, 0.9784]])]))
    for prec in precs:
        for rep in range(3):
            for combo in itertools.product(*[v for _, v in knobs]):
                for (name, _), val in zip(knobs, combo):
                    os.environ[name] = val
                wall, prof = timed(lambda: E.sos_forward(x, None, sos, None, None, precision=prec), reps=40 if T < 1e7 else 8, warm=3)
                ms = sum(prof.values())
                print(f"T={T} {prec} " + " ".join(f"{n}={v}" for (n, _), v in zip(knobs, combo)) + f": kernel {ms:.4f} ms wall {wall:.4f}  {8 * 64 * T / ms / 1e9 / 8 * 100:.1f}% of 8 TB/s", flush=True)
    del x
