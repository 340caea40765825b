import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torchfx_amd import torchfx_ext as E
from tools.quick_bench import timed
from scipy.signal import butter
C = 64
for T in (2_880_000, 28_800_000, 10_000_000):
    x = torch.randn(C, T, device="cuda:0")
    for K in (1, 4):
        sos = butter(2 * K, 2000 / 24000, output="sos")
        st = torch.from_numpy(sos)
        for prec in ("f64", "f32"):
            res = []
            for rnd in (32, 64, 128, 256, 512, 1024, 2048, 4096):
                os.environ["TFX_SOS_WARM_ROUND"] = str(rnd)
                wall, prof = timed(lambda: E.sos_forward(x, None, st, None, None, precision=prec), reps=10, warm=3)
                ms = prof.get("sos_stream_kernel<%s>" % prec, 0)
                res.append(f"{rnd}:{ms:.3f}")
            print(f"T={T} K={K} {prec} round->ms  " + "  ".join(res), flush=True)
    del x
