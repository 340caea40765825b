import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torchfx_amd import torchfx_ext as E
from tools.quick_bench import timed
from scipy.signal import butter
C = 64
for T in (2_880_000,):
    x = torch.randn(C, T, device="cuda:0")
    for K in (1, 4):
        sos = butter(2 * K, 2000 / 24000, output="sos")
        st = torch.from_numpy(sos)
        print("plan", E.sos_plan_info(sos))
        for prec in ("f64", "f32"):
            for var in (0, 1, 2, 3, 4, 5):
                for wpc in (0, 4, 8, 12, 16):
                    os.environ["TFX_SOS_VARIANT"] = str(var)
                    os.environ["TFX_SOS_WAVES_PER_CU"] = str(wpc)
                    try:
                        wall, prof = timed(lambda: E.sos_forward(x, None, st, None, None, precision=prec), reps=10, warm=3)
                    except Exception as e:
                        print("ERR", var, wpc, e); continue
                    ms = prof.get("sos_stream_kernel<%s>" % prec, 0)
                    print(f"T={T} K={K} {prec} var={var} wpc={wpc:2d}: {ms:.3f} ms = {8*C*T/ms/1e9:.2f} TB/s", flush=True)
