"""Developer sweep: cfg-4 overlap-save (64 ch x 600 s, 65536 taps) under sets of env knobs.
usage: python tools/ols_knobs.py "A=1,B=2" "A=0" ...   ('' = defaults).  Each set is timed on one
internal stream (clean per-kernel times) and with the default two streams (wall)."""
import os
os.environ.setdefault("TFX_ENV_DYNAMIC", "1")      # this tool flips TFX_* knobs inside one process
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchfx_amd import torchfx_ext as E  # noqa: E402
from tools.quick_bench import timed  # noqa: E402

C, T, K = 64, 28_800_000, 65536
x = torch.randn(C, T, device="cuda:0")
ir = np.random.default_rng(0).standard_normal(K) * np.exp(-np.arange(K) / 8000.0)
k = (ir / np.abs(ir).sum()).astype(np.float32)[::-1].copy()
base_env = dict(os.environ)
for spec in (sys.argv[1:] or [""]):
    os.environ.clear()
    os.environ.update(base_env)
    for kv in filter(None, spec.split(",")):
        a, b = kv.split("=")
        os.environ[a] = b
    user_streams = os.environ.get("TFX_OLS_STREAMS")
    os.environ["TFX_OLS_STREAMS"] = "1"
    w1, prof = timed(lambda: E.fft_conv_forward(x, k, (K - 1, 0)), reps=3, warm=1)
    if user_streams is None:
        del os.environ["TFX_OLS_STREAMS"]
    else:
        os.environ["TFX_OLS_STREAMS"] = user_streams
    w2, _ = timed(lambda: E.fft_conv_forward(x, k, (K - 1, 0)), reps=3, warm=1)
    print(f"[{spec or 'defaults'}] 1-stream wall {w1:7.3f} ms  " + " ".join(f"{n.replace('_kernel', '')}={v:.2f}" for n, v in prof.items())
          + f" | default-streams wall {w2:7.3f} ms = {C * T / w2 / 1e3:9.1f} Msamp/s", flush=True)
