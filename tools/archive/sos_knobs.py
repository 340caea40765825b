"""Developer sweep: the cascade kernel (cfg-2 filter) under sets of env knobs, float64 and float32 arithmetic, 64 x 2.88 M (cfg 2)
and 64 x 28.8 M.  usage: python tools/sos_knobs.py "A=1,B=2" "A=0" ...   ('' = defaults)."""
import os
os.environ.setdefault("TFX_ENV_DYNAMIC", "1")      # this tool flips TFX_* knobs inside one process
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from torchfx_amd import torchfx_ext as E  # noqa: E402
from tools.quick_bench import timed  # noqa: E402

f1, f2, _, _ = bench.build_filters()
sos = torch.cat([f1._sos, f2._sos]).contiguous()
base_env = dict(os.environ)
xs = {T: torch.randn(64, T, device="cuda:0") for T in (2_880_000, 28_800_000)}
for spec in (sys.argv[1:] or [""]):
    os.environ.clear()
    os.environ.update(base_env)
    for kv in filter(None, spec.split(",")):
        a, b = kv.split("=")
        os.environ[a] = b
    out = []
    for T, x in xs.items():
        for prec in ("f64", "f32"):
            wall, prof = timed(lambda: E.sos_forward(x, None, sos, None, None, precision=prec), reps=10, warm=3)
            ms = list(prof.values())[0]
            out.append(f"{T // 1000}k {prec}: {ms:.4f} ms {8 * 64 * T / ms / 1e9:.2f} TB/s")
    print(f"[{spec or 'defaults'}] " + " | ".join(out), flush=True)
