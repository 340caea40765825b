"""(Wave | LoButterworth-6 | ParametricEQ | FIR-1024).ys on 64 x 2.88 M: folded into one overlap-save pass vs cascade + FIR (development)."""
import os, sys, time
os.environ.setdefault("TFX_ENV_DYNAMIC", "1")      # this tool flips TFX_* knobs inside one process
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torchfx_amd import Wave
x = torch.randn(64, 2_880_000, device="cuda:0")
def run(spectral):
    f1, f2, fir, rev = bench.build_filters()
    w = Wave(x, 48000, device=x.device)
    w.fuse_spectral = spectral
    plan = (w | f1 | f2 | fir).plan()
    names = [type(m).__name__ + (f"[{m.kernel.numel()}]" if getattr(m, "kernel", None) is not None else "") for m in plan]
    for _ in range(3): bench.run_plan(plan, x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): bench.run_plan(plan, x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20 * 1e3, names
for rep in range(3):
    for sp in (True, False):
        ms, names = run(sp)
        print(f"fuse_spectral={sp}: {ms:.4f} ms  {names}", flush=True)
