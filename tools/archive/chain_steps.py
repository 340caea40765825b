"""Per-step wall times of the default bench step (development): looks for hiccups inside a run of back-to-back steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torchfx_amd import _lib, torchfx_ext as E
dev = torch.device("cuda:0")
_lib.load(); E.prewarm(dev)
x = bench.make_input(64, int(600 * bench.FS), dev, 0) if hasattr(bench, "make_input") else torch.randn(64, int(600 * bench.FS), device=dev)
step, desc, _ = bench.make_step("chain", x)
for _ in range(5): step()
torch.cuda.synchronize()
for rnd in range(3):
    ts = []
    t_all0 = time.perf_counter()
    for i in range(20):
        t0 = time.perf_counter(); y = step(); t1 = time.perf_counter()
        ts.append((t1 - t0) * 1e3)
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t_all0) * 1e3 / 20
    print(f"round {rnd}: {tot:.3f} ms/step; host enqueue ms per step: " + " ".join(f"{t:.1f}" for t in ts), flush=True)
