"""Development: wall-clock marks inside the first (Wave(x) | f1 | f2 | fir | rev).ys of a process (no profiler)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
t00 = time.perf_counter()
import bench
from torchfx_amd import Wave, torchfx_ext as E, _lib
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
_lib.load(); E.prewarm(dev)
x = torch.randn(64, 600 * 48000, device=dev)
x.mul_(1.0 / float(x.abs().max()))
torch.cuda.synchronize()
f1, f2, fir, rev = bench.build_filters()
marks = []
t0 = time.perf_counter()
w = Wave(x, 48000, device=dev) | f1 | f2 | fir | rev
marks.append(("pipes", time.perf_counter()))
plan = w.plan()
marks.append(("plan()", time.perf_counter()))
y = torch.empty_like(x)
torch.cuda.synchronize()
marks.append(("torch.empty_like + sync", time.perf_counter()))
del y
out = plan[0](x)
marks.append(("forward enqueue", time.perf_counter()))
torch.cuda.synchronize()
marks.append(("device done", time.perf_counter()))
prev = t0
print(" | ".join(f"{n} {(t - prev) * 1e3:.1f}" + ("" if (prev := t) else "") for n, t in marks), f"| total {(marks[-1][1] - t0) * 1e3:.1f} ms")
