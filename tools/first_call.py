"""Developer probe: where the FIRST (Wave(x) | f1 | f2 | fir | rev).ys of a process spends its host time (plan build, host tap
conversion, spectrum, tables, workspaces, kernels)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from torchfx_amd import Wave

FS = 48000
x = torch.randn(64, 600 * FS, device="cuda:0")
torch.cuda.synchronize()
f1, f2, fir, rev = bench.build_filters()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
y = (Wave(x, FS, device=x.device) | f1 | f2 | fir | rev).ys
torch.cuda.synchronize()
pr.disable()
print(f"first .ys: {(time.perf_counter() - t0) * 1e3:.1f} ms")
pstats.Stats(pr).sort_stats("cumtime").print_stats(18)
t0 = time.perf_counter()
y = (Wave(x, FS, device=x.device) | f1 | f2 | fir | rev).ys
torch.cuda.synchronize()
print(f"second .ys: {(time.perf_counter() - t0) * 1e3:.1f} ms")
