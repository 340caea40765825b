"""Developer probe: host cost per call of the small-shape entry points (launch-bound regime of the reference's published benchmarks)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from torchfx_amd import Wave
from torchfx_amd import filter as F

FS = 44100
x = torch.randn(8, 60 * FS, device="cuda:0")
mods = [F.LoButterworth(2000, order=2, fs=FS), F.HiButterworth(100, order=2, fs=FS), F.ParametricEQ(1000, 2.0, 3.0, fs=FS), F.LoButterworth(8000, order=2, fs=FS)]


def staged():
    y = x
    for m in mods:
        y = m(y)
    return y


def piped():
    w = Wave(x, FS, device=x.device)
    for m in mods:
        w = w | m
    return w.ys


for name, fn in (("4 IIR modules one after the other", staged), ("the same through Wave |", piped)):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t_enq = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n * 1e6
    print(f"{name}: host enqueue {t_enq:.1f} us per call, with the device {t_all:.1f} us per call")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        fn()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
