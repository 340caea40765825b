"""Developer tool: StreamProcessor throughput at small chunks, eager vs HIP-graph replay."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchfx_amd import effect as E  # noqa: E402
from torchfx_amd import filter as F  # noqa: E402
from torchfx_amd.realtime import StatefulFIR, StreamProcessor  # noqa: E402


def make():
    taps = (np.random.default_rng(4).standard_normal(301) / 30).tolist()
    return [F.LoButterworth(3000, order=4, fs=48000), F.ParametricEQ(frequency=800, q=1.0, gain=-3.0, fs=48000),
            StatefulFIR(taps, "fft"), E.Gain(0.8, clamp=True)]


print("StreamProcessor: LoButterworth-4 | ParametricEQ | StatefulFIR-301 (fft) | Gain(clamp), per-chunk latency")
for C, chunk in ((2, 512), (2, 1024), (2, 2048), (2, 4096), (2, 8192), (2, 16384), (2, 32768), (2, 65536), (16, 4096), (64, 4096)):
    x = torch.randn(C, chunk * 400, device="cuda:0")
    res = {}
    for g in (False, True):
        sp = StreamProcessor(make(), chunk_size=chunk, device="cuda:0", use_graph=g)
        sp.process_tensor(x[:, : chunk * 20], 48000)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        y = sp.process_tensor(x, 48000)
        torch.cuda.synchronize()
        res[g] = (time.perf_counter() - t0) / 400 * 1e6
    print(f"[{C} x {chunk}] eager {res[False]:7.1f} us/chunk   graph {res[True]:7.1f} us/chunk   x{res[False] / res[True]:.2f}"
          f"   real-time factor at 48 kHz: {chunk / 48000 * 1e6 / res[True]:.0f}x", flush=True)
