import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dev = torch.device("cuda:0")
C, T = 64, 600 * 48000
x = torch.randn(C, T, device=dev)
def sync(): torch.cuda.synchronize(dev)
from torchfx_amd import torchfx_ext as E, native
import bench
ir = bench.reverb_ir()
k64 = torch.from_numpy(ir[::-1].copy()).double().reshape(1, 1, -1)
ops = native.ops()
def loop(name, fn, n=18):
    o = None
    for _ in range(3):
        o = None; o = fn()
    sync()
    rows = []
    for i in range(n):
        sync(); t0 = time.perf_counter()
        o = None
        o = fn()
        t1 = time.perf_counter()
        sync(); t2 = time.perf_counter()
        rows.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
    print(name, " ".join(f"{a:.1f}/{b:.1f}" for a, b in rows), flush=True)
loop("float64 taps through E.fft_conv_forward (view each call)", lambda: E.fft_conv_forward(x, k64.reshape(-1), (65535, 0)))
loop("float64 taps, same tensor object", lambda: E.fft_conv_forward(x, k64, (65535, 0)))
k32 = k64.reshape(-1).float()
loop("fresh float32 copy per call, handed to the op directly", lambda: ops.fft_conv_forward(x, k64.reshape(-1).float(), 65535, 0))
keep = []
def leak():
    k = k64.reshape(-1).float(); keep.append(k)
    return ops.fft_conv_forward(x, k, 65535, 0)
loop("fresh float32 copy per call, never freed", leak)
loop("one float32 copy reused", lambda: ops.fft_conv_forward(x, k32, 65535, 0))
