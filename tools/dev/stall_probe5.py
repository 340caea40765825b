import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dev = torch.device("cuda:0")
def sync(): torch.cuda.synchronize(dev)
from torchfx_amd import torchfx_ext as E
from torchfx_amd import filter as F
import bench
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
ir = bench.reverb_ir()
x64 = torch.randn(16, 4_000_000, device=dev, dtype=torch.float64)
fir = F.FIR(ir)
loop("float64 signal, FIR module 65536 taps (rocFFT path)", lambda: fir(x64))
x = torch.randn(64, 2_880_000, device=dev)
from scipy.signal import firwin
fd = F.FIR(firwin(1024, 0.2), conv_mode="direct")
loop("direct FIR 1024", lambda: fd(x))
big = F.FIR(np.random.default_rng(1).standard_normal(40000) / 200, conv_mode="direct")
xs = torch.randn(4, 400_000, device=dev)
loop("direct FIR 40000 taps (long key)", lambda: big(xs))
