import os, sys, time, gc
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dev = torch.device("cuda:0")
C, T = 64, 600 * 48000
x = torch.randn(C, T, device=dev)
def sync(): torch.cuda.synchronize(dev)
def loop(name, fn, n=24):
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
from torchfx_amd import torchfx_ext as E
from torchfx_amd import filter as F
import bench
ir = bench.reverb_ir()
kt = torch.from_numpy(ir[::-1].copy())
loop("E.fft_conv_forward 65536", lambda: E.fft_conv_forward(x, kt, (65535, 0)))
fir = F.FIR(ir)
loop("FIR module 65536", lambda: fir(x))
k2 = np.convolve(ir.astype(np.float64), np.hanning(3442) / 1721.0).astype(np.float32)
print("taps", k2.shape)
kt2 = torch.from_numpy(k2[::-1].copy())
loop("E.fft_conv_forward 68977", lambda: E.fft_conv_forward(x, kt2, (k2.size - 1, 0)))
fir2 = F.FIR(k2)
loop("FIR module 68977", lambda: fir2(x))
plan, names = bench.plan_chain(x)
print(names, [type(m).__name__ for m in plan], plan[0].kernel.dtype, plan[0].kernel.device, plan[0].kernel.shape)
loop("plan[0](x)", lambda: plan[0](x))
loop("run_plan", lambda: bench.run_plan(plan, x))
gc.disable()
loop("run_plan, gc disabled", lambda: bench.run_plan(plan, x))
