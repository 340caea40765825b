"""Where the chain step's wall time goes beyond the kernels: the same 68 977-tap pass through the raw op, the planned module and
the full (Wave(x) | f1 | f2 | fir | rev).ys, on one box; plus the host time to enqueue one step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torchfx_amd import Wave, torchfx_ext as E

x = torch.randn(64, 28_800_000, device="cuda:0")
f1, f2, fir, rev = bench.build_filters()
plan, names = bench.plan_chain(x)
merged = plan[0]
taps = merged.kernel.reshape(-1)
cases = {
    "raw op fft_conv_forward(68977 taps)": lambda: E.fft_conv_forward(x, taps, (taps.numel() - 1, 0)),
    "planned module forward": lambda: merged(x),
    "(Wave(x) | f1 | f2 | fir | rev).ys": lambda: (Wave(x, 48000, device=x.device) | f1 | f2 | fir | rev).ys,
}
for rep in range(2):
    for name, fn in cases.items():
        for _ in range(3):
            y = None; y = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            y = None
            y = fn()
        t_enq = (time.perf_counter() - t0) / 10 * 1e3
        torch.cuda.synchronize()
        t_all = (time.perf_counter() - t0) / 10 * 1e3
        print(f"{name:40s}: host enqueue {t_enq:.3f} ms per step, wall {t_all:.3f} ms per step", flush=True)
