import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from torchfx_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
C, T = 64, 600 * 48000
x = torch.randn(C, T, device=dev); x.mul_(1.0 / float(x.abs().max()))
def sync(): torch.cuda.synchronize(dev)
for vname in ("chain_iir_kernel", "chain"):
    vstep, vdesc, _ = bench.make_step(vname, x)
    o = None
    for _ in range(3):
        o = None; o = vstep()
    sync()
    rows = []
    for i in range(40):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        sync(); t0 = time.perf_counter()
        e0.record()
        o = None
        o = vstep()
        t1 = time.perf_counter()
        e1.record()
        sync(); t2 = time.perf_counter()
        rows.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3, e0.elapsed_time(e1)))
    print(vname, "host enqueue ms / wall ms / gpu event ms")
    print(" ".join(f"{a:.1f}/{b:.1f}/{c:.1f}" for a, b, c in rows), flush=True)
    lib.tfx_prof_enable(1); lib.tfx_prof_collect()
    for i in range(10):
        o = None; o = vstep(); sync()
        p = json.loads(lib.tfx_prof_collect().decode())
        print(i, {k: round(v["total_ms"], 2) for k, v in p.items()})
    lib.tfx_prof_enable(0)
    del o, vstep
