import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np, json
import bench
from torchfx_amd import _lib
lib = _lib.load()
x = torch.randn(64, 28_800_000, device="cuda"); x.mul_(1.0/float(x.abs().max()))
def run(name, n=4, prof=False):
    step, desc, _ = bench.make_step(name, x)
    if prof: lib.tfx_prof_enable(1)
    for i in range(n):
        torch.cuda.synchronize(); t0=time.perf_counter()
        y = step()
        t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
        print(f"{name} step {i}: enqueue {1e3*(t1-t0):8.3f} ms  total {1e3*(t2-t0):8.3f} ms", flush=True)
    if prof:
        print(lib.tfx_prof_collect().decode()); lib.tfx_prof_enable(0)
run("chain"); run("chain_iir_kernel"); run("chain_iir_kernel", prof=True); run("chain_reference_staging")
xs = x[:, :2_880_000].contiguous()
step, _, _ = bench.make_step("sos", xs)
for prof in (0, 1):
    lib.tfx_prof_enable(prof)
    for i in range(4):
        torch.cuda.synchronize(); t0=time.perf_counter(); y = step(); t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
        print(f"sos prof={prof} step {i}: enqueue {1e3*(t1-t0):8.3f} ms  total {1e3*(t2-t0):8.3f} ms", flush=True)
    if prof: print(lib.tfx_prof_collect().decode())
