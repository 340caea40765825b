"""cfg-4 step: eager launches against a HIP-graph replay of the same step (development: are the ~360 launch boundaries visible?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torchfx_amd import torchfx_ext as E
C, T, K = 64, 28_800_000, 68977
x = torch.randn(C, T, device="cuda:0")
ir = np.random.default_rng(0).standard_normal(K) * np.exp(-np.arange(K) / 8000.0)
k = (ir / np.abs(ir).sum()).astype(np.float32)[::-1].copy()
def eager(n=8):
    y = E.fft_conv_forward(x, k, (K - 1, 0)); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        y = None
        y = E.fft_conv_forward(x, k, (K - 1, 0))
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("eager", [round(eager(), 3) for _ in range(3)], flush=True)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(2): y = E.fft_conv_forward(x, k, (K - 1, 0))
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    yg = E.fft_conv_forward(x, k, (K - 1, 0))
torch.cuda.synchronize()
def replay(n=8):
    g.replay(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("graph", [round(replay(), 3) for _ in range(3)], flush=True)
print("eager", [round(eager(), 3) for _ in range(3)], flush=True)
ref = E.fft_conv_forward(x, k, (K - 1, 0))
print("same result", bool(torch.equal(ref, yg)))
