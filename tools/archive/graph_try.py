"""Developer check: can a whole pipeline step be captured into a HIP graph (torch.cuda.CUDAGraph)?"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchfx_amd import torchfx_ext as E  # noqa: E402
from scipy.signal import butter, firwin  # noqa: E402

dev = "cuda:0"
sos = torch.from_numpy(butter(6, 2000 / 24000, output="sos"))
k = firwin(1024, 5000, fs=48000).astype(np.float32)[::-1].copy()
kl = (np.random.default_rng(0).standard_normal(65536) / 300).astype(np.float32)
for C, T in ((2, 4096), (8, 1 << 20)):
    x = torch.randn(C, T, device=dev)

    def step(inp):
        y, _, _ = E.sos_forward(inp, None, sos, None, None)
        y = E.fft_conv_forward(y, k, (1023, 0))
        if T >= (1 << 20):
            y = E.fft_conv_forward(y, kl, (65535, 0))
        return E.gain_forward(y, 0.5, True)

    for _ in range(3):
        ref = step(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    static_x = x.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step(static_x)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g, stream=s):
        out = step(static_x)
    g.replay()
    torch.cuda.synchronize()
    print(f"[{C}x{T}] capture ok, max diff vs eager {float((out - ref).abs().max()):.3e}")
    x2 = torch.randn(C, T, device=dev)
    static_x.copy_(x2)
    g.replay()
    torch.cuda.synchronize()
    print("   new input diff", float((out - step(x2)).abs().max()))
    t0 = time.perf_counter()
    for _ in range(200):
        g.replay()
    torch.cuda.synchronize()
    tg = (time.perf_counter() - t0) / 200 * 1e6
    t0 = time.perf_counter()
    for _ in range(200):
        step(static_x)
    torch.cuda.synchronize()
    te = (time.perf_counter() - t0) / 200 * 1e6
    print(f"   graph replay {tg:.1f} us vs eager {te:.1f} us per step")
