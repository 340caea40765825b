import os, sys, time, json, ctypes
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
loop("torch mul (allocating 7.4 GB per step)", lambda: x * 1.5)
y = torch.empty_like(x)
loop("torch mul into a preallocated output", lambda: torch.mul(x, 1.5, out=y))
from torchfx_amd import torchfx_ext as E
import bench
k = bench.reverb_ir()[::-1].copy()
kt = torch.from_numpy(k)
loop("fft_conv_forward (allocating)", lambda: E.fft_conv_forward(x, kt, (65535, 0)))
os.environ["TFX_OLS_STREAMS"] = "1"
loop("fft_conv_forward, one internal stream", lambda: E.fft_conv_forward(x, kt, (65535, 0)))
del os.environ["TFX_OLS_STREAMS"]
from torchfx_amd import _lib
lib = _lib.load()
kh = np.ascontiguousarray(k, dtype=np.float32)
def capi():
    rc = lib.tfx_fft_conv_forward(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), 0, C, T, kh.ctypes.data_as(ctypes.c_void_p), 65536, 65535, 0,
                                  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, lib.tfx_last_error()
    return None
try:
    loop("tfx_fft_conv_forward through the C ABI into a preallocated output", capi)
except Exception as e:
    print("capi path failed:", e)
