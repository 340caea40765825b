#!/usr/bin/env python3
"""Determinism soak (development tool, GPU box): every hot-path entry point is run N times on the same input and must return the
same bits every time -- the overlap-save pass runs its slabs on internal streams and reuses its workspaces, the cascade kernel
splits rows into segments, the fused chunk kernel carries state: a race between them would show up as a sporadic mismatch.
Then the same from two host threads on two HIP streams at once (workspaces are per stream and device).

usage: soak.py [iterations, default 200]"""
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from torchfx_amd import Wave, torchfx_ext as E
from torchfx_amd.filter import FIR, LoButterworth, ParametricEQ

FS = 48000


def digest(t: torch.Tensor) -> torch.Tensor:
    """Position-sensitive checksum as a DEVICE scalar: no host sync between the calls under test, so consecutive calls overlap on the
    device exactly as they do in a pipeline."""
    v = t.contiguous().view(torch.int32 if t.element_size() == 4 else torch.int64)
    acc = torch.zeros((), dtype=torch.int64, device=t.device)
    step = 1 << 26
    flat = v.reshape(-1)
    for lo in range(0, flat.numel(), step):
        part = flat[lo:lo + step].to(torch.int64)
        part *= torch.arange(lo, lo + part.numel(), device=t.device, dtype=torch.int64) * 2 + 1      # wraps mod 2^64
        acc += part.sum()
    return acc


def mismatches(fn, n: int) -> int:
    ds = torch.stack([digest(fn()) for _ in range(n + 1)])
    return int((ds[1:] != ds[0]).sum().item())


def cases(dev, seconds_long=300.0):
    g = torch.Generator(device=dev).manual_seed(7)
    x_long = torch.randn(64, int(seconds_long * FS), generator=g, device=dev) * 0.2
    g = torch.Generator().manual_seed(7)
    x_mid = (torch.randn(64, 10 * FS, generator=g) * 0.2).to(dev)
    rev = np.random.default_rng(0).standard_normal(65536) * np.exp(-np.arange(65536) / 8000.0)
    rev = (rev / np.abs(rev).sum()).astype(np.float32)
    from scipy.signal import butter, firwin
    sos = torch.from_numpy(np.vstack([butter(6, 2000 / 24000, output="sos"), [[1.0089, -1.9636, 0.9695, 1, -1.9636, 0.9784]]]))
    b1024 = firwin(1024, 5000, fs=FS).astype(np.float32)

    def chain():
        w = Wave(x_long, FS, device=dev) | LoButterworth(2000, order=6) | ParametricEQ(1000, 2.0, 3.0) | FIR(b1024) | FIR(rev)
        return w.ys
    yield "chain .ys (default: recursion inside pass A, 64 ch)", chain

    def chain_fold():
        w = Wave(x_long, FS, device=dev)
        w.fuse_spectral = True
        return (w | LoButterworth(2000, order=6) | ParametricEQ(1000, 2.0, 3.0) | FIR(b1024) | FIR(rev)).ys
    yield "chain .ys (spectral fold, opt-in)", chain_fold
    k20 = np.ascontiguousarray(rev[:20000][::-1])
    yield "fused cascade|FIR op, 2^20-point blocks forced", lambda: E.sos_fft_conv_forward(x_long[:32], sos, k20, (19999, 0), force_block=1)
    yield "fft_conv 65536 taps", lambda: E.fft_conv_forward(x_long, rev[::-1].copy(), (65535, 0))
    yield "cascade f64", lambda: E.sos_forward(x_mid, None, sos, None, None)[0]
    yield "cascade f32", lambda: E.sos_forward(x_mid, None, sos, None, None, precision="f32")[0]
    yield "direct FIR 1024", lambda: E.fir_direct_forward(x_mid, b1024[::-1].copy())
    k1024 = b1024[::-1].copy()
    yield "FIR 1024 through the FFT mode (one launch, transform in LDS)", lambda: E.fft_conv_forward(x_mid, k1024, (1023, 0))
    k4096 = (np.random.default_rng(1).standard_normal(4096) / 4096).astype(np.float32)
    x_short = x_mid[:8, :44100].contiguous()
    k8192 = (np.random.default_rng(2).standard_normal(8192) / 8192).astype(np.float32)
    k12k = (np.random.default_rng(3).standard_normal(12000) / 12000).astype(np.float32)
    yield "4096 taps on short rows (8192-point one-launch kernel)", lambda: E.fft_conv_forward(x_short, k4096, (4095, 0))
    yield "4096 taps on long rows (16 384 points, radix 4 around 4096)", lambda: E.fft_conv_forward(x_mid, k4096, (4095, 0))
    k3000 = k4096[:3000].copy()
    yield "3000 taps on long rows (8192-point one-launch kernel)", lambda: E.fft_conv_forward(x_mid, k3000, (2999, 0))
    yield "8192 taps on short rows (16 384 points, 1024-thread workgroup)", lambda: E.fft_conv_forward(x_short, k8192, (8191, 0))
    yield "8192 taps on long rows (16 384 points, radix 4 around 4096)", lambda: E.fft_conv_forward(x_mid, k8192, (8191, 0))
    yield "12000 taps on long rows (three passes, 256-point rows)", lambda: E.fft_conv_forward(x_mid, k12k, (11999, 0))
    x64 = x_mid[:16].double()
    yield "FIR 1024 float64 (LDS)", lambda: E.fft_conv_forward(x64, k1024.astype(np.float64), (1023, 0))
    yield "4096 taps float64 (8192-point one-launch kernel)", lambda: E.fft_conv_forward(x64, k4096.astype(np.float64), (4095, 0))
    xc = x_mid[:2, :512].contiguous()
    taps = b1024[:256][::-1].copy()

    def chunk():
        sx = sy = h = None
        out = []
        for _ in range(8):
            y, sx, sy, h = E.chunk_forward(xc, sos, sx, sy, taps, h, 0.7, True)
            out.append(y)
        return torch.cat(out, dim=1)
    yield "fused chunk kernel, 8 chunks with carried state", chunk


def main() -> None:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    dev = torch.device("cuda:0")
    bad = 0
    for name, fn in cases(dev):
        miss = mismatches(fn, n)
        bad += miss
        print(f"{name:52s} {n} runs, {miss} mismatches", flush=True)

    # two host threads, two HIP streams, same work at the same time
    results: dict = {}

    def worker(tag: int) -> None:
        s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s):
            for name, fn in cases(dev, seconds_long=120.0):
                ref = digest(fn())
                results[(tag, name)] = (int(ref.item()), mismatches(fn, max(10, n // 5)))
    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for name in sorted({k[1] for k in results}):
        a, b = results[(0, name)], results[(1, name)]
        ok = a[0] == b[0] and a[1] == 0 and b[1] == 0
        bad += 0 if ok else 1
        print(f"two threads / two streams: {name:40s} {'same bits on both, no mismatch' if ok else 'MISMATCH ' + str((a, b))}", flush=True)
    print("SOAK", "FAILED" if bad else "OK")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
