"""TEST INFRASTRUCTURE (bench.py's cpu_baseline leg only): time the REFERENCE's own compiled CPU
IIR kernel -- oracle/_ref/torchfx_ext.so, built by `make -C oracle ref` from the reference's
sources where they lie (binding.cpp + cpu/iir_cpu.cpp) -- on a synthetic signal.

Run as a subprocess (`python oracle/ref_time.py CHANNELS SECONDS SOS.npy`) so that OMP_NUM_THREADS is
in force when the OpenMP runtime starts and so that the extension's -ffast-math FTZ/DAZ mode does
not leak into the caller.  Mirrors the call sequence of the reference's host code for this path:
`x.to(float64)` -> `torchfx_ext.sos_forward(x, sos, sos_cpu, state_x, state_y)` -> `.to(x.dtype)`
(src/torchfx/_ops.py:119-176, src/torchfx/filter/iir.py:176).  Prints one JSON line."""
import importlib.util
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
FS = 48000


def main() -> None:
    channels, seconds = int(sys.argv[1]), float(sys.argv[2])
    so = os.path.join(HERE, "_ref", "torchfx_ext.so")
    if not os.path.exists(so):
        print(json.dumps({"error": "oracle/_ref/torchfx_ext.so not built"}))
        return
    spec = importlib.util.spec_from_file_location("torchfx_ext", so)
    ext = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ext)
    torch.set_num_threads(max(1, int(os.environ.get("OMP_NUM_THREADS", "1") or 1)))
    sos = torch.from_numpy(np.load(sys.argv[3])) if len(sys.argv) > 3 else None
    if sos is None:
        raise SystemExit("usage: ref_time.py CHANNELS SECONDS SOS.npy")
    T = int(seconds * FS)
    g = np.random.default_rng(7)
    uniq = min(channels, 8)
    x = g.standard_normal((uniq, T)).astype(np.float32)
    x /= np.abs(x).max()
    x = torch.from_numpy(np.ascontiguousarray(np.tile(x, (-(-channels // uniq), 1))[:channels]))
    K = sos.shape[0]
    t0 = time.perf_counter()
    x64 = x.to(torch.float64)
    sx = torch.zeros(K, channels, 2, dtype=torch.float64)
    sy = torch.zeros(K, channels, 2, dtype=torch.float64)
    y, _, _ = ext.sos_forward(x64, sos, sos, sx, sy)
    y = y.to(x.dtype)
    dt = time.perf_counter() - t0
    print(json.dumps({"value": round(channels * T / dt / 1e6, 3), "seconds": round(dt, 3),
                      "threads": int(os.environ.get("OMP_NUM_THREADS", "0") or 0)}))


if __name__ == "__main__":
    main()
