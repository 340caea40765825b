"""Developer tool: StreamProcessor throughput at small chunks, eager vs HIP-graph replay."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchfx_amd import effect as E  # noqa: E402
from torchfx_amd import filter as F  # noqa: E402
from torchfx_amd.realtime import StatefulFIR, StreamProcessor  # noqa: E402


def make():
    taps = (np.random.default_rng(4).standard_normal(301) / 30).tolist()
    return [F.LoButterworth(3000, order=4, fs=48000), F.ParametricEQ(frequency=800, q=1.0, gain=-3.0, fs=48000),
            StatefulFIR(taps, "fft"), E.Gain(0.8, clamp=True)]


print("StreamProcessor: LoButterworth-4 | ParametricEQ | StatefulFIR-301 (fft) | Gain(clamp), per-chunk latency")
for C, chunk in ((2, 512), (2, 1024), (2, 2048), (2, 4096), (2, 8192), (2, 16384), (2, 32768), (2, 65536), (16, 4096), (64, 4096)):
    x = torch.randn(C, chunk * 400, device="cuda:0")
    res = {}
    for fuse, g in (("0", False), ("0", True), ("1", False), ("1", True)):
        os.environ["TORCHFX_AMD_FUSE_CHUNK"] = fuse
        sp = StreamProcessor(make(), chunk_size=chunk, device="cuda:0", use_graph=g)
        sp.process_tensor(x[:, : chunk * 20], 48000)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        y = sp.process_tensor(x, 48000)
        torch.cuda.synchronize()
        res[fuse, g] = (time.perf_counter() - t0) / 400 * 1e6
    print(f"[{C} x {chunk}] staged: eager {res['0', False]:6.1f} graph {res['0', True]:6.1f} | one fused launch: eager "
          f"{res['1', False]:6.1f} graph {res['1', True]:6.1f} us/chunk   real-time factor at 48 kHz: "
          f"{chunk / 48000 * 1e6 / min(res.values()):.0f}x", flush=True)


# ---- RealtimeProcessor: round trip of one backend callback (host block in -> host block out) ----------------
from torchfx_amd.realtime import RealtimeProcessor, StreamConfig  # noqa: E402


class _Backend:
    def open_stream(self, config, callback=None):
        self.config, self.callback = config, callback

    def start(self): pass

    def stop(self): pass

    def close(self): pass


print("RealtimeProcessor callback (pinned staging in, chain, pinned staging out, one host wait), same chain, 2 channels")
for B in (128, 256, 512, 1024, 2048, 4096):
    res = {}
    for fuse, g in (("0", False), ("1", False), ("1", True)):
        os.environ["TORCHFX_AMD_FUSE_CHUNK"] = fuse
        be = _Backend()
        cfg = StreamConfig(sample_rate=48000, buffer_size=B, channels_in=2, channels_out=2)
        with RealtimeProcessor(make(), be, cfg, device="cuda:0", use_graph=g):
            xin, out = torch.randn(2, B), torch.zeros(2, B)
            for _ in range(30):
                be.callback(xin, out, B)
            ts = []
            for _ in range(300):
                t0 = time.perf_counter()
                be.callback(xin, out, B)
                ts.append(time.perf_counter() - t0)
        ts = np.array(ts) * 1e6
        res[fuse, g] = (np.median(ts), np.percentile(ts, 99))
    print(f"[2 x {B}] staged eager median {res['0', False][0]:6.1f} us p99 {res['0', False][1]:6.1f} | fused eager {res['1', False][0]:6.1f} "
          f"p99 {res['1', False][1]:6.1f} | fused graph {res['1', True][0]:6.1f} p99 {res['1', True][1]:6.1f}"
          f"   buffer period at 48 kHz {B / 48000 * 1e6:7.0f} us", flush=True)
