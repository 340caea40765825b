"""Developer micro-benchmark (not the driver's bench.py): times individual ops with the
library's own HIP-event profiler, sweeping the tuning knobs exposed as env vars."""
import ctypes
import json
import os
os.environ.setdefault("TFX_ENV_DYNAMIC", "1")      # this tool flips TFX_* knobs inside one process
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchfx_amd import _lib, torchfx_ext as E  # noqa: E402

lib = _lib.load()


def timed(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    lib.tfx_prof_enable(1)
    lib.tfx_prof_collect()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e3
    prof = json.loads(lib.tfx_prof_collect().decode())
    lib.tfx_prof_enable(0)
    return wall, {k: v["total_ms"] / reps for k, v in prof.items()}


def main():
    which = sys.argv[1:] or ["sos", "fir", "fft"]
    dev = "cuda:0"
    from scipy.signal import butter, firwin
    if "sos" in which:
        C, T = 64, 2_880_000
        x = torch.randn(C, T, device=dev)
        sos = np.vstack([butter(6, 2000 / 24000, output="sos"),
                         np.array([[1.0089, -1.9636, 0.9695, 1, -1.9636, 0.9784]])])
        sos_t = torch.from_numpy(sos)
        print("plan", E.sos_plan_info(sos))
        for prec in ("f64", "f32"):
            for var in (0, 1, 2, 3, 4, 5):
                for wpc in (0, 8):
                    os.environ["TFX_SOS_VARIANT"] = str(var)
                    os.environ["TFX_SOS_WAVES_PER_CU"] = str(wpc)
                    wall, prof = timed(lambda: E.sos_forward(x, None, sos_t, None, None, precision=prec))
                    ms = list(prof.values())[0]
                    print(f"sos {prec} var={var} waves/cu={wpc:2d}: kernel {ms:7.3f} ms  wall {wall:7.3f} ms  "
                          f"{C * T / ms / 1e3:9.1f} Msamp/s  {8 * C * T / ms / 1e9:6.2f} TB/s ({8 * C * T / ms / 1e9 / 8 * 100:5.1f}% of 8 TB/s)",
                          flush=True)
        del x
    if "fir" in which:
        C, T = 64, 2_880_000
        x = torch.randn(C, T, device=dev)
        k = firwin(1024, 5000, fs=48000).astype(np.float32)[::-1].copy()
        wall, prof = timed(lambda: E.fir_direct_forward(x, k), reps=3, warm=1)
        ms = list(prof.values())[0]
        print(f"fir direct 1024: kernel {ms:.3f} ms wall {wall:.3f}  {C * T / ms / 1e3:.1f} Msamp/s  "
              f"{2 * 1024 * C * T / ms / 1e9:.1f} TFLOP/s ({2 * 1024 * C * T / ms / 1e9 / 157.3 * 100:.1f}% of 157.3)", flush=True)
        wall, prof = timed(lambda: E.fft_conv_forward(x, k, (1023, 0)), reps=3, warm=1)
        print(f"fir fft 1024: wall {wall:.3f} ms  {C * T / wall / 1e3:.1f} Msamp/s", prof, flush=True)
        del x
    if "lds" in which:
        from scipy.signal import firwin as fw
        C, T = 64, 2_880_000
        x = torch.randn(C, T, device=dev)
        for K in (64, 256, 1024, 2048):
            k = fw(K, 5000, fs=48000).astype(np.float32)[::-1].copy()
            for lds in ("1", "0"):
                os.environ["TFX_OLS_LDS"] = lds
                wall, prof = timed(lambda: E.fft_conv_forward(x, k, (K - 1, 0)), reps=20, warm=3)
                tot = sum(prof.values())
                print(f"fir fft K={K:5d} lds={lds}: wall {wall:7.3f} ms kernels {tot:7.3f} ms {8 * C * T / wall / 1e9:5.2f} TB/s "
                      f"({8 * C * T / wall / 1e9 / 8 * 100:5.1f}% of 8 TB/s) " + " ".join(f"{n.replace('_kernel', '')}={v:.3f}" for n, v in prof.items()), flush=True)
        os.environ["TFX_OLS_LDS"] = "1"
        xd = x.double()[:32]
        k = fw(1024, 5000, fs=48000).astype(np.float32).astype(np.float64)[::-1].copy()
        wall, prof = timed(lambda: E.fft_conv_forward(xd, k, (1023, 0)), reps=10, warm=2)
        print(f"fir fft K=1024 float64 32 rows: wall {wall:7.3f} ms {16 * 32 * T / wall / 1e9:5.2f} TB/s of 16 B/sample", prof, flush=True)
        del x, xd
    if "efx" in which:
        C, T = 64, 28_800_000
        x = torch.randn(C, T, device=dev)
        for name, fn, nbytes in (("gain", lambda: E.gain_forward(x, 0.5, False), 8), ("gain+clamp", lambda: E.gain_forward(x, 0.5, True), 8),
                                 ("stat absmax", lambda: E.stat_forward(x, E.STAT_ABSMAX, True), 4),
                                 ("stat rms", lambda: E.stat_forward(x, E.STAT_RMS, False), 4),
                                 ("normalize peak", lambda: E.normalize_forward(x, 1.0, E.STAT_ABSMAX, False), 12),
                                 ("normalize per-channel", lambda: E.normalize_forward(x, 1.0, E.STAT_ABSMAX, True), 12),
                                 ("normalize rms", lambda: E.normalize_forward(x, 1.0, E.STAT_RMS, False), 12)):
            wall, prof = timed(fn, reps=5, warm=2)
            tot = sum(prof.values())
            print(f"{name:22s}: kernels {tot:7.3f} ms wall {wall:7.3f} ms  {nbytes * C * T / tot / 1e9:5.2f} TB/s of its {nbytes} B/sample  "
                  + " ".join(f"{n.replace('_kernel', '')}={v:.3f}" for n, v in prof.items()), flush=True)
        xs = [torch.randn(C, T // 4, device=dev) for _ in range(8)]
        wall, prof = timed(lambda: E.sum_forward(xs), reps=5, warm=2)
        tot = sum(prof.values())
        print(f"sum of 8 branches     : kernels {tot:7.3f} ms  {9 * 4 * C * (T // 4) / tot / 1e9:5.2f} TB/s (8 reads + 1 write)", flush=True)
        del x, xs
    if "plus" in which:
        from torchfx_amd import filter as F
        C, T = 64, 2_880_000
        x = torch.randn(C, T, device=dev)
        mk = lambda: [F.LoButterworth(800, order=4, fs=48000), F.HiButterworth(3000, order=4, fs=48000),
                      F.ParametricEQ(frequency=1000, q=2.0, gain=4.0, fs=48000)]
        a, b, c = mk()
        comb = a + b + c
        wall, prof = timed(lambda: comb(x), reps=5, warm=2)
        print(f"f1+f2+f3 one launch : wall {wall:7.3f} ms  {8 * C * T / wall / 1e9:5.2f} TB/s of 8 B/sample", prof, flush=True)
        fs_ = mk()
        wall, prof = timed(lambda: E.sum_forward([f(x) for f in fs_]), reps=5, warm=2)
        print(f"f1+f2+f3 staged     : wall {wall:7.3f} ms", prof, flush=True)
        del x
    if "io" in which:
        from torchfx_amd import io as tio
        for C, F in ((2, 28_800_000 * 4), (8, 28_800_000), (64, 28_800_000 // 4)):
            fr = torch.randn(F, C, device=dev)
            wall, prof = timed(lambda: E.deinterleave_forward(fr), reps=5, warm=2)
            t1 = sum(prof.values())
            pl = E.deinterleave_forward(fr)
            wall, prof = timed(lambda: E.interleave_forward(pl), reps=5, warm=2)
            t2 = sum(prof.values())
            pcm = torch.randint(-30000, 30000, (F, C), device=dev, dtype=torch.int16)
            wall, prof = timed(lambda: E.deinterleave_forward(pcm), reps=5, warm=2)
            t3 = sum(prof.values())
            print(f"layout C={C:3d} F={F}: deinterleave {t1:.3f} ms = {8 * C * F / t1 / 1e9:.2f} TB/s | interleave {t2:.3f} ms = "
                  f"{8 * C * F / t2 / 1e9:.2f} TB/s | pcm16->f32 {t3:.3f} ms = {6 * C * F / t3 / 1e9:.2f} TB/s", flush=True)
            del fr, pl, pcm
        host = np.random.default_rng(0).standard_normal((28_800_000, 2)).astype(np.float32)
        for dt in (np.float32, np.int16):
            h = host if dt == np.float32 else (host * 3000).astype(np.int16)
            tio.upload_interleaved(h, dev)
            t0 = time.perf_counter()
            tio.upload_interleaved(h, dev)
            torch.cuda.synchronize()
            dtm = time.perf_counter() - t0
            t0 = time.perf_counter()
            torch.from_numpy(np.ascontiguousarray(h.T)).to(dev)
            torch.cuda.synchronize()
            dtr = time.perf_counter() - t0
            print(f"upload 2 ch x 600 s {np.dtype(dt).name}: pipelined {dtm * 1e3:.1f} ms ({h.nbytes / dtm / 1e9:.1f} GB/s host->dev) "
                  f"vs host transpose + copy {dtr * 1e3:.1f} ms", flush=True)
    if "fft" in which:
        C, T = 64, 2_880_000 * (10 if "big" in which else 1)
        x = torch.randn(C, T, device=dev)
        K = 65536
        ir = np.random.default_rng(0).standard_normal(K) * np.exp(-np.arange(K) / 8000.0)
        k = (ir / np.abs(ir).sum()).astype(np.float32)[::-1].copy()
        for lg in (17, 18, 19, 20):
            for ws in (256, 2048, 16384):
                os.environ["TFX_FFT_LOG2N"] = str(lg)
                os.environ["TFX_FFT_WS_MB"] = str(ws)
                wall, prof = timed(lambda: E.fft_conv_forward(x, k, (K - 1, 0)), reps=2, warm=1)
                tot = sum(prof.values())
                print(f"fftconv 65536 log2N={lg} ws={ws:5d}MB: wall {wall:8.3f} ms kernels {tot:8.3f} ms  "
                      f"{C * T / wall / 1e3:9.1f} Msamp/s  {8 * C * T / wall / 1e9:5.2f} TB/s  "
                      + " ".join(f"{n.replace('_kernel', '')}={v:.2f}" for n, v in prof.items()), flush=True)


if __name__ == "__main__":
    main()
