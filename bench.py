#!/usr/bin/env python3
"""bench.py -- the driver's benchmark contract for the torchfx.filter hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the hot path over one batch of synthetic audio already resident in
HBM.  Default workload = the configuration BASELINE.json's metric is quoted on: the fused pipe
chain  4-section SOS cascade | FIR-1024 | 65536-tap reverb IR  over 64 channels x 600 s @ 48 kHz
float32 PER GPU (configs[4] is 512 channels over 8 GPUs = 64 per GPU).  Other workloads
(`--workload sos|fir|fftconv` = configs[1..3]) are there for profiling; the parity tests cover
them.  Channels shard across ranks with no data-path collective (weak scaling); `--gather`
additionally times the final RCCL gather to rank 0 outside the timed region.

Prints ONE JSON line on rank 0.  `roofline` follows SURVEY.md 8(d): algorithmic bytes are
8 B per sample-channel (4 B read + 4 B written) for every stage and for the fused chain.
`cpu_baseline` times the oracle (our CPU restatement of the reference path, pinned by golden
vectors) on a bounded sample of the same workload -- a reported baseline, never a target.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FS = 48000
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_PEAK_TF = 157.3


def reverb_ir(K: int = 65536) -> np.ndarray:
    """SURVEY.md 8(d) cfg 4: seeded exponentially-decaying noise tail, L1-normalised, float32."""
    ir = np.random.default_rng(0).standard_normal(K) * np.exp(-np.arange(K) / 8000.0)
    return (ir / np.abs(ir).sum()).astype(np.float32)


def build_filters():
    from scipy.signal import firwin
    from torchfx_amd import filter as F

    f1 = F.LoButterworth(2000, order=6, fs=FS)                       # 3 sections
    f2 = F.ParametricEQ(frequency=1000, q=2.0, gain=3.0, fs=FS)      # 1 section  (examples/quick_start.py chain)
    fir = F.FIR(firwin(1024, 5000, fs=FS))                           # cfg 3 taps
    rev = F.FIR(reverb_ir())                                         # cfg 4 IR
    for f in (f1, f2):
        f.compute_coefficients()
    return f1, f2, fir, rev


def make_step(workload: str, x: torch.Tensor):
    """Returns (step_fn, description, n_stages_reference)."""
    from torchfx_amd import filter as F
    from torchfx_amd import torchfx_ext as E
    from torchfx_amd.wave import _merge_fir_run

    f1, f2, fir, rev = build_filters()
    sos = torch.cat([f1._sos, f2._sos]).contiguous()
    if workload == "sos":
        return (lambda: E.sos_forward(x, None, sos, None, None)[0]), "cfg2: fused 4-section SOS cascade", 1
    if workload == "fir":
        k = fir.kernel.reshape(-1)
        return (lambda: E.fir_direct_forward(x, k)), "cfg3: direct FIR, 1024 taps", 1
    if workload == "fftconv":
        k = rev.kernel.reshape(-1)
        return (lambda: E.fft_conv_forward(x, k, (k.numel() - 1, 0))), "cfg4: overlap-save FFT conv, 65536 taps", 1
    if workload == "chain":
        merged = _merge_fir_run([fir, rev])          # conv associativity: one 66559-tap overlap-save pass
        k = merged.kernel.reshape(-1).to(torch.float32)
        pad = (k.numel() - 1, 0)

        def step():
            y = E.sos_forward(x, None, sos, None, None)[0]
            return E.fft_conv_forward(y, k, pad)
        return step, "cfg5/GPU: fused chain 4xbiquad | FIR-1024 | FFT-conv-65536 (FIRs merged)", 3
    if workload == "chain_spectral":
        # opt-in planner mode (Wave.fuse_spectral): the stateless IIR cascade is folded into the
        # FIR pass as its truncated impulse response -> ONE overlap-save pass for the whole chain
        from torchfx_amd.wave import _iir_as_fir
        eq = _iir_as_fir([f1, f2])
        merged = _merge_fir_run([eq, fir, rev])
        k = merged.kernel.reshape(-1).to(torch.float32)
        pad = (k.numel() - 1, 0)
        return (lambda: E.fft_conv_forward(x, k, pad)), f"chain as one spectral pass ({k.numel()} taps)", 3
    raise SystemExit(f"unknown workload {workload}")


def cpu_baseline(workload: str, seconds: float, channels: int) -> dict:
    """Oracle on the host, bounded sample of the same workload: all cores it can use (one per
    channel, at most 32 -- `cores` is what actually ran) and, on a quarter of the sample, one thread."""
    from oracle import oracle as O

    f1, f2, fir, rev = build_filters()
    sos = np.vstack([f1._sos.numpy(), f2._sos.numpy()])
    g = np.random.default_rng(7)
    T = int(seconds * FS)
    uniq = min(channels, 8)                      # 8 distinct channels, tiled: only the timing matters here
    x = g.standard_normal((uniq, T)).astype(np.float32)
    x /= np.abs(x).max()
    x = np.ascontiguousarray(np.tile(x, (-(-channels // uniq), 1))[:channels])
    kf, kr = fir.kernel.numpy().reshape(-1), rev.kernel.numpy().reshape(-1)

    def fn(xx, nthr):
        return {
            "sos": lambda: O.iir_module_forward(xx, sos)[0],
            "fir": lambda: O.fir_direct(xx, kf),
            "fftconv": lambda: O.fir_forward(xx, kr, "fft", threads=nthr),
            "chain": lambda: O.chain_forward(xx, sos, [kf, kr], threads=nthr),
        }[workload]

    want = max(1, min(os.cpu_count() or 1, channels, 32))
    # explicit thread pinning: the OpenMP runtime is already up, environment variables are too late
    cores = min(want, O.set_threads(want))
    t0 = time.perf_counter()
    fn(x, cores)()
    dt = time.perf_counter() - t0
    x1 = x[: max(1, channels // 4)]
    O.set_threads(1)
    t0 = time.perf_counter()
    fn(x1, 1)()
    dt1 = time.perf_counter() - t0
    single = round(x1.shape[0] * T / dt1 / 1e6, 3)
    # the SciPy path north_star names (sosfilt / lfilter / fftconvolve, float64, single thread),
    # on an eighth of the same sample -- secondary figure, the oracle above is the reported baseline
    scipy_val = None
    try:
        from scipy import signal as sg
        xs = x[: max(1, channels // 8)].astype(np.float64)
        b1, b2 = kf[::-1].astype(np.float64), kr[::-1].astype(np.float64)
        sfn = {
            "sos": lambda: sg.sosfilt(sos, xs, axis=-1),
            "fir": lambda: sg.lfilter(b1, [1.0], xs[:, : T // 8], axis=-1),
            "fftconv": lambda: sg.fftconvolve(xs, b2[None], axes=-1)[:, :T],
            "chain": lambda: sg.fftconvolve(sg.fftconvolve(sg.sosfilt(sos, xs, axis=-1), b1[None], axes=-1)[:, :T],
                                            b2[None], axes=-1)[:, :T],
        }[workload]
        s0 = time.perf_counter()
        sfn()
        sdt = time.perf_counter() - s0
        nsamp = xs.shape[0] * (T // 8 if workload == "fir" else T)
        scipy_val = round(nsamp / sdt / 1e6, 3)
    except Exception:
        pass
    # the reference's own compiled CPU IIR kernel (oracle/_ref, built from the reference's sources
    # in the build container; the prebuilt .so travels to the GPU box) on the same sample
    ref_iir = None
    if workload in ("sos", "chain"):
        try:
            import subprocess
            import tempfile

            def run_ref(nthr):
                with tempfile.NamedTemporaryFile(suffix=".npy") as tf:
                    np.save(tf.name, sos)
                    env = dict(os.environ, OMP_NUM_THREADS=str(nthr))
                    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ref_time.py"), str(channels),
                                        str(seconds), tf.name], env=env, capture_output=True, text=True, timeout=300)
                return json.loads(r.stdout.strip().splitlines()[-1])

            ref_iir = run_ref(cores)                     # its OpenMP loop over channels, like the reference runs it
            if "value" in ref_iir and cores > 1:
                ref_iir["single_thread_value"] = run_ref(1).get("value")
        except Exception as e:
            ref_iir = {"error": repr(e)}
    if workload == "sos" and ref_iir and "value" in ref_iir:
        return {"value": ref_iir["value"], "unit": "Msamples/s", "cores": cores, "kind": "reference",
                "single_thread_value": ref_iir.get("single_thread_value"),
                "port_value": round(channels * T / dt / 1e6, 3), "port_single_thread_value": single,
                "scipy_value": scipy_val,
                "sample": f"{channels} ch x {seconds:g} s @ 48 kHz float32, the reference's own sos_forward_cpu "
                          f"(oracle/_ref/torchfx_ext.so, -O3 -ffast-math -fopenmp as its CMakeLists) incl. its float64 "
                          f"casts, {cores} OpenMP threads (one per channel), {ref_iir['seconds']:.2f} s; port_value = our C "
                          f"oracle, same sample and threads"}
    return {"value": round(channels * T / dt / 1e6, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "single_thread_value": single, "scipy_value": scipy_val, "reference_iir_stage": ref_iir,
            "sample": f"{channels} ch x {seconds:g} s @ 48 kHz float32, oracle (C float64 DF1 + numpy overlap-save, "
                      f"reference framing N=int(5K)), {cores} threads (one per channel), {dt:.2f} s; "
                      f"single_thread_value on {x1.shape[0]} ch, {dt1:.2f} s"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="chain", choices=["chain", "sos", "fir", "fftconv"])  # chain_spectral: internal
    ap.add_argument("--channels", type=int, default=64, help="channels PER GPU")
    ap.add_argument("--seconds", type=float, default=None, help="signal length (default: 600 chain/fftconv, 60 sos/fir)")
    ap.add_argument("--gather", action="store_true", help="also time the final RCCL gather to rank 0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only warm-up + timed steps (for rocprofv3 runs: no single-stream / variant passes)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    import torch.distributed as dist

    # TFX_BENCH_SHARE_DEVICE=1 (development only): every rank uses cuda:0 and the collectives go over
    # gloo, so the N > 1 control flow can be exercised on a one-GPU box; never set by the driver
    share = os.environ.get("TFX_BENCH_SHARE_DEVICE", "0") == "1"
    local_dev = 0 if share else local
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from torchfx_amd import _lib
    lib = _lib.load()                      # fails loudly if the HIP extension is missing

    seconds = args.seconds if args.seconds is not None else (600.0 if args.workload in ("chain", "fftconv") else 60.0)
    C, T = args.channels, int(seconds * FS)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.randn(C, T, device=dev, generator=gen, dtype=torch.float32)
    x.mul_(1.0 / float(x.abs().max()))     # max|x| <= 1 (benchmarks/conftest.py:70-82 of the reference)

    step, desc, _ = make_step(args.workload, x)

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        out = step()
    sync()
    lib.tfx_prof_enable(1)
    lib.tfx_prof_collect()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync()
    elapsed = time.perf_counter() - t0
    prof = json.loads(lib.tfx_prof_collect().decode())
    lib.tfx_prof_enable(0)
    if world > 1:
        t = torch.tensor([elapsed], device="cpu" if share else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel durations without overlap: the overlap-save passes of consecutive slabs run on two
    # internal streams in the timed region, which inflates each kernel's elapsed time; one extra,
    # untimed pass on a single stream gives the clean per-kernel numbers
    prof_serial = None
    if args.workload in ("chain", "fftconv") and not args.no_extras:
        old_env = os.environ.get("TFX_OLS_STREAMS")
        os.environ["TFX_OLS_STREAMS"] = "1"
        try:
            out = step()
            sync()
            lib.tfx_prof_enable(1)
            lib.tfx_prof_collect()
            for _ in range(2):
                out = step()
            sync()
            prof_serial = json.loads(lib.tfx_prof_collect().decode())
            lib.tfx_prof_enable(0)
        finally:
            if old_env is None:
                os.environ.pop("TFX_OLS_STREAMS", None)
            else:
                os.environ["TFX_OLS_STREAMS"] = old_env
    variant = None
    if args.workload == "chain" and rank == 0 and not args.no_extras:
        try:                                   # secondary figure, outside the timed region
            vstep, vdesc, _ = make_step("chain_spectral", x)
            vout = vstep()
            torch.cuda.synchronize(dev)
            v0 = time.perf_counter()
            for _ in range(3):
                vout = vstep()
            torch.cuda.synchronize(dev)
            vms = (time.perf_counter() - v0) / 3 * 1e3
            diff = float((vout[:, : 4 * FS] - out[:, : 4 * FS]).abs().max())
            variant = {"what": vdesc + "; opt-in (Wave.fuse_spectral), float32 FFT arithmetic for the IIR too",
                       "ms_per_step": round(vms, 4), "Msamples_per_s": round(C * T / vms / 1e3, 1),
                       "frac_of_8TBps_at_8B_per_sample": round(8.0 * C * T / (vms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                       "max_abs_diff_vs_default_chain_first_4s": diff}
            del vout
        except Exception as e:
            variant = {"error": repr(e)}
    gather_ms = None
    if args.gather and world > 1:
        bufs = [torch.empty_like(out) for _ in range(world)] if rank == 0 else None
        sync()
        g0 = time.perf_counter()
        dist.gather(out, bufs, dst=0)
        sync()
        gather_ms = (time.perf_counter() - g0) * 1e3

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        samples = C * T
        value = world * samples / (elapsed / args.steps) / 1e6            # whole-job Msamples/s
        kernels = {}
        for name, v in prof.items():
            calls_per_step = v["calls"] / args.steps
            kernels[name] = {"launches_per_step": round(calls_per_step, 2),
                             "avg_ms_per_launch": round(v["total_ms"] / v["calls"], 4),
                             "ms_per_step": round(v["total_ms"] / args.steps, 4)}
        # per-kernel HBM model: the bytes each launch group must move by design (not the 8 B/sample
        # algorithmic figure) -> achieved GB/s of that kernel; PMC-measured traffic agrees within
        # a few % (profiles/r01_traffic.json)
        try:
            from torchfx_amd import torchfx_ext as E
            model = {"sos_stream_kernel<f64>": 8.0 * samples, "sos_stream_kernel<f32>": 8.0 * samples}
            if args.workload in ("chain", "fftconv"):
                kk = 66559 if args.workload == "chain" else 65536
                info = E.ols_plan_info(kk, T, (kk - 1, 0))
                frames = C * info["F"]
                pairs = (frames + 1) // 2
                n = info["N"]
                for sfx in ("", "16"):
                    model["ols_col_fwd%s_kernel" % sfx] = frames * n * 4.0 + pairs * n * 8.0
                    model["ols_col_inv%s_kernel" % sfx] = pairs * n * 8.0 + 4.0 * samples
                for nm in ("ols_row_kernel", "ols_row1024_kernel", "ols_row4096_kernel"):
                    model[nm] = pairs * n * 16.0
                line_ols = {"fft_block": n, "hop": info["S"], "blocks_per_row": info["F"], "native_lds_fft": info["native"]}
            else:
                line_ols = None
            for name, k in kernels.items():
                if name in model and k["ms_per_step"] > 0:
                    k["model_GB_per_step"] = round(model[name] / 1e9, 3)
                    k["GBps"] = round(model[name] / 1e9 / (k["ms_per_step"] * 1e-3), 1)
                    k["frac_of_8TBps"] = round(k["GBps"] / HBM_PEAK_GBS, 4)
        except Exception:
            line_ols = None
        kernels_serial = None
        if prof_serial:
            kernels_serial = {}
            for name, v in prof_serial.items():
                ms = v["total_ms"] / 2
                kernels_serial[name] = {"ms_per_step": round(ms, 4), "avg_ms_per_launch": round(v["total_ms"] / v["calls"], 4)}
                if name in kernels and "model_GB_per_step" in kernels[name]:
                    kernels_serial[name]["GBps"] = round(kernels[name]["model_GB_per_step"] / (ms * 1e-3), 1)
        dom = max(kernels, key=lambda n: kernels[n]["ms_per_step"]) if kernels else None
        gpu_ms = sum(k["ms_per_step"] for k in kernels.values())
        # per-GPU algorithmic bytes of one step: 8 B per sample-channel
        alg_gb = 8.0 * samples / 1e9
        # roofline object = the DOMINANT kernel: algorithmic work of one launch / its average
        # duration in the timed region (HIP events on the launch stream).  The whole-step figure
        # (all passes of the step share the one 8 B/sample) is reported beside it as step_*.
        if args.workload == "fir":
            step_ach = 2.0 * 1024 * samples / (ms_step * 1e-3) / 1e12
            roof = {"bound": "mfma", "peak": FP32_PEAK_TF, "unit": "TFLOP/s", "traffic": None,
                    "note": "2*1024 flop/sample on exact-f32 MFMA (v_mfma_f32_32x32x2_f32); HBM roof unreachable (SURVEY 7.3-3)"}
            unit_work = 2.0 * 1024 / 1e12                                  # TFLOP per sample
        else:
            step_ach = alg_gb / (ms_step * 1e-3)
            roof = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None,
                    "note": "achieved = 8 B/sample-channel x samples one launch of the dominant kernel covers / its "
                            "average duration; step_achieved = 8 B x samples of the step / step time (all passes)"}
            unit_work = 8.0 / 1e9                                          # GB per sample
        if dom:
            per_launch = samples / max(kernels[dom]["launches_per_step"], 1e-9)
            ach = unit_work * per_launch / (kernels[dom]["avg_ms_per_launch"] * 1e-3)
        else:
            ach = step_ach
        roof["achieved"] = round(ach, 2)
        roof["frac"] = round(ach / roof["peak"], 4)
        roof["step_achieved"] = round(step_ach, 2)
        roof["step_frac"] = round(step_ach / roof["peak"], 4)
        # HBM bytes measured with rocprofv3 PMC passes of this same command (profiles/): counters
        # cannot be read from inside the process, so the last profiled values are reported --
        # `traffic` per launch of the dominant kernel (like `achieved`), `step_traffic` per step
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
            if args.workload in tr and C == 64 and seconds == 600.0:
                ent = tr[args.workload]
                roof["step_traffic"] = ent["bytes_per_step"]
                pk = ent.get("per_kernel_GB_per_launch", {}).get(dom)
                if pk:
                    roof["traffic"] = round((pk["read"] + pk["write"]) * 1e9)
                roof["traffic_source"] = "profiles/r01_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
        except Exception:
            pass
        roof["dominant_kernel"] = dom
        if dom:
            roof["dominant_kernel_avg_ms"] = kernels[dom]["avg_ms_per_launch"]
            roof["dominant_kernel_ms_per_step"] = kernels[dom]["ms_per_step"]
        if "sos_stream_kernel<f64>" in kernels or "sos_stream_kernel<f32>" in kernels:
            kn = "sos_stream_kernel<f64>" if "sos_stream_kernel<f64>" in kernels else "sos_stream_kernel<f32>"
            a = alg_gb / (kernels[kn]["ms_per_step"] * 1e-3)
            roof["iir_kernel"] = {"name": kn, "achieved": round(a, 1), "unit": "GB/s", "frac": round(a / HBM_PEAK_GBS, 4)}
        line = {
            # BASELINE.json's metric string; `value` is the whole-job aggregate, `per_gpu_value` the per-GPU rate
            "metric": "Msamples/s/GPU (64-ch fused biquad\u2192FIR\u2192FFT-conv chain); % HBM roofline"
                      if args.workload == "chain" else f"Msamples/s ({args.workload})",
            "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": {"chain": "f64 (IIR) + f32 (FFT); f32 I/O", "sos": "f64; f32 I/O", "fir": "f32", "fftconv": "f32"}[args.workload],
            "data": "synthetic", "per_gpu_value": round(value / world, 1),
            "config": {"workload": desc, "channels_per_gpu": C, "seconds": seconds, "fs": FS,
                       "samples_per_gpu": samples, "parallelism": f"channel-shard x{world}, no data-path collective",
                       "iir_precision": os.environ.get("TORCHFX_AMD_IIR_PRECISION", "f64")},
            "roofline": roof,
            "kernels": kernels, "gpu_ms_per_step_sum_of_kernels": round(gpu_ms, 4),
            "kernels_note": "timed region: overlap-save passes of alternate slabs overlap on two internal streams, so "
                            "per-kernel times there include contention; kernels_single_stream = same kernels, one stream, untimed pass",
            "kernels_single_stream": kernels_serial,
        }
        if line_ols:
            line["config"]["overlap_save"] = line_ols
        if variant:
            line["variants"] = {"spectral_fusion": variant}
        if gather_ms is not None:
            line["gather_ms"] = round(gather_ms, 2)
        if not args.no_cpu_baseline and world == 1:      # rank 0 at N=1 only (contract)
            try:
                sec, ch = {"chain": (600.0, 32), "sos": (600.0, 32), "fir": (120.0, 16), "fftconv": (600.0, 32)}[args.workload]
                line["cpu_baseline"] = cpu_baseline(args.workload, sec, ch)     # ~10-20 s of CPU work
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                line["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
