#!/usr/bin/env python3
"""bench.py -- the driver's benchmark contract for the torchfx.filter hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the hot path over one batch of synthetic audio already resident in HBM.
Default workload = the configuration BASELINE.json's metric is quoted on: the fused pipe chain
4-section SOS cascade | FIR-1024 | 65536-tap reverb IR  over 64 channels x 600 s @ 48 kHz float32 PER
GPU (configs[4] is 512 channels over 8 GPUs = 64 per GPU), executed THROUGH THE PRODUCT ENTRY POINT: every
step is ``(Wave(x) | f1 | f2 | fir | rev).ys`` -- pipe operators, plan lookup (plan cache) and the planned
modules (default fusion policy) are all inside the timed region; ``end_to_end`` reports the first call
(planning included) beside the steady state.
At N = 1 the line also carries the driver-timed stage figures of configs[1..3] (``stages``) and two
untimed variants of the chain (``variants``).  Channels shard across ranks with no data-path
collective: ``--scaling weak`` (default) keeps 64 channels per GPU, ``--scaling strong
--total-channels 512`` splits a fixed batch; ``--gather`` adds the final RCCL gather to rank 0 and
reports ``value_with_gather`` beside ``value``.

Prints ONE JSON line on rank 0.  ``roofline`` follows SURVEY.md 8(d): algorithmic bytes are 8 B per
sample-channel (4 B read + 4 B written) for every stage and for the fused chain.  ``cpu_baseline``
times the oracle (our CPU restatement of the reference path, pinned by golden vectors) on a bounded
sample of the same workload -- a reported baseline, never a target.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FS = 48000
COPY_RATE_GBS = 6290.0     # sustained device copy rate (MI355X_MICROARCH.md), the ceiling of a multi-pass design
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_PEAK_TF = 157.3
def _latest_traffic_file() -> str:
    """profiles/rNN_traffic.json of the latest round that has one (tools/profile_gpu.sh -> tools/make_traffic.py)."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_traffic.json")))
    return os.path.relpath(found[-1], ROOT) if found else os.path.join("profiles", "r02_traffic.json")


TRAFFIC_FILE = _latest_traffic_file()


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


def plan_chain(x: torch.Tensor, fuse_fir: bool | None = None, fuse_spectral: bool | None = None, fuse_recursive: bool | None = None):
    """The product's execution plan for  x | f1 | f2 | fir | rev  (None = the Wave's default policy)."""
    from torchfx_amd import Wave

    f1, f2, fir, rev = build_filters()
    w = Wave(x, FS, device=x.device)
    if fuse_fir is not None:
        w.fuse_fir = fuse_fir
    if fuse_spectral is not None:
        w.fuse_spectral = fuse_spectral
    if fuse_recursive is not None:
        w.fuse_recursive = fuse_recursive
    plan = (w | f1 | f2 | fir | rev).plan()
    names = []
    for m in plan:
        taps = getattr(m, "kernel", None)
        if type(m).__name__ == "CascadeFIR":
            names.append(f"CascadeFIR[{m._sos.shape[0]} sections, float64 recursion inside the column pass | {m.fir.kernel.numel()} taps, fft]")
        else:
            names.append(f"{type(m).__name__}[{m._sos.shape[0]} sections]" if hasattr(m, "_sos") and taps is None
                         else f"{type(m).__name__}[{taps.numel()} taps, {m._conv_mode}]")
    return plan, " | ".join(names)


def run_plan(plan, x: torch.Tensor) -> torch.Tensor:
    y = x
    for m in plan:
        if hasattr(m, "reset_state") and hasattr(m, "_stream"):
            m.reset_state()                     # every step is a fresh wave: no state carried between steps
        y = m(y)
    return y


def _ols_taps(plan) -> int:
    return max(int((m.fir if hasattr(m, "fir") else m).kernel.numel()) for m in plan if hasattr(m, "kernel") or hasattr(m, "fir"))


def make_step(workload: str, x: torch.Tensor):
    """Returns (step_fn, description, taps of the overlap-save pass or None)."""
    from torchfx_amd import torchfx_ext as E

    f1, f2, fir, rev = build_filters()
    sos = torch.cat([f1._sos, f2._sos]).contiguous()
    if workload == "sos":
        return (lambda: E.sos_forward(x, None, sos, None, None)[0]), "cfg2: fused 4-section SOS cascade (LoButterworth-6 | ParametricEQ), float64 recursion", None
    if workload == "sos_auto":
        info = E.sos_plan_info(sos)
        return (lambda: E.sos_forward(x, None, sos, None, None, precision="auto")[0]), (
            f"cfg2 cascade with precision='auto' (opt-in): {info['auto_precision']} recursion, estimated largest error "
            f"{info['f32_error_bound']:.2e} of max(1, max|y|), bound 2e-5"), None
    if workload == "fir":
        k = fir.kernel.reshape(-1)
        return (lambda: E.fir_direct_forward(x, k)), "cfg3: direct FIR, 1024 taps (exact-f32 MFMA Toeplitz)", None
    if workload == "fir_fft":
        m = fir                                         # F.FIR(firwin(1024, ...)): conv_mode "fft" is the reference's DEFAULT (fir.py:510,552)
        assert m._conv_mode == "fft"
        return (lambda: m(x)), "cfg3's filter through FIR.forward's default FFT mode: one-launch LDS-resident overlap-save (8192-point block), 1024 taps", None
    if workload == "fftconv":
        k = rev.kernel.reshape(-1)
        return (lambda: E.fft_conv_forward(x, k, (k.numel() - 1, 0))), "cfg4: overlap-save FFT conv, 65536 taps", 65536
    if workload.startswith("fftconv_k"):                 # other kernel lengths of the FFT mode (not BASELINE configs): fftconv_k4096, ...
        K = int(workload[len("fftconv_k"):])
        kk = torch.from_numpy(reverb_ir(K)[::-1].copy())
        info = E.ols_plan_info(K, x.shape[-1], (K - 1, 0))
        return (lambda: E.fft_conv_forward(x, kk, (K - 1, 0))), (
            f"overlap-save FFT conv, {K} taps: path {info['path']}, block {info['N']}, hop {info['S']}"), None
    if workload == "iir_fir1024":
        from torchfx_amd import Wave
        plan = (Wave(x, FS, device=x.device) | f1 | f2 | fir).plan()
        names = " | ".join(type(m).__name__ + (f"[{m.kernel.numel()} taps]" if getattr(m, "kernel", None) is not None else "") for m in plan)
        return (lambda: (Wave(x, FS, device=x.device) | f1 | f2 | fir).ys), (
            f"(Wave(x) | 4xbiquad | FIR-1024).ys, default fusion policy = {names}"), None
    if workload == "chain":
        from torchfx_amd import Wave
        plan, names = plan_chain(x)                       # for the description only; every step plans for itself
        return (lambda: (Wave(x, FS, device=x.device) | f1 | f2 | fir | rev).ys), (
            f"cfg5/GPU: (Wave(x) | 4xbiquad | FIR-1024 | FFT-conv-65536).ys, default fusion policy = {names}"), _ols_taps(plan)
    if workload == "chain_iir_kernel":
        plan, names = plan_chain(x, fuse_fir=True, fuse_spectral=False, fuse_recursive=False)
        return (lambda: run_plan(plan, x)), f"chain with the IIR as its own float64 recursive pass (round 1-4 staging of the like-for-like plan) = {names}", _ols_taps(plan)
    if workload == "chain_fold":
        plan, names = plan_chain(x, fuse_fir=True, fuse_spectral=True)
        est = getattr(plan[0], "fold_error_estimate", None)
        make_step.fold_error_estimate = est
        return (lambda: run_plan(plan, x)), (f"chain with the spectral fold (opt-in, TORCHFX_AMD_FUSE_SPECTRAL=1): the cascade joins the FIR run as its "
                                            f"impulse response, IIR part in float32 FFT arithmetic = {names}"), _ols_taps(plan)
    if workload == "chain_reference_staging":
        plan, names = plan_chain(x, fuse_fir=False, fuse_spectral=False, fuse_recursive=False)
        return (lambda: run_plan(plan, x)), f"chain staged as the reference stages it (TORCHFX_AMD_FUSION=reference) = {names}", _ols_taps(plan)
    raise SystemExit(f"unknown workload {workload}")


def cpu_baseline(workload: str, seconds: float, channels: int) -> dict:
    """Oracle on the host, bounded sample of the same workload.  The reference's CPU path is parallel over
    channels only (OpenMP loop, iir_cpu.cpp:106), so the baseline runs one thread per channel -- `cores` is
    what actually ran -- plus, on a quarter of the sample, one thread."""
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

    want = max(1, min(os.cpu_count() or 1, channels))
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
                "scipy_value": scipy_val, "host_cores": os.cpu_count(),
                "sample": f"{channels} ch x {seconds:g} s @ 48 kHz float32, the reference's own sos_forward_cpu "
                          f"(oracle/_ref/torchfx_ext.so, -O3 -ffast-math -fopenmp as its CMakeLists) incl. its float64 "
                          f"casts, {cores} OpenMP threads (one per channel), {ref_iir['seconds']:.2f} s; port_value = our C "
                          f"oracle, same sample and threads"}
    return {"value": round(channels * T / dt / 1e6, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "single_thread_value": single, "scipy_value": scipy_val, "reference_iir_stage": ref_iir,
            "host_cores": os.cpu_count(),
            "sample": f"{channels} ch x {seconds:g} s @ 48 kHz float32, oracle (C float64 DF1 + numpy overlap-save, "
                      f"reference framing N=int(5K)), {cores} threads (one per channel: the reference's CPU path is "
                      f"parallel over channels only), {dt:.2f} s; single_thread_value on {x1.shape[0]} ch, {dt1:.2f} s"}


def source_digest() -> str:
    """Identifies the kernels that were profiled: sha256 over the HIP sources with comments and whitespace removed
    (profiles/*_traffic.json records the digest of the build its PMC numbers were taken from)."""
    import re
    h = hashlib.sha256()
    d = os.path.join(ROOT, "torchfx_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            src = open(os.path.join(d, name), encoding="utf-8", errors="replace").read()
            src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)          # the code, not its comments or layout
            src = re.sub(r"//[^\n]*", " ", src)
            h.update(name.encode() + b"\0" + " ".join(src.split()).encode())
    return h.hexdigest()[:16]


def timed_region(step, steps: int, warmup: int, sync, lib, profile: bool = True):
    """W untimed steps, then exactly K timed ones.  The previous step's result is dropped BEFORE the next step runs
    (a caller that is done with it): the caching allocator then hands the same 7.4 GB block to every step, so no
    step of the timed region -- not even the first one after a single warm-up -- contains a fresh hipMalloc.
    `profile`: record the library's per-kernel HIP events during the timed steps.  The headline region runs WITHOUT them
    (two event records per launch cost ~1.5 us each, and a cache-sized-slab overlap-save step is ~270 launches: 8.4 ms
    becomes 9.3 with events); the per-kernel table comes from an identical region with events right after it."""
    out = None
    first_ms = None
    for i in range(warmup):
        out = None
        if i == 0:
            sync()
            f0 = time.perf_counter()
        out = step()
        if i == 0:
            sync()
            first_ms = (time.perf_counter() - f0) * 1e3     # the very first call: planning, tables, workspaces
    sync()
    if profile:
        lib.tfx_prof_enable(1)
        lib.tfx_prof_collect()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = None
        out = step()
    sync()
    elapsed = time.perf_counter() - t0
    prof = {}
    if profile:
        prof = json.loads(lib.tfx_prof_collect().decode())
        lib.tfx_prof_enable(0)
    timed_region.first_ms = first_ms
    return elapsed, prof, out


def group_stats(groups: list[float]) -> dict:
    """min / median / max of the per-step times of the timed groups.  A FIRST group slower than 1.5 x the median of all groups
    (clock ramp after the idle gap, a one-off allocator stall) is dropped and reported as dropped; nothing else is."""
    used, dropped = list(groups), None
    if len(used) >= 4 and used[0] > 1.5 * float(np.median(used)):
        dropped, used = used[0], used[1:]
    return {"min": round(min(used), 4), "median": round(float(np.median(used)), 4), "max": round(max(used), 4),
            "groups_used": len(used), "dropped_first_group_ms": None if dropped is None else round(dropped, 4)}


def batch_timed(step, sync, lib, batches: int, per_batch: int, warmup: int = 2, min_group_ms: float = 20.0):
    """Secondary figures (stages, variants): `batches` groups of `per_batch` back-to-back steps (at least 9 groups when a step
    is shorter than 1 ms), device synchronised around every group, no event profiling; returns (median group time / per_batch
    in ms, all group values, per-kernel ms per step from the library's HIP events -- one extra group, recorded with events
    after the timed ones --, last output); `batch_timed.stats` = group_stats() of the groups (min / median / max, a slow first
    group dropped and named).  The median keeps a one-off allocator stall (a fresh multi-GB hipMalloc inside torch.empty) out
    of a 5-step figure."""
    out = None
    for _ in range(warmup):
        out = None
        out = step()
    sync()
    # sub-millisecond steps: a 5-step group is mostly the synchronisation around it and the clock ramp after the idle gap;
    # make every group last >= min_group_ms of back-to-back steps (the per-step figure is still group time / steps)
    t0 = time.perf_counter()
    out = None
    out = step()
    sync()
    est_ms = (time.perf_counter() - t0) * 1e3
    per_batch = max(per_batch, min(200, int(np.ceil(min_group_ms / max(est_ms, 1e-3)))))
    if est_ms < 1.0:
        batches = max(batches, 9)
    groups = []
    for _ in range(batches):
        sync()
        t0 = time.perf_counter()
        for _ in range(per_batch):
            out = None
            out = step()
        sync()
        groups.append((time.perf_counter() - t0) / per_batch * 1e3)
    lib.tfx_prof_enable(1)
    lib.tfx_prof_collect()
    for _ in range(per_batch):
        out = None
        out = step()
    sync()
    prof = json.loads(lib.tfx_prof_collect().decode())
    lib.tfx_prof_enable(0)
    kern = {k: round(v["total_ms"] / per_batch, 4) for k, v in prof.items()}
    batch_timed.per_batch = per_batch
    batch_timed.stats = group_stats(groups)
    return batch_timed.stats["median"], [round(g, 4) for g in groups], kern, out


def kernel_table(prof: dict, steps: int) -> dict:
    return {name: {"launches_per_step": round(v["calls"] / steps, 2),
                   "avg_ms_per_launch": round(v["total_ms"] / v["calls"], 4),
                   "ms_per_step": round(v["total_ms"] / steps, 4)} for name, v in prof.items()}


# the reference's own published benchmark shapes (IIR only), NVIDIA Quadro RTX 6000, pytest-benchmark
# means with a device synchronise per round.  Different card, launch-bound shapes: context, not the headline.
PUBLISHED_RTX6000_MS = {
    "iir_chain_4x_order2_1s_x_1ch": 1.13, "iir_chain_4x_order2_5s_x_2ch": 2.39, "iir_chain_4x_order2_60s_x_8ch": 75.8,
    "butterworth6_60s_x_1ch": 15.2, "butterworth6_60s_x_8ch": 75.3,
    "sos_cascade_order4_5s_x_2ch": 1.28, "sos_cascade_order8_5s_x_2ch": 2.43,
}


def published_context(dev, reps: int = 30, warm: int = 5) -> dict:
    """The reference's published benchmark shapes on this backend, called the way its benchmarks call them
    (benchmarks/test_iir_bench.py:28-34,162-180: ``nn.Sequential`` of four order-2 filters applied to a resident
    tensor, stateful, one device synchronise per round; test_pipeline_bench.py:19-38; 44.1 kHz), median of `reps`
    rounds after `warm`; beside the like-for-like ``nn.Sequential`` call the same chain through the ``Wave`` pipe
    (one fused cascade launch)."""
    from torch import nn
    from torchfx_amd import Wave
    from torchfx_amd import filter as F
    fs = 44100
    out = {}

    def signal(ch, sec):
        x = torch.randn(ch, int(fs * sec), device=dev, dtype=torch.float32)
        return x / x.abs().max()

    def per_call_ms(fn):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize(dev)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize(dev)
            ts.append((time.perf_counter() - t0) * 1e3)
        return float(np.median(ts))

    def chain():
        return nn.Sequential(F.HiButterworth(cutoff=1000, order=2, fs=fs), F.LoButterworth(cutoff=5000, order=2, fs=fs),
                             F.HiChebyshev1(cutoff=1500, order=2, fs=fs), F.LoChebyshev1(cutoff=1800, order=2, fs=fs))
    for sec, ch in ((1, 1), (5, 2), (60, 8)):
        key = f"iir_chain_4x_order2_{sec}s_x_{ch}ch"
        try:
            x = signal(ch, sec)
            seq = chain()
            for f in seq:
                f.compute_coefficients()
            ms_seq = per_call_ms(lambda: seq(x))
            members = list(chain())
            ms_pipe = per_call_ms(lambda: (Wave(x, fs, device=dev) | members[0] | members[1] | members[2] | members[3]).ys)
            out[key] = {"ms_per_call_nn_sequential": round(ms_seq, 4), "ms_per_call_wave_pipe": round(ms_pipe, 4),
                        "Msamples_per_s_wave_pipe": round(ch * sec * fs / ms_pipe / 1e3, 1)}
        except Exception as e:
            out[key] = {"error": repr(e)}
    for ch in (1, 8):
        key = f"butterworth6_60s_x_{ch}ch"
        try:
            x = signal(ch, 60)
            f = F.LoButterworth(cutoff=2000, order=6, fs=fs)
            f.compute_coefficients()
            out[key] = {"ms_per_call": round(per_call_ms(lambda: f(x)), 4)}
        except Exception as e:
            out[key] = {"error": repr(e)}
    for order in (4, 8):
        key = f"sos_cascade_order{order}_5s_x_2ch"
        try:
            x = signal(2, 5.0)
            f = F.LoButterworth(cutoff=2000, order=order, fs=fs)
            f.compute_coefficients()
            out[key] = {"ms_per_call": round(per_call_ms(lambda: f(x)), 4)}
        except Exception as e:
            out[key] = {"error": repr(e)}
    for key, ent in out.items():
        ours = ent.get("ms_per_call", ent.get("ms_per_call_nn_sequential"))
        if ours:
            ent["published_ms_rtx6000"] = PUBLISHED_RTX6000_MS[key]
            ent["published_over_ours"] = round(PUBLISHED_RTX6000_MS[key] / ours, 2)
    return out


def cfg5_on_one_gpu(dev, sync, lib, total_channels: int = 512, seconds: float = 600.0) -> dict:
    """BASELINE cfg 5 at its full size on ONE device (the N = 1 point of `--scaling strong --total-channels 512`):
    512 ch x 600 s = 59 GB in, 59 GB out, one `.ys` per step."""
    free, _ = torch.cuda.mem_get_info(dev)
    T = int(seconds * FS)
    need = 2 * total_channels * T * 4 + (8 << 30)
    if free < need:
        return {"skipped": f"needs {need >> 30} GB of free HBM, {free >> 30} GB free"}
    x = torch.empty(total_channels, T, device=dev, dtype=torch.float32)
    for c0 in range(0, total_channels, 64):
        g = torch.Generator(device=dev).manual_seed(1234 + c0 // 64)         # shard r of the 8-GPU run has seed 1234 + r
        blk = x[c0:c0 + 64]
        blk.normal_(generator=g)
        blk.mul_(1.0 / float(blk.abs().max()))
    step, desc, _ = make_step("chain", x)
    elapsed, _, out = timed_region(step, 3, 1, sync, lib, profile=False)
    ms = elapsed / 3 * 1e3
    del out, x
    torch.cuda.empty_cache()
    return {"workload": desc, "channels": total_channels, "seconds": seconds, "ms_per_step": round(ms, 3),
            "Msamples_per_s": round(total_channels * T / ms / 1e3, 1),
            "frac_of_8TBps_at_8B_per_sample": round(8.0 * total_channels * T / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "what": "the whole cfg-5 batch on one GPU = N = 1 of the strong-scaling curve; an 8-GPU run gives each rank 64 of these channels"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="chain",
                    choices=["chain", "sos", "fir", "fir_fft", "fftconv", "chain_iir_kernel", "chain_fold", "chain_reference_staging"])
    ap.add_argument("--channels", type=int, default=64, help="channels PER GPU (weak scaling)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--total-channels", type=int, default=512, help="fixed batch of --scaling strong (cfg 5)")
    ap.add_argument("--seconds", type=float, default=None, help="signal length (default: 600 chain/fftconv, 60 sos/fir)")
    ap.add_argument("--gather", action="store_true", help="also time the final RCCL gather to rank 0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only warm-up + timed steps (for rocprofv3 runs: no stage / variant / single-stream passes)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, rendezvous on 127.0.0.1) by
        # replacing this process with torch.distributed.run on the same command line; rank 0 prints the one JSON line
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.setdefault("OMP_NUM_THREADS", "8")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus} "
                         "(or run plain `python bench.py --gpus N`, which starts the ranks itself)")
    import torch.distributed as dist

    # TFX_BENCH_SHARE_DEVICE=1 (development only): every rank uses cuda:0 and the collectives go over
    # gloo, so the N > 1 control flow can be exercised on a one-GPU box; never set by the driver
    share = os.environ.get("TFX_BENCH_SHARE_DEVICE", "0") == "1"
    # one rank per GPU: LOCAL_RANK indexes the visible devices; a launcher that isolates each rank with HIP_VISIBLE_DEVICES leaves one
    # visible device per rank (index 0) -- the UUID check before the line is printed catches ranks that really share a GPU
    local_dev = 0 if (share or torch.cuda.device_count() <= local) else local
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
    from torchfx_amd import torchfx_ext as _E
    _E.prewarm(dev)                        # what a latency-conscious caller does at start-up: the library's one-time device set-up
    #                                        (code-object load, internal streams) runs on a helper thread while the signal is generated

    chainlike = args.workload.startswith("chain") or args.workload == "fftconv"
    seconds = args.seconds if args.seconds is not None else (600.0 if chainlike else 60.0)
    if args.scaling == "strong":
        # a fixed batch split like torchfx_amd.distributed.shard_bounds: contiguous blocks, the first `total % world` ranks one
        # row more (9 channels over 8 ranks = 2, 1, 1, ...; fewer channels than ranks leaves empty blocks: those ranks idle)
        from torchfx_amd.distributed import shard_bounds as _sb
        lo_, hi_ = _sb(args.total_channels, world, rank)
        C = hi_ - lo_
        total_rows = args.total_channels
    else:
        C = args.channels
        total_rows = C * world
    T = int(seconds * FS)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.randn(C, T, device=dev, generator=gen, dtype=torch.float32)
    if C > 0:
        x.mul_(1.0 / float(x.abs().max()))     # max|x| <= 1 (benchmarks/conftest.py:70-82 of the reference)
    if os.environ.get("TFX_BENCH_ZERO_INPUT", "0") == "1":
        x.zero_()                          # development only: a compute-bound kernel without data toggling (clock / power study)

    step, desc, ols_taps = make_step(args.workload, x)

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    elapsed, _, out = timed_region(step, args.steps, args.warmup, sync, lib, profile=False)      # THE timed region: wall clock only
    first_call_ms = timed_region.first_ms
    out = None
    elapsed_prof, prof, out = timed_region(step, args.steps, 0, sync, lib, profile=True)       # same steps again, with per-kernel events
    # what the process group itself says about its size (an all-reduce of ones over RCCL / gloo), and which device
    # every rank drives -- WORLD_SIZE is only what the launcher claimed
    from torchfx_amd.distributed import ranks_seen as _ranks_seen
    seen = _ranks_seen(device=None if share else dev) if world > 1 else 1
    props = torch.cuda.get_device_properties(dev)
    me = {"rank": rank, "device": f"cuda:{local_dev}", "name": props.name, "uuid": str(getattr(props, "uuid", "")),
          "gcn_arch": getattr(props, "gcnArchName", ""), "cus": props.multi_processor_count}
    devices = [me]
    if world > 1:
        devices = [None] * world
        dist.all_gather_object(devices, me)
    if world > 1:
        t = torch.tensor([elapsed], device="cpu" if share else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    extras = rank == 0 and world == 1 and not args.no_extras
    # per-kernel durations without overlap: the overlap-save passes of consecutive slabs run on internal
    # streams in the timed region, which inflates each kernel's elapsed time; one extra, untimed pass on a
    # single stream gives the clean per-kernel numbers
    prof_serial = None
    if extras and chainlike:
        old_env = {k: os.environ.get(k) for k in ("TFX_OLS_STREAMS", "TFX_OLS_SOS_STREAMS")}
        os.environ["TFX_OLS_STREAMS"] = os.environ["TFX_OLS_SOS_STREAMS"] = "1"
        lib.tfx_env_reload()               # the library reads its knobs once per process
        try:
            _, prof_serial, _ = timed_region(step, 2, 1, sync, lib)
        finally:
            for k, v in old_env.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            lib.tfx_env_reload()

    variants = None
    if extras and args.workload == "chain":
        # secondary figures, outside the timed region: the same chain with the IIR as its own float64
        # recursive pass (round-1 default) and staged exactly like the reference
        variants = {}
        head = out[:, : 4 * FS].clone()
        for vname in ("chain_fold", "chain_iir_kernel", "chain_reference_staging"):
            try:
                vstep, vdesc, _ = make_step(vname, x)
                vel, _, vout = timed_region(vstep, args.steps, 2, sync, lib, profile=False)     # the same region as the headline
                vms = vel / args.steps * 1e3
                variants[vname] = {"what": vdesc, "ms_per_step": round(vms, 4), "steps": args.steps,
                                   "Msamples_per_s": round(C * T / vms / 1e3, 1),
                                   "frac_of_8TBps_at_8B_per_sample": round(8.0 * C * T / (vms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                   "max_abs_diff_vs_default_chain_first_4s": float((vout[:, : 4 * FS] - head).abs().max())}
                if vname == "chain_fold":
                    variants[vname]["fold_error_estimate"] = getattr(make_step, "fold_error_estimate", None)
                del vout, vstep
            except Exception as e:
                variants[vname] = {"error": repr(e)}
        del head

    stages = None
    if extras and args.workload == "chain":
        # configs[1..3] of BASELINE.json, timed by this same driver-run process (wall clock, see batch_timed),
        # kernel time from the library's HIP events beside it
        stages = {}
        for key, wl, sec, bound in (("cfg2", "sos", 60.0, "hbm"), ("cfg2_precision_auto", "sos_auto", 60.0, "hbm"),
                                    ("cfg3", "fir", 60.0, "mfma"), ("cfg3_fft_mode", "fir_fft", 60.0, "hbm"),
                                    ("fft_mode_4096_taps", "fftconv_k4096", 60.0, "hbm"), ("fft_mode_8192_taps", "fftconv_k8192", 60.0, "hbm"),
                                    ("iir_fir1024_pipe", "iir_fir1024", 60.0, "hbm"), ("cfg4", "fftconv", 600.0, "hbm")):
            try:
                xs = x if sec == seconds else x[:, : int(sec * FS)].contiguous()
                sstep, sdesc, _ = make_step(wl, xs)
                sms, sgroups, skern, sout = batch_timed(sstep, sync, lib, 5, 5)
                n = xs.numel()
                ent = {"workload": sdesc, "channels": C, "seconds": sec, "ms_per_step": round(sms, 4),
                       "steps_per_group": batch_timed.per_batch, "ms_per_step_groups": sgroups, "ms_per_step_stats": batch_timed.stats,
                       "Msamples_per_s": round(n / sms / 1e3, 1), "bound": bound, "kernel_ms_per_step": skern}
                if bound == "hbm":
                    ent["achieved_GBps"] = round(8.0 * n / (sms * 1e-3) / 1e9, 1)
                    ent["frac"] = round(ent["achieved_GBps"] / HBM_PEAK_GBS, 4)
                else:
                    ent["achieved_TFLOPs"] = round(2.0 * 1024 * n / (sms * 1e-3) / 1e12, 2)
                    ent["frac"] = round(ent["achieved_TFLOPs"] / FP32_PEAK_TF, 4)
                    ent["frac_of_hbm_roofline"] = round(8.0 * n / (sms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                stages[key] = ent
                del sout, sstep, xs
            except Exception as e:
                stages[key] = {"error": repr(e)}

    cfg5_one = published = None
    plan_host_ms = None
    if extras and args.workload == "chain":
        try:                                   # host cost of planning in steady state (plan cache hit), per .ys
            from torchfx_amd import Wave
            pf = build_filters()
            (Wave(x, FS, device=dev) | pf[0] | pf[1] | pf[2] | pf[3]).plan()
            t0 = time.perf_counter()
            for _ in range(50):
                (Wave(x, FS, device=dev) | pf[0] | pf[1] | pf[2] | pf[3]).plan()
            plan_host_ms = (time.perf_counter() - t0) / 50 * 1e3
        except Exception:
            pass
        try:
            published = published_context(dev)
        except Exception as e:
            published = {"error": repr(e)}
        try:
            out = None
            torch.cuda.empty_cache()
            cfg5_one = cfg5_on_one_gpu(dev, sync, lib, args.total_channels, seconds)
        except Exception as e:
            cfg5_one = {"error": repr(e)}

    gather_ms = None
    if args.gather and world > 1:
        from torchfx_amd.distributed import gather_rows
        sync()
        g0 = time.perf_counter()
        gathered = gather_rows(out, total_rows, dst=0)         # one RCCL gather: each peer -> root over its own xGMI link
        sync()
        gather_ms = (time.perf_counter() - g0) * 1e3
        if rank == 0:
            assert gathered.shape == (total_rows, out.shape[1])
        del gathered
        if world > 1:
            tg = torch.tensor([gather_ms], device="cpu" if share else dev, dtype=torch.float64)
            dist.all_reduce(tg, op=dist.ReduceOp.MAX)
            gather_ms = float(tg.item())

    if world > 1 and seen != world:
        raise SystemExit(f"bench.py: the process group sees {seen} ranks, --gpus says {world}: no line printed")
    uuids = [d.get("uuid", "") for d in devices]
    if world > 1 and not share and len(set(uuids)) != world:
        raise SystemExit(f"bench.py: {world} ranks drive {len(set(uuids))} distinct devices ({sorted(set(uuids))}): no line printed")
    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        samples = C * T
        value = total_rows * T / (elapsed / args.steps) / 1e6             # whole-job Msamples/s (all ranks' rows / slowest rank's time)
        kernels = kernel_table(prof, args.steps)
        # per-kernel HBM model: the bytes each launch group must move by design (not the 8 B/sample
        # algorithmic figure) -> achieved GB/s of that kernel; PMC-measured traffic agrees within a few %
        line_ols = None
        try:
            from torchfx_amd import torchfx_ext as E
            model = {"sos_stream_kernel<f64>": 8.0 * samples, "sos_stream_kernel<f32>": 8.0 * samples,
                     "ols_lds4096_kernel": 8.0 * samples, "ols_lds8192_kernel": 8.0 * samples, "ols_lds16k_r4_kernel": 8.0 * samples}
            if ols_taps:
                kk = ols_taps
                info = E.ols_plan_info(kk, T, (kk - 1, 0))
                _f = build_filters()
                _sos = torch.cat([_f[0]._sos, _f[1]._sos])
                fused = E.sos_fft_conv_plan_info(T, _sos, kk, (kk - 1, 0)) if args.workload == "chain" else None
                if fused and "CascadeFIR" in desc:      # the recursion-in-pass-A pipeline picks its own block (2^21 on long rows)
                    info = dict(info, N=fused["N"], S=fused["S"], F=fused["F"])
                frames = C * info["F"]
                pairs = (frames + 1) // 2
                n = info["N"]
                model["ols_col_fwd16_kernel"] = frames * n * 4.0 + pairs * n * 8.0
                # the recursion's column pass also reads every row's warm-up (64 of 256 blocks for the cfg-2 cascade at 2^-48, round 5; fewer at round 6's 2^-40)
                wb = -(-max(0, E.sos_fft_conv_warmup(_sos)) // 32)
                rows_blk = n // 256 // 32                               # column blocks per row: 128 (4096-sample rows) or 256 (8192)
                model["ols_col_fwd16_sos_kernel"] = frames * n * 4.0 * (1.0 + min(wb, rows_blk) / rows_blk) + pairs * n * 8.0
                model["ols_row8192_kernel"] = pairs * n * 16.0
                model["ols_col_inv16_kernel"] = pairs * n * 8.0 + 4.0 * samples
                for nm in ("ols_row_kernel", "ols_row1024_kernel", "ols_row4096_kernel"):
                    model[nm] = pairs * n * 16.0
                line_ols = {"taps": kk, "fft_block": n, "hop": info["S"], "blocks_per_row": info["F"],
                            "native_lds_fft": info["native"], "model_bytes_per_sample": round((20.0 * n / info["S"]) + 4.0, 2)}
            for name, k in kernels.items():
                if name in model and k["ms_per_step"] > 0:
                    k["model_GB_per_step"] = round(model[name] / 1e9, 3)
                    k["GBps"] = round(model[name] / 1e9 / (k["ms_per_step"] * 1e-3), 1)
                    k["frac_of_8TBps"] = round(k["GBps"] / HBM_PEAK_GBS, 4)
        except Exception:
            pass
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
        multi_pass = chainlike and ols_taps is not None
        if multi_pass:
            # overlap-save steps: three kernels of comparable weight overlap on internal streams, so "the dominant
            # kernel" flips between runs and its elapsed time is stretched by its neighbours.  The roofline figure
            # is therefore the STEP: algorithmic bytes of the step / step time; the per-kernel view sits beside it.
            roof["achieved"] = round(step_ach, 2)
            roof["frac"] = round(step_ach / roof["peak"], 4)
            roof["frac_is"] = "whole step (all passes share the one 8 B/sample-channel)"
            roof["dominant_kernel_in_timed_region"] = {"achieved": round(ach, 2), "frac": round(ach / roof["peak"], 4),
                                                       "note": "three passes overlap: per-kernel elapsed times include contention"}
        else:
            roof["achieved"] = round(ach, 2)
            roof["frac"] = round(ach / roof["peak"], 4)
            roof["frac_is"] = "dominant kernel: algorithmic work of one launch / its average duration (HIP events)"
        roof["step_achieved"] = round(step_ach, 2)
        roof["step_frac"] = round(step_ach / roof["peak"], 4)
        if multi_pass and line_ols:
            # what THIS design can reach at best: a transform longer than LDS is three passes over a complex workspace, i.e.
            # model bytes per sample (20 N / S + 4, + the warm-up re-reads of the recursion's column pass) moved at the rate a
            # plain device copy sustains on this part (MI355X_MICROARCH.md: 6.29 TB/s) -- read `frac` against this, not 1.0
            mb = float(line_ols["model_bytes_per_sample"])
            if "ols_col_fwd16_sos_kernel" in kernels:
                mb += 4.0 * min(wb, rows_blk) / rows_blk * line_ols["fft_block"] / line_ols["hop"]
            roof["design_ceiling"] = {"frac": round(8.0 / mb * COPY_RATE_GBS / HBM_PEAK_GBS, 4), "model_bytes_per_sample": round(mb, 2),
                                      "copy_rate_GBps": COPY_RATE_GBS,
                                      "what": "8 B/sample / (model bytes per sample of the three-pass overlap-save) x the device's sustained copy rate / peak"}
        if args.workload == "chain" and "ols_col_fwd16_sos_kernel" in kernels:
            # the default plan IS the reference's arithmetic: float64 recursion (inside the forward column pass), float32 overlap-save
            roof["step_frac_reference_arithmetic"] = roof["step_frac"]
            roof["reference_arithmetic_means"] = ("float64 DF1 recursion, one rounding to float32, float32 FFT convolution -- the reference's "
                                                  "operations, not its bits: a row of the transform restarts from a warm-up (state true to 2^-40), "
                                                  "FMA contraction and the unit-b0 form reassociate the numerator; parity bars: every section "
                                                  "2e-11 in float64, the float32 cascade output 1 ulp, the chain 1e-5 (tests/gpu_common.py)")
        if variants and "ms_per_step" in variants.get("chain_fold", {}):
            roof["step_frac_spectral_fold"] = variants["chain_fold"]["frac_of_8TBps_at_8B_per_sample"]
        if variants and "ms_per_step" in variants.get("chain_iir_kernel", {}):
            roof["step_frac_iir_as_its_own_pass"] = variants["chain_iir_kernel"]["frac_of_8TBps_at_8B_per_sample"]
        if variants and "ms_per_step" in variants.get("chain_reference_staging", {}):
            roof["step_frac_reference_staging"] = variants["chain_reference_staging"]["frac_of_8TBps_at_8B_per_sample"]
        if dom and prof_serial and dom in prof_serial:
            # the same kernel with nothing else on the chip (the untimed single-stream pass): in the timed region the
            # three overlap-save passes of different slabs run concurrently, so every kernel's elapsed time there is
            # stretched by its neighbours -- the wall clock gains, the per-kernel figure does not show the kernel alone
            alone_ms = prof_serial[dom]["total_ms"] / prof_serial[dom]["calls"]
            per_launch_alone = samples / max(prof_serial[dom]["calls"] / 2.0, 1e-9)
            a1 = unit_work * per_launch_alone / (alone_ms * 1e-3)
            roof["alone"] = {"avg_ms_per_launch": round(alone_ms, 4), "achieved": round(a1, 2), "frac": round(a1 / roof["peak"], 4),
                             "what": "the dominant kernel on one internal stream, untimed extra pass (no concurrent kernels)"}
        # HBM bytes from rocprofv3 PMC passes of this same command (tools/profile_gpu.sh -> profiles/):
        # counters cannot be read from inside the process, so the file written by the last profiling
        # session is reported together with the digest of the HIP sources it was taken from
        try:
            tr = json.load(open(os.path.join(ROOT, TRAFFIC_FILE)))
            ent = tr.get(args.workload)
            if ent and C == 64 and seconds == ent.get("seconds", 600.0):
                roof["step_traffic"] = ent["bytes_per_step"]
                pk = ent.get("per_kernel_GB_per_launch", {}).get(dom)
                if multi_pass:
                    roof["traffic"] = ent["bytes_per_step"]                       # per step, like `achieved`
                elif pk:
                    lps = pk.get("launches_per_step")
                    scale = (lps / kernels[dom]["launches_per_step"]) if lps else 1.0   # same bytes, other slab count
                    roof["traffic"] = round((pk["read"] + pk["write"]) * 1e9 * scale)
                roof["traffic_source"] = (f"{TRAFFIC_FILE} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of this command): L2-miss "
                                          "traffic; the overlap-save workspace share of it is served by the Infinity Cache (64 MB slabs), not HBM")
                roof["traffic_from_this_build"] = tr.get("source_digest") == source_digest()
        except Exception:
            pass
        roof["dominant_kernel"] = dom
        if dom:
            roof["dominant_kernel_avg_ms"] = kernels[dom]["avg_ms_per_launch"]
            roof["dominant_kernel_ms_per_step"] = kernels[dom]["ms_per_step"]
        for kn in ("sos_stream_kernel<f64>", "sos_stream_kernel<f32>"):
            if kn in kernels:
                a = alg_gb / (kernels[kn]["ms_per_step"] * 1e-3)
                roof["iir_kernel"] = {"name": kn, "achieved": round(a, 1), "unit": "GB/s", "frac": round(a / HBM_PEAK_GBS, 4)}
        dtype = {"chain": "f64 (IIR) + f32 (FFT); f32 I/O", "chain_fold": "f32 (overlap-save FFT arithmetic for the whole folded chain); f32 I/O",
                 "chain_iir_kernel": "f64 (IIR) + f32 (FFT); f32 I/O", "chain_reference_staging": "f64 (IIR) + f32 (FFT); f32 I/O",
                 "sos": "f64; f32 I/O", "fir": "f32", "fir_fft": "f32", "fftconv": "f32"}[args.workload]
        if args.workload == "chain" and "sos_stream_kernel<f64>" in kernels:
            dtype = "f64 (IIR) + f32 (FFT); f32 I/O"
        iir_knob = os.environ.get("TORCHFX_AMD_IIR_PRECISION", "f64")
        if "ols_col_fwd16_sos_kernel" in kernels:
            iir_how = ("float64 DF1 recursion (iir_cpu.cpp:132-147) in registers inside the overlap-save pipeline's forward column pass "
                       "(ols_col_fwd16_sos_kernel: one thread per 4096-sample row, warm-up to 2^-40 from zero state), rounded once to float32; "
                       "every section's output readable through y_sections (tests/test_gpu_sos_ols.py)")
        elif args.workload.startswith("chain") and not any(n.startswith("sos_stream_kernel") for n in kernels):
            iir_how = ("folded into the f32 overlap-save pass (fuse_spectral, opt-in: the cascade's impulse response joins the FIR run; "
                       "no recursive kernel runs)")
        elif args.workload in ("fir", "fir_fft", "fftconv"):
            iir_how = "n/a (no IIR stage)"
        else:
            iir_how = f"{iir_knob} recursion in sos_stream_kernel (TORCHFX_AMD_IIR_PRECISION)"
        line = {
            # BASELINE.json's metric string; `value` is the whole-job aggregate, `per_gpu_value` the per-GPU rate
            "metric": "Msamples/s/GPU (64-ch fused biquad→FIR→FFT-conv chain); % HBM roofline"
                      if args.workload == "chain" else f"Msamples/s ({args.workload})",
            "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_step, 4), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None,      # no published number for THIS metric; see vs_baseline_context
            "dtype": dtype,
            "data": "synthetic", "per_gpu_value": round(value / world, 1),
            "config": {"workload": desc, "channels_per_gpu": C, "total_channels": total_rows, "seconds": seconds, "fs": FS,
                       "samples_per_gpu": samples, "parallelism": f"channel-shard x{world}, no data-path collective",
                       "fusion_policy": os.environ.get("TORCHFX_AMD_FUSION", "auto"),
                       "iir_precision": iir_how,
                       "source_digest": source_digest()},
            "roofline": roof,
            "kernels": kernels, "gpu_ms_per_step_sum_of_kernels": round(gpu_ms, 4),
            "kernels_from": "an identical region of the same steps run right after the timed one WITH the library's per-kernel HIP "
                            "events (the timed region itself runs without them: two event records per launch)",
            "ms_per_step_with_event_profiling": round(elapsed_prof / args.steps * 1e3, 4),
            "kernels_note": "timed region: overlap-save passes of different slabs overlap on internal streams, so "
                            "per-kernel times there include contention; kernels_single_stream = same kernels, one stream, untimed pass",
            "kernels_single_stream": kernels_serial,
        }
        line["ranks_seen"] = seen                         # all-reduce of ones on the process group (RCCL at N > 1)
        line["devices"] = devices
        line["distinct_device_uuids"] = len(set(uuids))
        if args.workload == "chain":
            line["end_to_end"] = {
                "what": "(Wave(x) | f1 | f2 | fir | rev).ys from Python: pipe operators + plan lookup + kernels",
                "first_ys_ms": None if first_call_ms is None else round(first_call_ms, 2),
                "steady_ys_ms": round(ms_step, 4),
                "plan_host_ms_steady": None if plan_host_ms is None else round(plan_host_ms, 4),
                "note": "first call = impulse response of the cascade, FFT-multiplied taps, float64 spectrum of the merged "
                        "kernel, tables and workspaces; afterwards the plan cache and the device-side caches are hit"}
        if cfg5_one:
            line["cfg5_512ch_on_one_gpu"] = cfg5_one
        if published:
            line["published_context"] = {"hardware_of_published_numbers": "NVIDIA Quadro RTX 6000 (BASELINE.md section 1) -- a different card",
                                         "note": "the reference's own IIR benchmark shapes (tiny, launch-bound); context only",
                                         "cases": published}
            line["vs_baseline_context"] = {k: v.get("published_over_ours") for k, v in published.items() if isinstance(v, dict)}
        if line_ols:
            line["config"]["overlap_save"] = line_ols
        if stages:
            if first_call_ms is not None:
                stages["first_ys_ms"] = round(first_call_ms, 2)       # first (Wave(x) | ...).ys of the process, planning included
            stages["_timing"] = ("every stage: median of 5 groups (9 when a step is under 1 ms) of back-to-back steps (>= 20 ms per group), wall "
                                 "clock, device synchronised around each group, no event profiling; ms_per_step_stats = min / median / max of the "
                                 "groups used (a first group above 1.5 x the median is dropped and named); kernel_ms_per_step from one more group "
                                 "with the library's HIP events")
            line["stages"] = stages
        if variants:
            line["variants"] = variants
        if gather_ms is not None:
            line["gather_ms"] = round(gather_ms, 2)
            line["value_with_gather"] = round(total_rows * T / (elapsed / args.steps + gather_ms * 1e-3) / 1e6, 1)
        if not args.no_cpu_baseline and world == 1:      # rank 0 at N=1 only (contract)
            try:
                base_wl = "chain" if args.workload.startswith("chain") else args.workload
                base_wl = "fir" if base_wl == "fir_fft" else base_wl
                sec, ch = {"chain": (300.0, 64), "sos": (600.0, 64), "fir": (60.0, 32), "fftconv": (300.0, 64)}[base_wl]
                line["cpu_baseline"] = cpu_baseline(base_wl, sec, ch)     # ~10-20 s of CPU work
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                line["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
