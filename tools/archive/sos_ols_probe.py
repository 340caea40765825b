"""Where the time of the cascade-in-pass-A kernel goes: per-launch duration of ols_col_fwd16_sos_kernel (the library's HIP
events) on ONE lane, for 1 / 2 / 4 sections, with and without warm-up blocks, at one and two workgroups per CU.
usage: python tools/sos_ols_probe.py"""
import ctypes
import json
import os
os.environ.setdefault("TFX_ENV_DYNAMIC", "1")      # this tool flips TFX_* knobs inside one process
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchfx_amd import _lib, torchfx_ext as E  # noqa: E402
from torchfx_amd import filter as F  # noqa: E402

lib = _lib.load()
C, T = 64, 28_800_000
f1 = F.LoButterworth(2000, order=6, fs=48000)
f2 = F.ParametricEQ(frequency=1000, q=2.0, gain=3.0, fs=48000)
f1.compute_coefficients(); f2.compute_coefficients()
sos = torch.cat([f1._sos, f2._sos])
K = 66559
k = np.random.default_rng(0).standard_normal(K) * np.exp(-np.arange(K) / 8000.0)
k = torch.from_numpy((k / np.abs(k).sum()).astype(np.float32)[::-1].copy())
x = torch.rand((C, T), device="cuda") * 2 - 1
os.environ["TFX_OLS_SOS_STREAMS"] = "1"


def run(nsec, warm, pairs, tag):
    os.environ["TFX_OLS_SOS_PAIRS"] = str(pairs)
    if warm is None:
        os.environ.pop("TFX_OLS_SOS_WARM", None)
    else:
        os.environ["TFX_OLS_SOS_WARM"] = str(warm)
    s = sos[:nsec]
    E.sos_fft_conv_forward(x, s, k, (K - 1, 0)); torch.cuda.synchronize()
    lib.tfx_prof_enable(1); lib.tfx_prof_collect()
    for _ in range(2):
        E.sos_fft_conv_forward(x, s, k, (K - 1, 0))
    torch.cuda.synchronize()
    prof = json.loads(lib.tfx_prof_collect().decode())
    lib.tfx_prof_enable(0)
    out = []
    for name, v in (prof.items() if isinstance(prof, dict) else []):
        out.append(f"{name.replace('ols_', '')}: {v}")
    print(f"{tag:44s} sections={nsec} warm={warm} pairs/launch={pairs}\n    {prof}", flush=True)


for pairs in (256, 512):
    run(4, None, pairs, "default warm-up")
    run(4, 0, pairs, "no warm-up blocks")
    run(2, 0, pairs, "no warm-up blocks")
    run(1, 0, pairs, "no warm-up blocks")
    run(1, 2432, pairs, "1 section, 76 warm-up blocks")
