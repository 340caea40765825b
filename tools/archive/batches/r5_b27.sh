#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_b27; mkdir -p $O
export TFX_OLS_SOS_SPLIT=1
for cfg in "2 240" "2 160" "2 120" "3 160" "3 120" "3 80" "4 120" "2 480"; do
  set -- $cfg
  echo "== split streams=$1 pairs=$2"; TFX_OLS_SOS_STREAMS=$1 TFX_OLS_SOS_PAIRS=$2 timeout 600 python tools/sos_ols_bench.py 7 fused 2>&1 | grep "pass A"
done | tee $O/split_sweep.txt
TFX_OLS_SOS_STREAMS=1 TFX_OLS_SOS_PAIRS=480 timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu | tee -a $O/split_sweep.txt
import os, sys, json
sys.path.insert(0, os.getcwd())
import runpy
os.environ["TFX_ENV_DYNAMIC"] = "1"
import numpy as np, torch
from torchfx_amd import _lib, torchfx_ext as E, filter as F
lib = _lib.load()
f1 = F.LoButterworth(2000, order=6, fs=48000); f2 = F.ParametricEQ(frequency=1000, q=2.0, gain=3.0, fs=48000)
f1.compute_coefficients(); f2.compute_coefficients()
sos = torch.cat([f1._sos, f2._sos]); K = 66559
k = np.random.default_rng(0).standard_normal(K) * np.exp(-np.arange(K) / 8000.0)
k = torch.from_numpy((k / np.abs(k).sum()).astype(np.float32)[::-1].copy())
x = torch.rand((64, 28_800_000), device="cuda") * 2 - 1
for sp in ("1", "0"):
    os.environ["TFX_OLS_SOS_SPLIT"] = sp
    E.sos_fft_conv_forward(x, sos, k, (K - 1, 0)); torch.cuda.synchronize()
    lib.tfx_prof_enable(1); lib.tfx_prof_collect()
    for _ in range(3): E.sos_fft_conv_forward(x, sos, k, (K - 1, 0))
    torch.cuda.synchronize()
    prof = json.loads(lib.tfx_prof_collect().decode()); lib.tfx_prof_enable(0)
    print("one lane, 480 pairs per launch, split", sp, {n.replace("ols_", ""): round(v["total_ms"] / 3, 3) for n, v in prof.items()})
PY
