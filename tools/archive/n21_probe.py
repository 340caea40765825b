"""Per-kernel time (library HIP events, ONE lane) of the plain overlap-save pipeline at N = 2^20 and N = 2^21."""
import json, os, sys
os.environ.setdefault("TFX_ENV_DYNAMIC", "1")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchfx_amd import _lib, torchfx_ext as E
lib = _lib.load()
C, T, K = 64, 28_800_000, 66559
k = np.random.default_rng(0).standard_normal(K) * np.exp(-np.arange(K) / 8000.0)
k = torch.from_numpy((k / np.abs(k).sum()).astype(np.float32)[::-1].copy())
x = torch.rand((C, T), device="cuda") * 2 - 1
os.environ["TFX_OLS_STREAMS"] = "1"
for n21 in (0, 1, 0, 1):
    for mb in (64, 1024):
        os.environ.update(TFX_OLS_N21=str(n21), TFX_OLS_SLAB_MB=str(mb))
        E.fft_conv_forward(x, k, (K - 1, 0)); torch.cuda.synchronize()
        lib.tfx_prof_enable(1); lib.tfx_prof_collect()
        for _ in range(3):
            E.fft_conv_forward(x, k, (K - 1, 0))
        torch.cuda.synchronize()
        prof = json.loads(lib.tfx_prof_collect().decode()); lib.tfx_prof_enable(0)
        print(f"N21={n21} slab {mb} MB:", {n.replace('ols_', ''): round(v['total_ms'] / 3, 3) for n, v in prof.items()}, flush=True)
