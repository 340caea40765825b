"""[2, 44100] (the reference's test / benchmark shape): one-launch kernels against each other, wall us per call (development)."""
import os, sys, time
os.environ.setdefault("TFX_ENV_DYNAMIC", "1")      # this tool flips TFX_* knobs inside one process
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torchfx_amd import torchfx_ext as E
x = torch.randn(2, 44100, device="cuda:0")
def wall(K, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    k = np.random.default_rng(0).standard_normal(K).astype(np.float32)
    for _ in range(20): E.fft_conv_forward(x, k, (K - 1, 0))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): E.fft_conv_forward(x, k, (K - 1, 0))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 300 * 1e6
    info = E.ols_plan_info(K, 44100, (K - 1, 0))
    for k_, v in old.items():
        if v is None: os.environ.pop(k_, None)
        else: os.environ[k_] = v
    return dt, info["path"], info["N"]
for K in (1025, 1500, 2048, 3000, 4096, 5000, 8192):
    for name, env in (("default", {}), ("no 8k", {"TFX_OLS_LDS8K_MINK": "0"}), ("r4 16k", {"TFX_OLS_LDS16K_R4": "2", "TFX_FFT_LOG2N": "14"}),
                      ("1024-thread 16k", {"TFX_OLS_LDS16K_R4": "0", "TFX_OLS_LDS16K": "2", "TFX_FFT_LOG2N": "14"})):
        for rep in range(2):
            print(K, name, "%.1f us  %s N=%d" % wall(K, env), flush=True)
