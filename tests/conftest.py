import os
import sys

import numpy as np
import pytest

os.environ.setdefault("TFX_ENV_DYNAMIC", "1")      # tests flip TFX_* knobs inside one process: the library re-reads them per call
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # a gpu test started without a device is a setup error, not a silent pass
    try:
        import torch
        has = torch.cuda.is_available()
    except Exception:
        has = False
    if has:
        return
    skip = pytest.mark.skip(reason="no ROCm device")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _built_artifacts():
    """The shared libraries are git-ignored build products: build them if this checkout does not
    have them yet (hipcc cross-compiles gfx950 without a GPU; ~2 minutes once)."""
    import subprocess
    import glob
    lib = os.path.join(ROOT, "torchfx_amd", "libtorchfx_hip.so")
    ext = glob.glob(os.path.join(ROOT, "torchfx_amd", "native", "torchfx_ext*.so"))
    if not os.path.exists(lib) or not ext:
        subprocess.check_call(["make", "-s", "-j", "8", "-C", os.path.join(ROOT, "torchfx_amd", "csrc")])
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        return cache[name]

    return load


@pytest.fixture()
def oracle_backend(monkeypatch):
    """Route torchfx_amd.torchfx_ext to the CPU oracle so HOST logic (planner, shapes, state
    rules, error behaviour, sharding) can be tested without a GPU.  Test-only: the product has
    no such path."""
    from tests import _fake_backend
    from torchfx_amd import torchfx_ext

    for name in ("sos_forward", "sos_bank_forward", "sos_bank_sum_forward", "biquad_forward", "fir_direct_forward", "fft_conv_forward",
                 "fir_stream_forward", "chunk_forward", "chunk_supported", "quantile_abs", "normalize_apply", "sum_forward", "delay_line_forward", "gain_forward", "stat_forward", "normalize_forward"):
        monkeypatch.setattr(torchfx_ext, name, getattr(_fake_backend, name))
    return _fake_backend


# ---- fixtures shared by the -m gpu parity files ------------------------------------------------------------
# 4 = LC 64 (the float64 default that ships), 2 = LC 32 + prefetch (the float32 default); 5 = LC 64 + prefetch
@pytest.fixture(params=[0, 1, 2, 3, 4, 5], ids=["lc32", "lc16", "lc32pf", "lc16pf", "lc64", "lc64pf"])
def sos_variant(request, monkeypatch):
    monkeypatch.setenv("TFX_SOS_VARIANT", str(request.param))
    return request.param


@pytest.fixture(params=["dispatch", "mfma"])
def fir_kernel(request, monkeypatch):
    """The direct FIR has two float32 kernels: the exact-f32 MFMA Toeplitz kernel (throughput) and the plain LDS-tiled
    one (short rows, launches whose tiles are all resident at once).  "dispatch" = the library's choice (mostly the plain
    kernel at test sizes), "mfma" = the MFMA kernel wherever it can run."""
    if request.param == "mfma":
        monkeypatch.setenv("TFX_FIR_ONE_ROUND_TILES", "0")
        monkeypatch.setenv("TFX_FIR_MFMA_MIN_T", "0")
    return request.param
