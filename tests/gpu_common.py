"""Shared helpers of the `-m gpu` parity tests (tests/test_gpu_*.py): the HIP path (through torchfx_ext -> C ABI of
libtorchfx_hip.so) is compared against the CPU oracle and the golden vectors generated from the real reference.


Stated tolerances (signals are scaled to max|x| <= 1; `scale` = max(1, max|expected|)):
  IIR, float64 arithmetic (default, what the reference does):
      float32 output : 1.5e-7 * scale   (one float32 ulp of the downcast; float64 sums may be
                                          associated differently than iir_cpu.cpp's -ffast-math build)
      float64 output : 2e-11 * scale ; states 2e-10 * scale
  IIR, float32 arithmetic (opt-in, TFX_PREC_F32): 5e-6 * scale on well-conditioned filters
  FIR direct / FFT convolution (float32): 1e-5 * scale   (reference's own bar is 1e-4:
      tests/test_fir.py:90, tests/test_fftconv.py:77) ; float64: 1e-11 * scale
"""
import numpy as np
import torch

TOL_IIR_F32OUT = 1.5e-7


TOL_IIR_F64OUT = 2e-11


TOL_STATE = 2e-10


TOL_IIR_F32MATH = 5e-6


TOL_CONV_F32 = 1e-5


TOL_CONV_F64 = 1e-11


DEV = "cuda:0"


def ext():
    from torchfx_amd import torchfx_ext
    return torchfx_ext


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def close(got, exp, tol, what=""):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    exp = np.asarray(exp)
    assert got.shape == exp.shape, f"{what}: shape {got.shape} != {exp.shape}"
    if exp.size == 0:
        return
    scale = max(1.0, float(np.abs(exp).max()))
    err = float(np.abs(got.astype(np.float64) - exp.astype(np.float64)).max())
    assert np.isfinite(err) and err <= tol * scale, f"{what}: max err {err:.3e} > {tol * scale:.3e}"


def rnd(shape, seed, dtype=np.float32):
    g = np.random.default_rng(seed)
    x = g.standard_normal(shape)
    return (x / np.abs(x).max()).astype(dtype)


def reverb_ir(K=65536):
    ir = np.random.default_rng(0).standard_normal(K) * np.exp(-np.arange(K) / 8000.0)
    return (ir / np.abs(ir).sum()).astype(np.float32)
