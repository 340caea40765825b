"""PyTorch custom ops of the HIP backend: ``torch.ops.torchfx_hip.*``.

north_star asks for the kernels to be "exposed to Python through PyTorch-ROCm custom ops".  The ops are
defined and implemented in the compiled extension (``TORCH_LIBRARY(torchfx_hip)`` in
``torchfx_amd/csrc/ext/torchfx_ext.cpp``: CUDA-key kernels on the current HIP stream, Meta kernels for
shape inference, an explicit error for CPU tensors); importing this module loads it.  Forward-only,
like the reference (every forward there is ``@torch.no_grad()``).
"""
from torchfx_amd import native

native.load()
