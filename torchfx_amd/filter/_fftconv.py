"""``fft_conv1d`` with the reference's signature and semantics
(``src/torchfx/filter/_fftconv.py:70-141``), executed by the HIP overlap-save op
(rocFFT + frame / spectrum-multiply / un-frame kernels).

``block_ratio`` is validated like the reference (``:116-117``) but does not choose the
block: the backend picks a power-of-two FFT length suited to MI355X; results agree to
float rounding because every block size computes the same causal linear convolution.
"""
from __future__ import annotations

from torch import Tensor


def fft_conv1d(x: Tensor, kernel: Tensor, padding: tuple[int, int] = (0, 0),
               block_ratio: float = 5.0) -> Tensor:
    """``x [B,C,T]``, ``kernel [1,1,K]`` (flipped taps, shared by all channels) ->
    ``[B,C,T+l+r-K+1]``."""
    from torchfx_amd import torchfx_ext

    batch, channels, time = x.shape
    ksize = kernel.shape[-1]
    length = time + int(padding[0]) + int(padding[1])
    if length < ksize:
        raise RuntimeError(
            f"Input should be at least as large as the kernel size {ksize}, "
            f"but it is only {length} samples long.")
    if block_ratio < 1:
        raise RuntimeError("Block ratio must be greater than 1.")
    y = torchfx_ext.fft_conv_forward(x.reshape(batch * channels, time), kernel.to(x.dtype), padding)
    return y.reshape(batch, channels, -1)


def pad_to(tensor: Tensor, target_length: int) -> Tensor:
    """Zero-pad the last dimension to ``target_length`` (reference ``_fftconv.py:23-29``).
    Host-side helper kept for API parity; the HIP op frames and pads inside its kernels."""
    extra = target_length - tensor.shape[-1]
    if extra == 0:
        return tensor
    out = tensor.new_zeros(*tensor.shape[:-1], target_length)
    out[..., : tensor.shape[-1]] = tensor
    return out


def unfold(x: Tensor, kernel_size: int, stride: int) -> Tensor:
    """Overlapping frames ``[*, F, kernel_size]`` with ``F = 1 + ceil((max(T, k) - k) / stride)``,
    the tail zero-padded (reference ``_fftconv.py:32-67``).  Kept for API parity only."""
    length = x.shape[-1]
    n_frames = -(-(max(length, kernel_size) - kernel_size) // stride) + 1
    padded = pad_to(x, (n_frames - 1) * stride + kernel_size)
    return padded.unfold(-1, kernel_size, stride)
