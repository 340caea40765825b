"""The file edge of the path: decoded audio <-> device tensors.

Reference: ``Wave.from_file`` / ``Wave.save`` (``src/torchfx/wave.py:406-576``) read and write through
``soundfile`` and transpose ``[frames, channels] <-> [channels, frames]`` on the host.  Here the
decoder's interleaved buffer is uploaded as it is -- in chunks, through two pinned staging buffers, on a
copy stream that runs ahead of the de-interleave kernel -- and the transposition (and, for 16-bit
PCM, the int16 -> float conversion, which halves the PCIe bytes) happens on the GPU.  The way back
interleaves on the device and downloads chunk by chunk.  No decoding is done here: ``soundfile`` stays
the codec, exactly as in the reference.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import Tensor

PCM16_SCALE = 1.0 / 32768.0          # libsndfile's normalisation of 16-bit PCM read as float


# Staging buffers live as long as the process: a streaming caller (StreamProcessor.process_file) uploads and downloads
# one chunk after the other, and fresh pinned / pageable host buffers per chunk are exactly the host-side churn that
# makes the driver hold the process's GPU queues (docs/HISTORY.md section 6.2).  One set per (device, geometry); a call leaves
# them idle (it waits for its last kernel / copy before returning).
_UP: dict = {}
_DOWN: dict = {}


def _upload_staging(dev: torch.device, tdt: torch.dtype, chunk: int, C: int):
    key = (str(dev), tdt, chunk, C)
    st = _UP.get(key)
    if st is None:
        if len(_UP) > 8:
            _UP.clear()
        st = _UP[key] = ([torch.empty((chunk, C), dtype=tdt, pin_memory=True) for _ in range(2)],
                         [torch.empty((chunk, C), dtype=tdt, device=dev) for _ in range(2)],
                         torch.cuda.Stream(dev), [torch.cuda.Event() for _ in range(2)], [torch.cuda.Event() for _ in range(2)])
    return st


def upload_interleaved(frames: np.ndarray, device, chunk_frames: int = 1 << 22) -> Tensor:
    """Host ``[F, C]`` float32 or int16 -> device planar float32 ``[C, F]``.

    H2D copies go through two pinned staging buffers on a side stream; chunk *i+1* is being copied
    while chunk *i* is de-interleaved on the caller's stream."""
    from torchfx_amd import torchfx_ext

    if frames.ndim != 2 or frames.dtype not in (np.float32, np.int16):
        raise ValueError(f"expected [frames, channels] float32 or int16, got {frames.shape} {frames.dtype}")
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("upload_interleaved: target must be a ROCm device")
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    F, C = frames.shape
    tdt = torch.float32 if frames.dtype == np.float32 else torch.int16
    out = torch.empty((C, F), dtype=torch.float32, device=dev)
    if F == 0:
        return out
    chunk = max(1, min(int(chunk_frames), F))
    src = torch.from_numpy(np.ascontiguousarray(frames))
    pinned, staged, copy, done_copy, done_use = _upload_staging(dev, tdt, chunk, C)
    compute = torch.cuda.current_stream(dev)
    # the staging blocks may have been used on another compute stream by an earlier call; that call waited for its
    # last kernel before it returned, so only this call's own ordering matters: copy stream after compute stream
    copy.wait_stream(compute)
    for i, f0 in enumerate(range(0, F, chunk)):
        b = i & 1
        n = min(chunk, F - f0)
        if i >= 2:
            done_use[b].synchronize()                    # the kernel that read staged[b] has finished
        pinned[b][:n].copy_(src[f0:f0 + n])              # host memcpy into pinned memory
        with torch.cuda.stream(copy):
            staged[b][:n].copy_(pinned[b][:n], non_blocking=True)
            done_copy[b].record(copy)
        compute.wait_event(done_copy[b])
        torchfx_ext.deinterleave_forward(staged[b][:n], out, frame_base=f0, scale=PCM16_SCALE)
        done_use[b].record(compute)
    for b in range(min(2, -(-F // chunk))):
        done_use[b].synchronize()                        # the staging buffers are idle again when we return
    return out


def download_interleaved(x: Tensor, chunk_frames: int = 1 << 22, out: np.ndarray | None = None) -> np.ndarray:
    """Device planar float32 ``[C, F]`` -> host interleaved ``[F, C]`` (what ``soundfile.write`` takes).  Interleaving
    runs on the device, chunks come back through one pinned buffer; ``out`` (a float32 array with at least ``F`` rows
    of ``C``) lets a streaming caller reuse its host buffer -- the first ``F`` rows are filled and returned."""
    from torchfx_amd import torchfx_ext

    if x.dim() != 2:
        raise ValueError(f"expected [channels, frames], got {tuple(x.shape)}")
    C, F = x.shape
    if out is None:
        out = np.empty((F, C), dtype=np.float32)
    elif out.dtype != np.float32 or out.ndim != 2 or out.shape[1] != C or out.shape[0] < F or not out.flags.c_contiguous:
        raise ValueError(f"out must be a C-contiguous float32 [>= {F}, {C}] array")
    res = out[:F]
    if F == 0:
        return res
    xf = x if x.dtype == torch.float32 else x.to(torch.float32)
    chunk = max(1, min(int(chunk_frames), F))
    key = (str(x.device), chunk, C)
    pin = _DOWN.get(key)
    if pin is None:
        if len(_DOWN) > 8:
            _DOWN.clear()
        pin = _DOWN[key] = torch.empty((chunk, C), dtype=torch.float32, pin_memory=True)
    for f0 in range(0, F, chunk):
        n = min(chunk, F - f0)
        pin[:n].copy_(torchfx_ext.interleave_forward(xf, f0, n), non_blocking=True)
        torch.cuda.current_stream(x.device).synchronize()
        res[f0:f0 + n] = pin[:n].numpy()
    return res
