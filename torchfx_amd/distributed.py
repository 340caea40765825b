"""Multi-GPU execution of the filter hot path: channel sharding, one gather at the end.

The reference has no distributed code at all.  Every op on the hot
path treats the ``B*C`` rows independently (``iir_cpu.cpp:106``, grouped conv ``fir.py:568``,
broadcast spectrum multiply ``_fftconv.py:131``) and the coefficients are O(K) bytes, so rows
shard across ranks with NO data-path collective: rank r owns a contiguous block of rows, runs the
same plan on it, and -- only if the caller wants the result in one place -- a single
``torch.distributed.gather`` to the root moves the outputs (on backend "nccl" that is RCCL
grouped send/recv; each peer->root transfer rides that pair's own xGMI link, so never an
all-gather).  IIR state ``[K, C, 2]`` shards with the rows.

One process per GPU; the process group is whatever the caller initialised (tests use "gloo"
on CPU with the kernels replaced by the oracle).
"""
from __future__ import annotations

import torch
import torch.distributed as dist
from torch import Tensor, nn


def shard_bounds(n_rows: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous, balanced row blocks: the first ``n_rows % world`` ranks get one extra row."""
    base, extra = divmod(n_rows, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_rows(x: Tensor, group=None) -> Tensor:
    """This rank's block of rows of a ``[C, T]`` signal that every rank holds (or can index)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(x.shape[0], world, rank)
    return x[lo:hi]


def run_sharded(pipeline: nn.Module | list, x_local: Tensor, fs: int | None = None,
                fuse_fir: bool | None = None) -> Tensor:
    """Run a filter pipeline on this rank's rows (``fuse_fir=None``: the planner's default policy).
    No communication."""
    from torchfx_amd.wave import Wave

    w = Wave(x_local, fs if fs is not None else 0, device=x_local.device)
    if fuse_fir is not None:
        w.fuse_fir = fuse_fir
    steps = pipeline if isinstance(pipeline, (list, tuple)) else [pipeline]
    for s in steps:
        w = w | s
    return w.ys


def gather_rows(y_local: Tensor, n_rows: int, dst: int = 0, group=None) -> Tensor | None:
    """Single gather of the per-rank row blocks to ``dst`` -> ``[n_rows, T]`` there, None elsewhere.
    Blocks may differ by one row; they are padded to the largest block for the collective."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = [shard_bounds(n_rows, world, r) for r in range(world)]
    big = max(hi - lo for lo, hi in sizes)
    T = y_local.shape[1]
    send = y_local
    if y_local.shape[0] < big:
        send = torch.zeros((big, T), dtype=y_local.dtype, device=y_local.device)
        send[: y_local.shape[0]] = y_local
    send = send.contiguous()
    # "nccl" (= RCCL) moves device buffers peer to root directly; a gloo group (CPU collectives: the
    # development set-up where several ranks share one GPU) stages device rows through host memory
    staged = send.is_cuda and dist.get_backend(group) == "gloo"
    wire = send.cpu() if staged else send
    bufs = [torch.empty_like(wire) for _ in range(world)] if rank == dst else None
    dist.gather(wire, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    out = torch.cat([bufs[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)], dim=0)
    return out.to(y_local.device) if staged else out


def filter_sharded(pipeline, x: Tensor, fs: int, gather: bool = True, dst: int = 0, group=None,
                   fuse_fir: bool | None = None) -> Tensor | None:
    """shard -> run -> (optionally) gather.  ``x`` is the full ``[C, T]`` signal (each rank only
    touches its own rows)."""
    y = run_sharded(pipeline, shard_rows(x, group), fs, fuse_fir)
    return gather_rows(y, x.shape[0], dst, group) if gather else y
