"""GPU -- lifetime of the overlap-save plan caches: spectra and tables are owned by the plans that point into
them, a plan stays alive with the caller that got it until its launches are enqueued, eviction is per entry (least recently
used), and a filter's first use inside a stream capture is refused with a clear message instead of allocating there."""
import threading

import numpy as np
import pytest
import torch

from tests.gpu_common import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


def _filters(n, taps, seed):
    g = np.random.default_rng(seed)
    ks = g.standard_normal((n, taps)) * np.exp(-np.arange(taps) / (taps / 4.0))
    return [torch.from_numpy((k / np.abs(k).sum()).astype(np.float32)) for k in ks]


@pytest.mark.parametrize("taps,T", [(256, 70_000), (9000, 300_000)])
def test_two_threads_cycle_more_filters_than_the_cache_holds(taps, T):
    """70 distinct filters (the one-launch LDS kernels' cache holds 64; 9000 taps: 20 filters, the three-pass pipeline's
    cache holds 16) cycled by two host threads on two streams, every thread in its own order: bit-equal to the serial run."""
    E = ext()
    n = 70 if taps == 256 else 20
    filt = _filters(n, taps, 3)
    x = dev(rnd((2, T), 5))
    serial = [E.fft_conv_forward(x, k, (taps - 1, 0)).clone() for k in filt]
    torch.cuda.synchronize()
    out = [[None] * n, [None] * n]
    err = []

    def work(tid):
        try:
            s = torch.cuda.Stream()
            order = list(range(n)) if tid == 0 else list(range(n - 1, -1, -1))
            with torch.cuda.stream(s):
                for rep in range(3):
                    for i in order:
                        out[tid][i] = E.fft_conv_forward(x, filt[i], (taps - 1, 0))
            s.synchronize()
        except Exception as e:       # noqa: BLE001
            err.append(repr(e))
    th = [threading.Thread(target=work, args=(t,)) for t in (0, 1)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not err, err
    torch.cuda.synchronize()
    for tid in (0, 1):
        for i in range(n):
            assert torch.equal(out[tid][i], serial[i]), (tid, i)


def test_first_use_inside_a_capture_is_refused_and_a_warm_filter_captures():
    E = ext()
    x = dev(rnd((2, 70_000), 6))
    warm, cold = _filters(2, 300, 11)
    y_ref = E.fft_conv_forward(x, warm, (299, 0)).clone()          # first use outside any capture: spectrum uploaded
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            y = E.fft_conv_forward(x, warm, (299, 0))
        g.replay()
    s.synchronize()
    assert torch.equal(y, y_ref)
    with torch.cuda.stream(s):
        g2 = torch.cuda.CUDAGraph()
        with pytest.raises(RuntimeError, match="inside a stream capture"):
            with torch.cuda.graph(g2, stream=s):
                E.fft_conv_forward(x, cold, (299, 0))
    torch.cuda.synchronize()
    y2 = E.fft_conv_forward(x, cold, (299, 0))                      # ... and the library is fine afterwards
    assert torch.isfinite(y2).all()


def test_a_workspace_that_must_grow_inside_a_capture_is_refused():
    """A warm filter captures as long as its workspaces exist for the shapes captured; a capture that would have to ALLOCATE one
    (here: more rows than the warm-up call had) is refused with a clear message -- under PyTorch's allocator the block would
    belong to the graph's private pool and die with the graph while the library still holds it."""
    import numpy as np
    E = ext()
    K = 9001
    k = np.random.default_rng(3).standard_normal(K) * np.exp(-np.arange(K) / 1500.0)
    kf = torch.from_numpy((k / np.abs(k).sum()).astype(np.float32)[::-1].copy())
    x = dev(rnd((6, 300_000), 8))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        y_ref = E.fft_conv_forward(x[:1], kf, (K - 1, 0)).clone()   # warm: spectrum + a one-row workspace on this stream
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            y = E.fft_conv_forward(x[:1], kf, (K - 1, 0))
        g.replay()
        s.synchronize()
        assert torch.equal(y, y_ref)
        g2 = torch.cuda.CUDAGraph()
        with pytest.raises(RuntimeError, match="inside a stream capture"):
            with torch.cuda.graph(g2, stream=s):
                E.fft_conv_forward(x, kf, (K - 1, 0))              # six rows: the workspace would have to grow
    torch.cuda.synchronize()
    assert torch.isfinite(E.fft_conv_forward(x, kf, (K - 1, 0))).all()


def test_workspaces_live_in_torchs_allocator_and_shrink_under_pressure():
    """tfx_set_workspace_allocator: the overlap-save workspaces come from PyTorch's caching allocator -- counted
    by torch.cuda.memory_allocated, released by clear_caches + empty_cache -- and with most of the device taken by torch
    tensors the chain step either runs on smaller slabs (same numbers) or raises torch's OutOfMemoryError, never a raw HIP one."""
    import numpy as np
    from tests.gpu_common import DEV, ext, rnd
    E = ext()
    E.clear_caches()
    torch.cuda.empty_cache()
    from torchfx_amd import filter as F
    f1 = F.LoButterworth(2000, order=6, fs=48000)
    f2 = F.ParametricEQ(frequency=1000, q=2.0, gain=3.0, fs=48000)
    f1.compute_coefficients(); f2.compute_coefficients()
    sos = torch.cat([f1._sos, f2._sos])
    K = 20001
    k = np.random.default_rng(0).standard_normal(K) * np.exp(-np.arange(K) / 3000.0)
    kf = torch.from_numpy((k / np.abs(k).sum()).astype(np.float32)[::-1].copy())
    x = torch.from_numpy(rnd((24, 6 << 20), 3)).to(DEV)             # 24 x 7 frames of 2^20 points = 84 pairs
    base = torch.cuda.memory_allocated()
    assert E.workspace_bytes() == 0
    y_ref = E.sos_fft_conv_forward(x, sos, kf, (K - 1, 0), force_block=1)
    held = E.workspace_bytes()
    assert held >= 84 * (1 << 23) // 3                                # three lanes of 28 pairs x 8 MB
    assert torch.cuda.memory_allocated() - base >= held               # ... counted by torch (plus y)
    E.clear_caches()
    ybytes = y_ref.numel() * 4
    assert E.workspace_bytes() == 0 and torch.cuda.memory_allocated() - base - ybytes < held // 4
    torch.cuda.empty_cache()
    # take all of the device with torch tensors but the output and ~400 MB: the slabs must shrink (84 pairs need 700 MB)
    free, _ = torch.cuda.mem_get_info()
    hog = torch.empty(max(0, free - ybytes - (400 << 20)), dtype=torch.uint8, device=DEV)
    try:
        y = E.sos_fft_conv_forward(x, sos, kf, (K - 1, 0), force_block=1)
        assert E.workspace_bytes() < held
        assert torch.equal(y, y_ref)                                  # the slab size never changes a sample
    except torch.OutOfMemoryError:
        pass                                                          # torch's own error type: acceptable, a raw HIP error is not
    finally:
        del hog
        E.clear_caches()
        torch.cuda.empty_cache()
