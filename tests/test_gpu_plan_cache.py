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
