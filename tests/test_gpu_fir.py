"""GPU parity -- direct FIR (row a13) and the stateful streaming FIR.
Tolerances and helpers: tests/gpu_common.py."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.gpu_common import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("K", [5, 32, 1024])
def test_fir_golden(golden, K, fir_kernel):
    g = golden("fir")
    x = dev(g["x"])
    close(ext().fir_direct_forward(x, g[f"k{K}"]), g[f"direct{K}"], TOL_CONV_F32, "direct")
    close(ext().fft_conv_forward(x, g[f"k{K}"], (K - 1, 0)), g[f"fft{K}"], TOL_CONV_F32, "fft")


def test_fir_short_and_f64_golden(golden, fir_kernel):
    g = golden("fir")
    close(ext().fir_direct_forward(dev(g["xs"]), g["ks"]), g["ys_direct"], TOL_CONV_F32)
    close(ext().fft_conv_forward(dev(g["xs"]), g["ks"], (31, 0)), g["ys_fft"], TOL_CONV_F32)
    close(ext().fir_direct_forward(dev(g["xt"]), g["kt"]), g["yt_direct"], TOL_CONV_F64)     # T < K, f64
    close(ext().fft_conv_forward(dev(g["xt"]), g["kt"], (63, 0)), g["yt_fft"], TOL_CONV_F64)


@pytest.mark.parametrize("C,T,K", [(1, 1, 1), (2, 100, 3), (3, 5000, 1025), (1, 40000, 2500), (5, 16385, 64)])
def test_fir_direct_shapes_vs_f64(C, T, K, fir_kernel):
    from scipy.signal import lfilter
    rng = np.random.default_rng(K)
    b = (rng.standard_normal(K) / K).astype(np.float32)
    x = rnd((C, T), T + K)
    exp = lfilter(b.astype(np.float64), [1.0], x.astype(np.float64), axis=-1)
    y = ext().fir_direct_forward(dev(x), b[::-1].copy())
    close(y, exp.astype(np.float32), TOL_CONV_F32, "direct vs lfilter f64")
    y2 = ext().fft_conv_forward(dev(x), b[::-1].copy(), (K - 1, 0))
    close(y2, exp.astype(np.float32), TOL_CONV_F32, "fft vs lfilter f64")


@pytest.mark.parametrize("kc", [None, 128, 512, 1024])
@pytest.mark.parametrize("C,T,K", [(2, 777, 1), (1, 20000, 101), (2, 16384, 128), (1, 16500, 129), (2, 33000, 400),
                                   (1, 9000, 513), (1, 50000, 1024), (1, 20000, 1100), (1, 3, 700)])
def test_fir_direct_chunk_sizes_vs_oracle(C, T, K, kc, monkeypatch):
    """Every tap-chunk instantiation of the MFMA kernel (and the cost-based default) against the
    oracle's float32 direct form: tile edges, rows shorter than the filter, K on chunk borders."""
    if kc is not None:
        monkeypatch.setenv("TFX_FIR_KC", str(kc))
        monkeypatch.setenv("TFX_FIR_MFMA_MIN_T", "0")      # short rows and small launches too go through the MFMA kernel here
        monkeypatch.setenv("TFX_FIR_ONE_ROUND_TILES", "0")
    rng = np.random.default_rng(1000 * K + T)
    kf = (rng.standard_normal(K) / np.sqrt(K)).astype(np.float32)
    x = rnd((C, T), K * 7 + T)
    exp = O.fir_direct(x, kf)
    y = ext().fir_direct_forward(dev(x), kf)
    close(y, exp, TOL_CONV_F32, f"direct K={K} kc={kc}")


@pytest.mark.parametrize("direct", [True, False])
@pytest.mark.parametrize("K,chunks", [
    (33, [100, 7, 1500, 3393]),                    # short rows: LDS-tiled kernel / rocFFT path; a chunk shorter than K-1
    (700, [300, 5000, 200, 9000]),                 # history longer than a chunk (old history shifts through)
    (129, [20_000, 4_096, 30_000]),                # MFMA Toeplitz kernel (rows >= 4096) / rocFFT path
    (200, [70_000, 66_000, 131_072]),              # LDS-resident overlap-save path, one and two blocks
    (5000, [140_000, 70_001]),                     # long taps through the native path with history (16 384-point one-launch kernel)
    (3000, [100_000, 5_000, 2_000, 70_000]),       # 8192-point one-launch kernel, long and short chunks, history longer than a chunk
    (6000, [20_000, 9_000, 100_000]),              # 16 384 points: the 1024-thread workgroup on short chunks, the radix-4 kernel on long ones
    (1500, [3_000, 80_000, 1_000]),                # 4096 points on short chunks, 8192 on long ones
])
def test_fir_stream_forward_chunks_equal_one_shot(K, chunks, direct, fir_kernel):
    """tfx_fir_stream_forward: every chunk continues the previous one through a [C, K-1] history buffer the
    kernels read beside the chunk; concatenated outputs == one-shot float64 lfilter of the whole signal."""
    rng = np.random.default_rng(K)
    C, T = 3, sum(chunks)
    x = rng.standard_normal((C, T)).astype(np.float32)
    taps = (rng.standard_normal(K) / np.sqrt(K)).astype(np.float32)
    import scipy.signal as sg
    ref = sg.lfilter(taps.astype(np.float64), [1.0], x.astype(np.float64), axis=-1)
    kernel = torch.from_numpy(taps[::-1].copy())
    hist, outs, off = None, [], 0
    for n in chunks:
        y, hist = ext().fir_stream_forward(dev(x[:, off:off + n]), kernel, hist, direct)
        assert y.shape == (C, n) and hist.shape == (C, K - 1)
        lo = max(0, off + n - (K - 1))
        assert np.array_equal(hist.cpu().numpy()[:, (K - 1) - (off + n - lo):], x[:, lo:off + n])   # the new history
        outs.append(y)
        off += n
    close(torch.cat(outs, dim=1), ref.astype(np.float32), TOL_CONV_F32, f"streamed FIR K={K} direct={direct}")
