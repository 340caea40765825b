"""GPU parity -- overlap-save FFT convolution (row a14): rocFFT path and the hand-written LDS-FFT passes.
Tolerances and helpers: tests/gpu_common.py."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.gpu_common import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


def test_fftconv_golden(golden):
    g = golden("fftconv")
    x = dev(g["x"])
    for K in (64, 4097):
        close(ext().fft_conv_forward(x, g[f"k{K}"], (K - 1, 0)), g[f"y{K}"], TOL_CONV_F32, f"K={K}")
    close(ext().fft_conv_forward(x, g["k16"], (8, 7)), g["y16_pad87"], TOL_CONV_F32, "pad (8,7)")


def test_fftconv_65536_golden(golden, monkeypatch):
    g = golden("fftconv")
    kf = reverb_ir()[::-1].copy()
    for lg in ("0", "17", "19"):
        monkeypatch.setenv("TFX_FFT_LOG2N", lg)
        y = ext().fft_conv_forward(dev(g["x_long"]), kf, (65535, 0))
        close(y, g["y_long"], TOL_CONV_F32, f"log2N={lg}")


def test_fftconv_slabbing(monkeypatch):
    """Channel slabs (bounded workspace) give the same result as one slab."""
    x = dev(rnd((6, 70000), 1))
    k = rnd((300,), 2)
    monkeypatch.setenv("TFX_FFT_WS_MB", "4096")
    y1 = ext().fft_conv_forward(x, k, (299, 0))
    monkeypatch.setenv("TFX_FFT_WS_MB", "1")
    y2 = ext().fft_conv_forward(x, k, (299, 0))
    assert torch.equal(y1, y2)


@pytest.mark.parametrize("C,T,K", [(1, 70000, 4096), (3, 200001, 9000), (2, 300000, 16384),
                                   (1, 262144, 65536), (3, 600000, 65536), (2, 700003, 66559),
                                   (5, 400000, 40000), (2, 2500000, 65536), (3, 1100000, 5000),
                                   (2, 200000, 1024), (1, 70000, 16), (3, 150016, 100), (1, 2_200_000, 140_000)])
def test_native_ols_vs_rocfft_and_f64(C, T, K, monkeypatch):
    """The hand-written four-step pipeline (two frames per complex FFT) against the rocFFT path
    and against a float64 FFT convolution; odd frame counts leave an unpaired frame.  The last case (140 000 taps) takes
    the 2^20-point block on its own -- whose filter spectrum is computed ON THE DEVICE in float32 by the pipeline's forward
    kernels (olsnative.hip, `TFX_OLS_GPU_SPECTRUM`), like the reference's own float32 `rfft` of the kernel
    (_fftconv.py:123-124): same 4e-6 of the float64 convolution as the host-float64 spectra of the smaller blocks."""
    from scipy.signal import fftconvolve
    rng = np.random.default_rng(K + T)
    k = (rng.standard_normal(K) * np.exp(-np.arange(K) / (K / 6))).astype(np.float32)
    k /= np.abs(k).sum()
    x = rnd((C, T), T)
    xd = dev(x)
    monkeypatch.setenv("TFX_OLS_NATIVE", "1")
    yn = ext().fft_conv_forward(xd, k[::-1].copy(), (K - 1, 0))
    monkeypatch.setenv("TFX_OLS_NATIVE", "0")
    yr = ext().fft_conv_forward(xd, k[::-1].copy(), (K - 1, 0))
    exp = np.stack([fftconvolve(x[c].astype(np.float64), k.astype(np.float64))[:T] for c in range(C)])
    close(yn, exp.astype(np.float32), 4e-6, "native vs f64")
    close(yr, exp.astype(np.float32), 4e-6, "rocfft vs f64")
    # and the native path is really the one that ran: different rounding than rocFFT
    monkeypatch.setenv("TFX_OLS_PAIRS_PER_SLAB", "3")
    monkeypatch.setenv("TFX_OLS_NATIVE", "1")
    yn2 = ext().fft_conv_forward(xd, k[::-1].copy(), (K - 1, 0))
    assert torch.equal(yn, yn2)          # slab size does not change results
    # every block size the native path implements (256 x {256, 1024, 4096})
    for lg in (16, 18, 20, 21):                          # 21: 256 x 8192 (ols_row8192_kernel)
        if (1 << lg) >= 2 * K and T + K - 1 >= (1 << lg):
            monkeypatch.setenv("TFX_FFT_LOG2N", str(lg))
            yl = ext().fft_conv_forward(xd, k[::-1].copy(), (K - 1, 0))
            close(yl, exp.astype(np.float32), 4e-6, f"native log2N={lg} vs f64")
    monkeypatch.setenv("TFX_OLS_ROW_R4", "1")          # the radix-4 row pass stays as a cross-check
    monkeypatch.setenv("TFX_FFT_LOG2N", "18")
    if (1 << 18) >= 2 * K and T + K - 1 >= (1 << 18):
        close(ext().fft_conv_forward(xd, k[::-1].copy(), (K - 1, 0)), exp.astype(np.float32), 4e-6, "radix-4 rows")


def test_native_ols_padding_variants(monkeypatch):
    monkeypatch.setenv("TFX_OLS_NATIVE", "1")
    K = 5000
    k = rnd((K,), 1)
    x = rnd((2, 150000), 2)
    for pad in ((0, 0), (K - 1, 0), (100, 77), (0, K)):
        y = ext().fft_conv_forward(dev(x), k, pad)
        e = O.fft_conv1d(x.astype(np.float64), k.astype(np.float64), pad)
        close(y, e.astype(np.float32), 2e-5, f"pad={pad}")


@pytest.mark.parametrize("seed", range(24))
def test_fft_conv_random_geometry_vs_float64(seed):
    """Random (C, T, K, left/right padding) through every overlap-save geometry decision (native vs
    rocFFT path, block size, aligned / unaligned frames, ragged last block) against a float64
    correlation computed with SciPy."""
    from scipy.signal import fftconvolve
    rng = np.random.default_rng(7000 + seed)
    C = int(rng.integers(1, 5))
    K = int(rng.choice([1, 2, 15, 16, 17, 100, 1000, 4097, 20000, 70000]))
    T = int(rng.integers(max(1, K // 3), 400_000))
    pl = int(rng.choice([0, K - 1, int(rng.integers(0, K + 40))]))
    pr = int(rng.choice([0, 0, int(rng.integers(0, 50))]))
    if T + pl + pr < K:
        pl = K - T
    x = rnd((C, T), seed)
    kf = (rng.standard_normal(K) / np.sqrt(K)).astype(np.float32)
    y = ext().fft_conv_forward(dev(x), kf, (pl, pr))
    xp = np.pad(x.astype(np.float64), ((0, 0), (pl, pr)))
    exp = fftconvolve(xp, kf[::-1].astype(np.float64)[None], mode="valid", axes=-1)
    assert y.shape == exp.shape == (C, T + pl + pr - K + 1)
    close(y, exp.astype(np.float32), TOL_CONV_F32, f"C={C} T={T} K={K} pad=({pl},{pr})")


@pytest.mark.parametrize("seed", range(40))
def test_one_launch_kernels_random_geometry_vs_float64(seed):
    """The tap range of the one-launch kernels (1 ... 8192) with random rows, paddings (aligned rows take `lead` zero taps and a
    hop of whole cache lines), row counts that leave an unpaired last frame, both dtypes: whatever block (4096 / 8192 / 16 384
    points, either 16 384-point kernel) or fallback the dispatch picks, against a float64 correlation computed with SciPy."""
    from scipy.signal import fftconvolve
    rng = np.random.default_rng(9100 + seed)
    dtype = np.float64 if seed % 4 == 3 else np.float32
    C = int(rng.integers(1, 8))
    K = int(np.exp(rng.uniform(0, np.log(8192.0))))
    K = int(rng.choice([K, K, 639, 640, 699, 700, 1024, 2048, 2049, 3399, 3400, 4096, 4097, 8192])) if seed % 3 == 0 else K
    T = int(rng.integers(max(1, K // 2), 300_000))
    if seed % 2:
        T = (T + 31) // 32 * 32                                    # rows of whole cache lines: the aligned geometry
    pl = int(rng.choice([K - 1, 0, int(rng.integers(0, K + 40))]))
    pr = int(rng.choice([0, 0, int(rng.integers(0, 50))]))
    if T + pl + pr < K:
        pl = K - T
    x = rnd((C, T), seed, dtype)
    kf = (rng.standard_normal(K) / np.sqrt(K)).astype(np.float32).astype(dtype)
    info = ext().ols_plan_info(K, T, (pl, pr), torch.float32 if dtype == np.float32 else torch.float64)
    y = ext().fft_conv_forward(dev(x), kf, (pl, pr))
    xp = np.pad(x.astype(np.float64), ((0, 0), (pl, pr)))
    exp = fftconvolve(xp, kf[::-1].astype(np.float64)[None], mode="valid", axes=-1)
    assert y.shape == exp.shape == (C, T + pl + pr - K + 1)
    close(y, exp.astype(dtype), 4e-6 if dtype == np.float32 else TOL_CONV_F64,
          f"C={C} T={T} K={K} pad=({pl},{pr}) {dtype.__name__} path={info['path']} N={info['N']}")


def test_fft_conv_kernel_longer_than_the_native_limit():
    """600 001 taps is beyond the hand-written path (K <= 2^19): the rocFFT path takes over."""
    from scipy.signal import fftconvolve
    K, T = 600_001, 1_500_000
    assert not ext().ols_plan_info(K, T, (K - 1, 0))["native"]
    rng = np.random.default_rng(8)
    kf = (rng.standard_normal(K) * np.exp(-np.arange(K) / 90000.0) / 300).astype(np.float32)
    x = rnd((2, T), 4)
    y = ext().fft_conv_forward(dev(x), kf, (K - 1, 0))
    exp = fftconvolve(np.pad(x.astype(np.float64), ((0, 0), (K - 1, 0))), kf[::-1].astype(np.float64)[None], mode="valid", axes=-1)
    close(y, exp.astype(np.float32), TOL_CONV_F32, "600k taps")


# ---- one launch, transform in LDS (olslds.hip): K <= 2048, float32 and float64, rows of any length ----------
def _f64_corr(x, kf, pl, pr):
    from scipy.signal import fftconvolve
    xp = np.pad(x.astype(np.float64), ((0, 0), (pl, pr)))
    return fftconvolve(xp, kf[::-1].astype(np.float64)[None], mode="valid", axes=-1)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("C,T,K", [(2, 44100, 5), (2, 44100, 32), (2, 44100, 256), (2, 44100, 1024), (1, 3073, 1024),
                                   (3, 1000, 64), (1, 1, 1), (2, 7, 3), (5, 100_003, 2048), (4, 65_536, 1024),
                                   (64, 30_000, 1000), (1, 6146, 1024), (3, 12_288, 1025), (7, 9_217, 2047)])
def test_lds_ols_reference_shapes_vs_float64(C, T, K, dtype):
    """The reference's own test shapes for FFT-mode FIR and fft_conv1d (tests/test_fir.py:79-90,
    tests/test_fftconv.py:64-122: [2, 44100], K = 5 ... 1024) run on the single-launch LDS kernel -- not rocFFT --
    in float32 and float64, odd frame counts (an unpaired last frame), rows shorter than one block, T < K."""
    info = ext().ols_plan_info(K, T, (K - 1, 0), torch.float32 if dtype == np.float32 else torch.float64)
    big = K >= (640 if dtype == np.float32 else 700) and T + K - 1 >= 65536       # short rows: the smallest block that fits
    assert info["path"] == "lds" and info["N"] == (8192 if big else 4096)
    rng = np.random.default_rng(K * 7 + T)
    kf = (rng.standard_normal(K) / np.sqrt(K)).astype(np.float32).astype(dtype)      # taps are float32 values (fir.py:516)
    x = rnd((C, T), T + K, dtype)
    y = ext().fft_conv_forward(dev(x), kf, (K - 1, 0))
    assert y.dtype == (torch.float32 if dtype == np.float32 else torch.float64)
    close(y, _f64_corr(x, kf, K - 1, 0).astype(dtype), 4e-6 if dtype == np.float32 else TOL_CONV_F64, f"C={C} T={T} K={K}")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_lds_ols_padding_alignment_and_env_switch(dtype, monkeypatch):
    """Every padding flavour (aligned rows take `lead` zero taps and a hop of whole cache lines, unaligned ones do not),
    and the LDS kernel against the other two paths on the same input."""
    K = 1000
    kf = rnd((K,), 3, dtype)
    for T in (150_016, 150_001):
        x = rnd((3, T), T, dtype)
        for pad in ((0, 0), (K - 1, 0), (100, 77), (0, K), (999 + 33, 0)):
            y = ext().fft_conv_forward(dev(x), kf, pad)
            close(y, _f64_corr(x, kf, *pad).astype(dtype), 1e-5 if dtype == np.float32 else TOL_CONV_F64, f"T={T} pad={pad}")
    x = rnd((3, 150_016), 9, dtype)
    y = ext().fft_conv_forward(dev(x), kf, (K - 1, 0))
    monkeypatch.setenv("TFX_OLS_LDS", "0")
    assert ext().ols_plan_info(K, 150_016, (K - 1, 0))["path"] == "passes"
    y2 = ext().fft_conv_forward(dev(x), kf, (K - 1, 0))
    monkeypatch.setenv("TFX_OLS_NATIVE", "0")
    assert ext().ols_plan_info(K, 150_016, (K - 1, 0))["path"] == "rocfft"
    y3 = ext().fft_conv_forward(dev(x), kf, (K - 1, 0))
    tol = 4e-6 if dtype == np.float32 else 1e-12
    close(y, y2.cpu().numpy(), tol, "lds vs three passes / rocFFT (f64)")
    close(y, y3.cpu().numpy(), tol, "lds vs rocFFT")
    if dtype == np.float32:
        assert not torch.equal(y, y3)            # a different kernel really ran


@pytest.mark.parametrize("C,T,K", [(2, 44100, 2049), (3, 100_000, 4096), (1, 30_000, 8192), (5, 250_003, 5000), (2, 12_289, 4097),
                                   (64, 40_000, 3441), (1, 5000, 2500), (4, 98_304, 8000)])
def test_lds16k_ols_vs_float64(C, T, K, monkeypatch):
    """2048 < K <= 8192 (float32) at 16 384 points: the 1024-thread workgroup transform (fftpk16k.h; the default for
    4096 < K <= 8192 on rows shorter than 65 536 samples) and the radix-4 step around four 4096-point transforms in a 256-thread
    workgroup (the default on longer rows), each forced on every shape, against a float64 FFT convolution and against the
    three-pass pipeline / rocFFT.  K <= 4096 belongs to the 8192-point kernel by default."""
    short = T + K - 1 < 65536
    info = ext().ols_plan_info(K, T, (K - 1, 0))
    assert (info["path"], info["N"]) == (("lds", 16384) if (K > 4096 or (K >= 3400 and not short)) else ("lds", 8192))
    i64 = ext().ols_plan_info(K, T, (K - 1, 0), torch.float64)                               # float64: 8192 points up to 4096 taps; 16 384 would need 272 KB of LDS
    assert (i64["path"], i64["N"]) == (("lds", 8192) if K <= 4096 else ("rocfft", i64["N"]))
    rng = np.random.default_rng(K + T)
    kf = (rng.standard_normal(K) * np.exp(-np.arange(K) / (K / 5)) / np.sqrt(K)).astype(np.float32)
    x = rnd((C, T), T + K)
    exp = _f64_corr(x, kf, K - 1, 0).astype(np.float32)
    y_default = ext().fft_conv_forward(dev(x), kf, (K - 1, 0))
    close(y_default, exp, 4e-6, "default route")
    monkeypatch.setenv("TFX_FFT_LOG2N", "14")
    outs = {}
    for name, r4, wg in (("1024-thread workgroup", "0", "2"), ("radix 4 around 4096", "2", "1")):
        monkeypatch.setenv("TFX_OLS_LDS16K_R4", r4)
        monkeypatch.setenv("TFX_OLS_LDS16K", wg)
        assert ext().ols_plan_info(K, T, (K - 1, 0))["path"] == "lds" and ext().ols_plan_info(K, T, (K - 1, 0))["N"] == 16384
        outs[name] = ext().fft_conv_forward(dev(x), kf, (K - 1, 0))
        close(outs[name], exp, 4e-6, f"C={C} T={T} K={K} {name}")
        for pad in ((100, 77), (0, K)):
            if T + pad[0] + pad[1] >= K:
                close(ext().fft_conv_forward(dev(x), kf, pad), _f64_corr(x, kf, *pad).astype(np.float32), 4e-6, f"{name} pad={pad}")
    if C * T > 100:
        assert not torch.equal(outs["1024-thread workgroup"], outs["radix 4 around 4096"])       # two different kernels ran
    if K > 4096 or (K >= 3400 and not short):
        assert torch.equal(y_default, outs["1024-thread workgroup" if short else "radix 4 around 4096"])
    monkeypatch.delenv("TFX_FFT_LOG2N")
    monkeypatch.setenv("TFX_OLS_LDS16K", "0")
    monkeypatch.setenv("TFX_OLS_LDS16K_R4", "0")
    monkeypatch.setenv("TFX_OLS_LDS8K_MINK", "0")
    assert ext().ols_plan_info(K, T, (K - 1, 0))["path"] == ("rocfft" if short else "passes")
    close(ext().fft_conv_forward(dev(x), kf, (K - 1, 0)), exp, 4e-6, "three passes / rocFFT")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("C,T,K", [(2, 44100, 1025), (2, 44100, 2048), (3, 100_003, 4096), (1, 9000, 5), (5, 250_003, 3000), (2, 12_289, 2049),
                                   (64, 40_000, 1500), (1, 1, 1), (4, 98_304, 4000), (1, 8193, 4096), (3, 20_481, 4095)])
def test_lds8k_ols_vs_float64(C, T, K, dtype, monkeypatch):
    """The 8192-point block of the one-launch kernel (default for 640 <= K <= 4096 in float32 and 700 <= K <= 4096 in float64 on rows of 65 536 samples and more;
    TFX_FFT_LOG2N=13 forces it for smaller K): one radix-2 step in registers around two 4096-point transforms.  Against a
    float64 FFT convolution for every padding flavour, and against the 4096-point kernel / the three-pass pipeline / rocFFT on
    the same input."""
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    tol = 4e-6 if dtype == np.float32 else TOL_CONV_F64
    if K >= (640 if dtype == np.float32 else 700) and (T + K - 1 >= 65536 or K > 2048):
        info = ext().ols_plan_info(K, T, (K - 1, 0), tdt)            # the default route: 16 384 points take over at 3400 taps (float32, long rows)
        assert (info["path"], info["N"]) == ("lds", 16384 if (dtype == np.float32 and K >= 3400 and T + K - 1 >= 65536) else 8192)
    monkeypatch.setenv("TFX_FFT_LOG2N", "13")
    rng = np.random.default_rng(K * 3 + T)
    kf = (rng.standard_normal(K) / np.sqrt(K)).astype(np.float32).astype(dtype)
    x = rnd((C, T), T + K, dtype)
    for pad in ((K - 1, 0), (100, 77), (0, K)):
        if T + pad[0] + pad[1] < K:
            continue
        info = ext().ols_plan_info(K, T, pad, tdt)
        assert (info["path"], info["N"]) == ("lds", 8192)
        y = ext().fft_conv_forward(dev(x), kf, pad)
        assert y.dtype == tdt
        close(y, _f64_corr(x, kf, *pad).astype(dtype), tol, f"C={C} T={T} K={K} pad={pad}")
    y8 = ext().fft_conv_forward(dev(x), kf, (K - 1, 0))
    monkeypatch.delenv("TFX_FFT_LOG2N")
    monkeypatch.setenv("TFX_OLS_LDS8K_MINK", "0")
    other = ext().ols_plan_info(K, T, (K - 1, 0), tdt)
    assert other["N"] != 8192
    y_other = ext().fft_conv_forward(dev(x), kf, (K - 1, 0))
    close(y8, y_other.cpu().numpy(), 4e-6 if dtype == np.float32 else 1e-12, f"8192-point kernel vs {other['path']} N={other['N']}")
    if C * T > 100 and dtype == np.float32:
        assert not torch.equal(y8, y_other)          # a different kernel really ran


def test_lds_ols_plan_info_paths():
    e = ext()
    assert e.ols_plan_info(1024, 2_880_000, (1023, 0))["path"] == "lds"
    assert e.ols_plan_info(2048, 2_880_000, (2047, 0), torch.float64)["path"] == "lds"
    i = e.ols_plan_info(2049, 2_880_000, (2048, 0))                                     # up to 4096 taps: the 8192-point block, any row length
    assert (i["path"], i["N"]) == ("lds", 8192)
    i = e.ols_plan_info(4097, 2_880_000, (4096, 0))                                     # up to 8192 taps: 16 384 points in one launch
    assert (i["path"], i["N"]) == ("lds", 16384)
    assert e.ols_plan_info(3400, 2_880_000, (3399, 0))["N"] == 16384 and e.ols_plan_info(3399, 2_880_000, (3398, 0))["N"] == 8192
    assert e.ols_plan_info(8193, 2_880_000, (8192, 0))["path"] == "passes"
    i = e.ols_plan_info(8000, 44100, (7999, 0))                                         # short rows: one launch instead of rocFFT
    assert (i["path"], i["N"]) == ("lds", 16384)
    i = e.ols_plan_info(4096, 44100, (4095, 0))
    assert (i["path"], i["N"]) == ("lds", 8192)
    assert e.ols_plan_info(8193, 44100, (8192, 0))["path"] == "rocfft"
    i = e.ols_plan_info(2049, 2_880_000, (2048, 0), torch.float64)
    assert (i["path"], i["N"]) == ("lds", 8192) and abs(i["bytes_per_sample"] - (8 * 8192 / i["S"] + 8)) < 1e-9
    i = e.ols_plan_info(4097, 2_880_000, (4096, 0), torch.float64)                      # float64 beyond 4096 taps: three passes in float64 (round 6)
    assert (i["path"], i["N"]) == ("passes", 1 << 20) and abs(i["bytes_per_sample"] - (40 * (1 << 20) / i["S"] + 8)) < 1e-9
    assert e.ols_plan_info(4097, 500_000, (4096, 0), torch.float64)["path"] == "rocfft" # ... on rows of at least one 2^20-point block
    i = e.ols_plan_info(1024, 2_880_000, (1023, 0))                      # aligned rows: one lead tap, a hop of whole cache lines
    assert i["N"] == 8192 and i["S"] == 7168 and i["F"] == 402 and abs(i["bytes_per_sample"] - (4 * 8192 / 7168 + 4)) < 1e-9
    i = e.ols_plan_info(512, 2_880_000, (511, 0))
    assert i["N"] == 4096 and i["S"] == 3584 and i["F"] == 804 and abs(i["bytes_per_sample"] - (4 * 4096 / 3584 + 4)) < 1e-9
    i = e.ols_plan_info(1024, 2_880_001, (1023, 0))                      # unaligned rows: no lead, odd hop
    assert i["S"] == 8192 - 1024 + 1


@pytest.mark.parametrize("K,block", [(512, 4096), (1024, 8192), (3000, 8192), (3442, 16384), (4096, 16384), (8192, 16384)])
def test_lds_ols_many_rows_full_config(K, block):
    """cfg-3's shape through the FFT mode (64 x 2.88 M; 1024 taps = cfg 3's filter, 3442 = the default plan's fold of
    cfg 2's cascade into it, 4096 / 8192 = the larger one-launch blocks): channels {0, 31, 63} against float64."""
    from scipy.signal import firwin
    assert ext().ols_plan_info(K, 2_880_000, (K - 1, 0))["N"] == block
    k = firwin(K, 5000, fs=48000).astype(np.float32)
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.rand(64, 2_880_000, generator=g, device=DEV) * 2 - 1
    y = ext().fft_conv_forward(x, k[::-1].copy(), (K - 1, 0))
    for c in (0, 31, 63):
        xc = x[c].cpu().numpy()
        close(y[c:c + 1], _f64_corr(xc[None], k[::-1].copy(), K - 1, 0).astype(np.float32), 2e-6, f"row {c}")


# ---- float64 beyond 4096 taps: the three-pass pipeline in float64 (olsnative64.hip) ----------------------------
@pytest.mark.parametrize("C,T,K,off", [(1, 1 << 20, 4097, 0), (3, 2_500_003, 66559, 0), (2, 1_100_000, 20000, 5), (1, 3_300_017, 140_000, 0),
                                       (5, (1 << 20) + 12345, 9001, 3)])
def test_float64_three_pass_vs_float64_reference(C, T, K, off, monkeypatch):
    """The reference keeps a float64 signal float64 through FIR.forward (fir.py:526-579 -> _fftconv.py:70-141).  Above the
    one-launch kernels' 4096 taps the float64 three-pass pipeline runs (2^20-point blocks, any row length, any float64
    alignment, odd frame counts): against a float64 FFT convolution at 1e-11 and against the rocFFT path it replaces."""
    info = ext().ols_plan_info(K, T, (K - 1, 0), torch.float64)
    assert info["path"] == "passes" and info["N"] == 1 << 20 and info["bytes_per_sample"] < 60
    x = rnd((C, T), 31 + C, np.float64)
    k = np.random.default_rng(K).standard_normal(K) * np.exp(-np.arange(K) / (K / 6.0))
    kf = (k / np.abs(k).sum())[::-1].copy()
    buf = torch.zeros(C * T + 32, dtype=torch.float64, device=DEV)
    xv = buf[off: off + C * T].view(C, T)
    xv.copy_(torch.from_numpy(x))
    y = ext().fft_conv_forward(xv, torch.from_numpy(kf), (K - 1, 0))
    assert y.dtype == torch.float64 and tuple(y.shape) == (C, T)
    close(y, _f64_corr(x, kf, K - 1, 0), TOL_CONV_F64, f"float64 three passes C={C} T={T} K={K}")
    for pad in ((K - 1 + 7, 3), (0, 0), (K // 2, K // 2)):
        close(ext().fft_conv_forward(xv, torch.from_numpy(kf), pad), _f64_corr(x, kf, *pad), TOL_CONV_F64, f"pad={pad}")
    monkeypatch.setenv("TFX_OLS_NATIVE64", "0")
    ext().env_reload()
    try:
        assert ext().ols_plan_info(K, T, (K - 1, 0), torch.float64)["path"] == "rocfft"
        y2 = ext().fft_conv_forward(xv, torch.from_numpy(kf), (K - 1, 0))
    finally:
        monkeypatch.delenv("TFX_OLS_NATIVE64")
        ext().env_reload()
    close(y, y2.cpu().numpy(), TOL_CONV_F64, "three passes vs rocFFT (float64)")


def test_float64_fir_module_keeps_float64_on_long_taps():
    """`FIR.forward` on a float64 wave with a 20 000-tap kernel: float64 out, the float64 pipeline underneath, epilogue as passes."""
    from torchfx_amd import filter as F
    K, T = 20000, (1 << 20) + 4321
    k = np.random.default_rng(1).standard_normal(K) * np.exp(-np.arange(K) / 3000.0)
    k /= np.abs(k).sum()
    x = rnd((2, T), 77, np.float64)
    fir = F.FIR(k)
    y = fir(dev(x))
    assert y.dtype == torch.float64
    k32 = k.astype(np.float32).astype(np.float64)                # the module keeps its taps in float32 (fir.py:516), the signal stays float64
    close(y, _f64_corr(x, k32[::-1].copy(), K - 1, 0), TOL_CONV_F64, "FIR module, float64")
    ep = ext().Epilogue(gain=0.5, clamp=True, stat="absmax", per_row=True)
    y2 = ext().fft_conv_forward(dev(x), torch.from_numpy(k32[::-1].copy()), (K - 1, 0), epilogue=ep)
    exp = torch.clamp(y * 0.5, -1.0, 1.0)
    assert torch.equal(y2, exp)
    close(ep.stat_value, exp.abs().amax(dim=1).cpu().numpy(), 1e-12, "statistic")


# ---- rows are independent signals: a non-finite sample never reaches another row (round 6) ------------------------------------
@pytest.mark.parametrize("dtype,K,T", [(np.float32, 1024, 3 * 7168 - 500), (np.float32, 300, 5 * 3584 + 17), (np.float32, 6000, 3 * 10385 - 9),
                                       (np.float32, 20000, 2_600_000), (np.float32, 66559, 2_900_000), (np.float64, 2000, 3 * 6193 - 3),
                                       (np.float64, 20000, 2_600_000)],
                         ids=["lds8192", "lds4096", "lds16k", "passes-2^18", "passes-2^20", "lds8192-f64", "passes-f64"])
@pytest.mark.parametrize("bad", [float("nan"), float("inf")], ids=["nan", "inf"])
def test_non_finite_sample_stays_in_its_row(dtype, K, T, bad):
    """Two frames ride one complex transform.  When a row has an odd number of frames, the first frame of row c shares its
    transform with the last frame of row c - 1: a NaN / Inf in row c must not come out in row c - 1 (the reference convolves
    every row on its own, _fftconv.py:119-140).  Checked on every paired path: the rows before and after stay finite and equal to
    a clean run (to rounding: the real part of a complex product rounds differently when the imaginary operand changes), the
    poisoned row is non-finite around the sample and equal to the clean run where it is finite."""
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    info = ext().ols_plan_info(K, T, (K - 1, 0), tdt)
    assert info["path"] in ("lds", "passes")
    if info["F"] % 2 == 0:
        pytest.skip(f"even frame count {info['F']}: no pair straddles two rows at this geometry")
    x = rnd((3, T), 5, dtype)
    k = np.random.default_rng(K).standard_normal(K) * np.exp(-np.arange(K) / (K / 5.0))
    kf = torch.from_numpy((k / np.abs(k).sum()).astype(dtype)[::-1].copy())
    clean = ext().fft_conv_forward(dev(x), kf, (K - 1, 0)).cpu().numpy()
    xb = x.copy()
    xb[1, 37] = bad                                              # inside row 1's FIRST frame = the partner of row 0's last frame
    y = ext().fft_conv_forward(dev(xb), kf, (K - 1, 0)).cpu().numpy()
    tol = 2e-6 if dtype == np.float32 else 1e-13
    assert np.isfinite(y[0]).all(), "the non-finite sample of row 1 leaked into row 0 through the shared transform"
    close(y[0], clean[0], tol, "row 0 against the clean run")
    close(y[2], clean[2], tol, "row 2 against the clean run")
    assert not np.isfinite(y[1, 37:37 + K]).any(), "the samples the bad one reaches must be non-finite"
    fin = np.isfinite(y[1])
    assert fin[info["S"] + info["N"]:].all()
    close(y[1][fin], clean[1][fin], tol, "row 1 where it is finite")
    ep = ext().Epilogue(gain=1.0, stat="absmax", per_row=True)
    ext().fft_conv_forward(dev(xb), kf, (K - 1, 0), epilogue=ep)
    st = ep.stat_value.cpu().numpy()
    assert np.isnan(st[1]) and np.isfinite(st[0]) and np.isfinite(st[2])
