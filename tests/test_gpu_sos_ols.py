"""GPU parity -- the SOS cascade inside the forward column pass of the three-pass overlap-save pipeline
(`tfx_sos_fft_conv_forward`, `ols_col_fwd16_sos_kernel`): `iir-cascade | FIR` in the reference's own arithmetic
(float64 DF1 recursion, one rounding to float32, float32 overlap-save) without a pass of the recursion's own.

The cascade is compared SECTION BY SECTION with the reference's float64 section outputs (`iir_cfg2_sections.npz`,
written by oracle/make_golden.py from the real reference) through the optional `y_sections` tap, the result with the
oracle's staged chain (iir_cpu.cpp -> float32 -> _fftconv.py) and with the staged HIP path.
Tolerances and helpers: tests/gpu_common.py."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.gpu_common import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu

CFG2_SOS = None


def cfg2_sos():
    global CFG2_SOS
    if CFG2_SOS is None:
        from torchfx_amd import filter as F
        f1 = F.LoButterworth(2000, order=6, fs=48000)
        f2 = F.ParametricEQ(frequency=1000, q=2.0, gain=3.0, fs=48000)
        f1.compute_coefficients()
        f2.compute_coefficients()
        CFG2_SOS = np.vstack([f1._sos.numpy(), f2._sos.numpy()])
    return CFG2_SOS


def taps(K, seed=3):
    k = np.random.default_rng(seed).standard_normal(K) * np.exp(-np.arange(K) / max(8.0, K / 8.0))
    return (k / np.abs(k).sum()).astype(np.float32)


def test_supported_predicate():
    E = ext()
    sos = cfg2_sos()
    assert E.sos_fft_conv_supported(28_800_000, sos, 66559, (66558, 0))
    assert E.sos_fft_conv_supported(28_800_001, sos, 66559, (66558, 0))              # any row length (round 6: row_shift)
    assert E.sos_fft_conv_supported(28_799_993, sos, 66559, (66558, 0))
    assert not E.sos_fft_conv_supported(2_880_000, sos, 1024, (1023, 0))              # one-launch LDS territory
    assert E.sos_fft_conv_supported(8192, sos, 513, (512, 0), force_block=True)
    import scipy.signal as sg
    slow = sg.butter(2, 20 / 24000, "highpass", output="sos")                          # memory far longer than a row
    assert not E.sos_fft_conv_supported(28_800_000, slow, 66559, (66558, 0))
    nine = np.vstack([sos, sos, sos[:1]])
    assert not E.sos_fft_conv_supported(28_800_000, nine, 66559, (66558, 0))          # more than 8 sections


def test_cfg2_sections_golden_through_the_fused_pass(golden):
    """north_star: "IIR compared section-by-section" -- on the kernel that runs the recursion inside pass A."""
    g = golden("iir_cfg2_sections")
    k = taps(513)
    y, sec = ext().sos_fft_conv_forward(dev(g["x"]), g["sos"], torch.from_numpy(k[::-1].copy()), (512, 0),
                                        return_sections=True, force_block=True)
    assert sec.dtype == torch.float64 and tuple(sec.shape) == g["y_sections"].shape
    for s in range(g["sos"].shape[0]):
        close(sec[s], g["y_sections"][s], TOL_IIR_F64OUT, f"section {s} (fused pass A)")
    ref = O.fft_conv1d(g["y"], k[::-1].copy(), (512, 0))          # the reference's own float32 cascade output, then its FFT convolution
    close(y, ref, TOL_CONV_F32, "chain")
    y2 = ext().sos_fft_conv_forward(dev(g["x"]), g["sos"], torch.from_numpy(k[::-1].copy()), (512, 0), force_block=True)
    assert torch.equal(y, y2)                                     # the tap instantiation computes the same samples


@pytest.mark.parametrize("block", [1, 2], ids=["2^20", "2^21"])
@pytest.mark.parametrize("C,T,K", [(1, 1 << 20, 8193), (3, 2_500_000 // 32 * 32, 66559), (2, 1_100_000 // 32 * 32, 20000),
                                   (1, 4_500_000 // 32 * 32, 66559)])
def test_multi_frame_against_oracle_and_staged(C, T, K, block):
    """Several frames per row, an odd number of frames (a pair without a second frame), frames that start in the left
    padding and end beyond the row; both block sizes (rows of 4096 and of 8192 samples)."""
    sos = cfg2_sos()
    x = rnd((C, T), 11)
    k = taps(K)
    kf = torch.from_numpy(k[::-1].copy())
    y = ext().sos_fft_conv_forward(dev(x), sos, kf, (K - 1, 0), force_block=block)
    assert tuple(y.shape) == (C, T)
    # staged HIP path: cascade kernel (float64 arithmetic, float32 out) then the overlap-save pipeline
    ys, _, _ = ext().sos_forward(dev(x), None, torch.from_numpy(sos), None, None)
    ys = ext().fft_conv_forward(ys, kf, (K - 1, 0))
    close(y, ys.cpu().numpy(), 2e-6, "fused vs staged HIP")
    # oracle on the first row (seconds of CPU)
    ref = O.chain_forward(x[:1], sos, [k[::-1].copy()])
    close(y[:1], ref, TOL_CONV_F32, "fused vs oracle")


@pytest.mark.parametrize("block", [1, 2], ids=["2^20", "2^21"])
def test_sections_on_a_multi_frame_row(block):
    """Section taps over rows longer than a frame: every sample of every section against the oracle's float64 recursion
    (rows of frames that overlap by K - 1 samples are written by both frames -- same values to float64 round-off)."""
    sos = cfg2_sos()
    T = (block << 20) + 300_000 // 32 * 32
    x = rnd((2, T), 5)
    k = taps(66559)
    y, sec = ext().sos_fft_conv_forward(dev(x), sos, torch.from_numpy(k[::-1].copy()), (66558, 0), return_sections=True,
                                        force_block=block)
    _, _, _, ref = O.sos_forward(x.astype(np.float64), sos, sections=True)
    for s in range(sos.shape[0]):
        close(sec[s], ref[s], TOL_IIR_F64OUT, f"section {s}")


@pytest.mark.parametrize("block", [1, 2], ids=["2^20", "2^21"])
@pytest.mark.parametrize("nsec", [1, 2, 3, 5, 6, 8])
def test_every_section_count(nsec, block):
    """One kernel instantiation per section count (1 ... 8; up to 4 sections two workgroups per CU, above one): each against
    the staged HIP path and, section by section, against the oracle's float64 recursion."""
    import scipy.signal as sg
    rows = [sg.butter(2, 0.05 + 0.07 * i, "lowpass" if i % 2 == 0 else "highpass", output="sos")[0] for i in range(nsec)]
    sos = np.ascontiguousarray(np.vstack(rows))
    if not ext().sos_fft_conv_supported(1 << 20, sos, 9001, (9000, 0), force_block=block):
        pytest.skip("this cascade's memory is longer than a row")
    T = (block << 20) + 64_000
    x = rnd((2, T), 40 + nsec)
    k = taps(9001)
    kf = torch.from_numpy(k[::-1].copy())
    y, sec = ext().sos_fft_conv_forward(dev(x), sos, kf, (9000, 0), return_sections=True, force_block=block)
    y2 = ext().sos_fft_conv_forward(dev(x), sos, kf, (9000, 0), force_block=block)
    assert torch.equal(y, y2)
    _, _, _, ref = O.sos_forward(x.astype(np.float64), sos, sections=True)
    for s in range(nsec):
        close(sec[s], ref[s], TOL_IIR_F64OUT, f"section {s} of {nsec}")
    ys, _, _ = ext().sos_forward(dev(x), None, torch.from_numpy(sos), None, None)
    ys = ext().fft_conv_forward(ys, kf, (9000, 0))
    close(y, ys.cpu().numpy(), 2e-6, "fused vs staged HIP")


@pytest.mark.parametrize("seed", range(16))
def test_random_geometry_against_the_staged_path(seed):
    """Seeded fuzz: rows, row length (ANY length), taps, sections, block size and left padding drawn at random; the fused
    pipeline against cascade kernel + overlap-save (same arithmetic, different kernels) and, on one row, against the oracle."""
    import scipy.signal as sg
    rng = np.random.default_rng(1000 + seed)
    block = int(rng.integers(1, 3))
    C = int(rng.integers(1, 5))
    T = int(rng.integers((block << 20), (3 << 20) * block))
    if seed % 4 == 0:
        T -= T % 32                                            # a quarter of the draws keep whole 128-byte lines
    K = int(rng.integers(8193, 90_000))
    nsec = int(rng.integers(1, 9))
    rows = []
    for i in range(nsec):
        f = float(rng.uniform(0.03, 0.6))
        kind = ["lowpass", "highpass"][int(rng.integers(0, 2))]
        rows.append(sg.butter(2, f, kind, output="sos")[0])
    sos = np.ascontiguousarray(np.vstack(rows))
    extra = int(rng.integers(0, 4)) * 32 + (int(rng.integers(0, 32)) if seed % 3 == 0 else 0)   # more causal padding than K - 1 (output longer than the row)
    pad = (K - 1 + extra, 0)
    if not ext().sos_fft_conv_supported(T, sos, K, pad, force_block=block):
        pytest.skip("geometry not served (memory longer than a row)")
    x = rnd((C, T), 2000 + seed)
    k = taps(K, seed=seed)
    kf = torch.from_numpy(k[::-1].copy())
    y = ext().sos_fft_conv_forward(dev(x), sos, kf, pad, force_block=block)
    assert tuple(y.shape) == (C, T + extra)
    ys, _, _ = ext().sos_forward(dev(x), None, torch.from_numpy(sos), None, None)
    ys = ext().fft_conv_forward(ys, kf, pad)
    close(y, ys.cpu().numpy(), 2e-6, f"fused vs staged (block 2^{19 + block}... C={C} T={T} K={K} sections={nsec} pad+{extra})")
    yi, _, _ = O.iir_module_forward(x[:1], sos)
    ref = O.fft_conv1d(yi, k[::-1].copy(), pad)
    close(y[:1], ref, TOL_CONV_F32, "fused vs oracle")


@pytest.mark.parametrize("block", [1, 2], ids=["2^20", "2^21"])
@pytest.mark.parametrize("off,cut", [(0, 7), (5, 0), (3, 1), (31, 33)])
def test_rows_that_are_not_whole_lines(block, off, cut):
    """T % 32 != 0 and / or a base pointer inside a 128-byte line: rows shift their frame grid (row_shift) -- bit-equal
    outputs to the same rows served from an aligned copy is too much to ask (another frame grid = another rounding), so:
    against the staged HIP path, the oracle, and section by section."""
    sos = cfg2_sos()
    C, T = 3, (block << 20) + 200_000 - cut
    x = rnd((C, T), 77 + off)
    buf = torch.zeros(C * T + 64, dtype=torch.float32, device=DEV)
    xv = buf[off: off + C * T].view(C, T)
    xv.copy_(torch.from_numpy(x))
    assert xv.data_ptr() % 128 == 4 * off
    K = 20001
    k = taps(K)
    kf = torch.from_numpy(k[::-1].copy())
    y, sec = ext().sos_fft_conv_forward(xv, sos, kf, (K - 1, 0), return_sections=True, force_block=block)
    y2 = ext().sos_fft_conv_forward(xv, sos, kf, (K - 1, 0), force_block=block)
    assert torch.equal(y, y2) and tuple(y.shape) == (C, T)
    _, _, _, ref = O.sos_forward(x.astype(np.float64), sos, sections=True)
    for s in range(sos.shape[0]):
        close(sec[s], ref[s], TOL_IIR_F64OUT, f"section {s}")
    ys, _, _ = ext().sos_forward(xv, None, torch.from_numpy(sos), None, None)
    ys = ext().fft_conv_forward(ys, kf, (K - 1, 0))
    close(y, ys.cpu().numpy(), 2e-6, "fused vs staged HIP")
    close(y[1:2], O.chain_forward(x[1:2], sos, [k[::-1].copy()]), TOL_CONV_F32, "fused vs oracle")


def _first_poisoned_output(info, K, T, c, n, base_off=0):
    """First output sample of row c that the staged pair (same block size) returns non-finite when x[c, n] is: the first
    sample of the first frame whose window holds sample n."""
    N, S = info["N"], info["S"]
    lead = (32 - (K - 1) % 32) % 32
    pad_left = K - 1 + lead
    sh = (base_off + c * T) % 32 if (T % 32 or base_off) else 0
    f = 0
    while f * S - pad_left - sh + N <= n:
        f += 1
    return max(0, f * S - sh)


@pytest.mark.parametrize("block", [1, 2], ids=["2^20", "2^21"])
@pytest.mark.parametrize("bad", [float("nan"), float("inf"), float("-inf")], ids=["nan", "inf", "-inf"])
@pytest.mark.parametrize("where", ["row0", "midframe", "lastrow", "warmup_of_frame1", "second_frame", "tail"])
def test_non_finite_samples_poison_the_rest_of_their_row(block, bad, where):
    """iir_cpu.cpp:132-147: a non-finite sample never leaves the recursion state, so the cascade's output is non-finite from
    that sample to the end of its row -- and so is every block of the FFT convolution whose window reaches it.  The recursion
    inside pass A restarts per 4096 / 8192-sample row of the transform; its end-state flags + the fix-up pass give the same
    answer as the staged pair of launches: NaN from the first frame that holds the bad sample to the end of the row, other
    rows untouched, sections non-finite exactly where the oracle's float64 recursion is."""
    sos = cfg2_sos()
    N = 1 << (19 + block)
    C, T = 3, 3 * N + 12_345
    K = 30001
    info = ext().sos_fft_conv_plan_info(T, sos, K, (K - 1, 0), force_block=block)
    assert info is not None and info["N"] == N
    pad_left = K - 1 + (32 - (K - 1) % 32) % 32
    n = {"row0": 17, "midframe": N // 2 + 1001, "lastrow": N - pad_left - 100, "warmup_of_frame1": info["S"] - pad_left - 5,
         "second_frame": info["S"] + N // 3, "tail": T - 3}[where]
    x = rnd((C, T), 123)
    x[1, n] = bad
    k = taps(K)
    kf = torch.from_numpy(k[::-1].copy())
    y, sec = ext().sos_fft_conv_forward(dev(x), sos, kf, (K - 1, 0), return_sections=True, force_block=block)
    y2 = ext().sos_fft_conv_forward(dev(x), sos, kf, (K - 1, 0), force_block=block)
    y, y2, sec = y.cpu().numpy(), y2.cpu().numpy(), sec.cpu().numpy()
    t0 = _first_poisoned_output(info, K, T, 1, n)
    assert t0 <= n
    fin = np.isfinite(y)
    assert fin[0].all() and fin[2].all(), "other rows must stay finite"
    assert fin[1, :t0].all() and not fin[1, t0:].any(), f"row 1 must be non-finite exactly from sample {t0} (bad sample {n})"
    assert np.array_equal(np.isfinite(y2), fin) and np.array_equal(y2[fin], y[fin])
    # sections: non-finite exactly where the oracle's recursion is
    _, _, _, ref = O.sos_forward(x.astype(np.float64), sos, sections=True)
    for s in range(sos.shape[0]):
        assert np.array_equal(np.isfinite(sec[s]), np.isfinite(ref[s])), f"section {s}: non-finite in other places than the oracle"
        m = np.isfinite(ref[s])
        close(sec[s][m], ref[s][m], TOL_IIR_F64OUT, f"section {s} (finite part)")
        assert not np.isfinite(ref[s][1, n:]).any()
    # staged HIP pair: same places when it runs the same block size, never finite behind the bad sample
    ys, _, _ = ext().sos_forward(dev(x), None, torch.from_numpy(sos), None, None)
    ys = ext().fft_conv_forward(ys, kf, (K - 1, 0)).cpu().numpy()
    assert not np.isfinite(ys[1, n:]).any() and not np.isfinite(y[1, n:]).any()
    both = np.isfinite(ys) & fin
    close(y[both], ys[both], 2e-6, "fused vs staged where both are finite")
    # oracle (reference framing N = 5 K: other block boundaries, same rule)
    refc = O.chain_forward(x[1:2], sos, [k[::-1].copy()])
    assert not np.isfinite(refc[0, n:]).any()
    both = np.isfinite(refc[0]) & fin[1]
    close(y[1][both], refc[0][both], TOL_CONV_F32, "fused vs oracle where both are finite")


def test_non_finite_with_epilogue_statistic():
    """The peak an epilogue leaves for Normalize is NaN for a poisoned row (torch.max semantics), untouched for the others."""
    sos = cfg2_sos()
    T = (2 << 20) + 999
    x = rnd((2, T), 8)
    x[0, 1_500_000] = float("nan")
    k = taps(30000)
    kf = torch.from_numpy(k[::-1].copy())
    E = ext()
    ep = E.Epilogue(gain=0.5, clamp=True, stat="absmax", per_row=True)
    y = E.sos_fft_conv_forward(dev(x), sos, kf, (29999, 0), force_block=1, epilogue=ep)
    st = ep.stat_value.cpu().numpy()
    assert np.isnan(st[0]) and np.isfinite(st[1])
    assert np.isfinite(y[1].cpu().numpy()).all() and not np.isfinite(y[0, 1_500_000:].cpu().numpy()).any()


def test_cascade_without_a_unit_b0_form():
    """A section with b0 = 0 (a pure delay in the numerator) has no unit-b0 form: the kernel's plain instantiation runs."""
    sos = np.array([[0.0, 0.5, 0.25, 1.0, -0.6, 0.2], [0.3, 0.1, 0.0, 1.0, 0.4, 0.1]])
    T = (1 << 20) + 32_000
    x = rnd((1, T), 9)
    k = taps(9001)
    kf = torch.from_numpy(k[::-1].copy())
    y, sec = ext().sos_fft_conv_forward(dev(x), sos, kf, (9000, 0), return_sections=True, force_block=1)
    _, _, _, ref = O.sos_forward(x.astype(np.float64), sos, sections=True)
    for s in range(2):
        close(sec[s], ref[s], TOL_IIR_F64OUT, f"section {s}")
    close(y, O.chain_forward(x, sos, [k[::-1].copy()]), TOL_CONV_F32, "chain")


def test_epilogue_rides_along():
    sos = cfg2_sos()
    T = 1_200_000 // 32 * 32
    x = rnd((2, T), 7)
    k = taps(30000)
    kf = torch.from_numpy(k[::-1].copy())
    E = ext()
    ep = E.Epilogue(gain=0.5, clamp=True, stat="absmax", per_row=True)
    y = E.sos_fft_conv_forward(dev(x), sos, kf, (29999, 0), force_block=True, epilogue=ep)
    y0 = E.sos_fft_conv_forward(dev(x), sos, kf, (29999, 0), force_block=True)
    exp = torch.clamp(y0 * 0.5, -1.0, 1.0)
    assert torch.equal(y, exp)
    close(ep.stat_value, exp.abs().amax(dim=1).double().cpu().numpy(), 1e-12, "statistic")


def test_refuses_what_it_does_not_serve():
    import scipy.signal as sg
    slow = sg.butter(2, 20 / 24000, "highpass", output="sos")                          # memory far longer than a row of the transform
    x = dev(rnd((1, 8191), 1))
    with pytest.raises(RuntimeError, match="unsupported here"):
        ext().sos_fft_conv_forward(x, slow, torch.from_numpy(taps(65)), (64, 0), force_block=True)
    # a tiny odd row IS served when the block is forced (one frame, mostly padding) -- and right
    sos = cfg2_sos()
    k = taps(65)
    y = ext().sos_fft_conv_forward(x, sos, torch.from_numpy(k[::-1].copy()), (64, 0), force_block=True)
    close(y, O.chain_forward(x.cpu().numpy(), sos, [k[::-1].copy()]), TOL_CONV_F32, "one short odd row")
