"""Round-6 experiment: the cascade inside the ONE-LAUNCH overlap-save kernel (ols_lds8192_sos_kernel, force_block=3) -- parity
against the staged pair and the oracle's float64 sections, then the time on 64 x 2.88 M against the two launches."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torchfx_amd import torchfx_ext as E
from torchfx_amd import filter as F
from oracle import oracle as O

f1 = F.LoButterworth(2000, order=6, fs=48000); f2 = F.ParametricEQ(frequency=1000, q=2.0, gain=3.0, fs=48000)
f1.compute_coefficients(); f2.compute_coefficients()
sos = torch.cat([f1._sos, f2._sos])
for (C, T, K) in ((3, 200_000, 1024), (2, 100_001, 700), (5, 65_536, 2000), (1, 300_000, 4000)):
    rng = np.random.default_rng(K)
    x = rng.standard_normal((C, T)).astype(np.float32); x /= np.abs(x).max()
    k = rng.standard_normal(K) * np.exp(-np.arange(K) / (K / 6.0))
    kf = torch.from_numpy((k / np.abs(k).sum()).astype(np.float32)[::-1].copy())
    xd = torch.from_numpy(x).cuda()
    if not E.sos_fft_conv_supported(T, sos, K, (K - 1, 0), force_block=3):
        print("not served", C, T, K); continue
    y, sec = E.sos_fft_conv_forward(xd, sos, kf, (K - 1, 0), force_block=3, return_sections=True)
    y2 = E.sos_fft_conv_forward(xd, sos, kf, (K - 1, 0), force_block=3)
    ys, _, _ = E.sos_forward(xd, None, sos, None, None)
    ys = E.fft_conv_forward(ys, kf, (K - 1, 0))
    _, _, _, ref = O.sos_forward(x.astype(np.float64), sos.numpy(), sections=True)
    es = max(float(np.abs(sec[s].cpu().numpy() - ref[s]).max()) for s in range(4))
    print(f"C={C} T={T} K={K}: |fused - staged| {float((y - ys).abs().max()):.2e}  sections vs oracle {es:.2e}  taps-instantiation equal {bool(torch.equal(y, y2))}", flush=True)

C, T, K = 64, 2_880_000, 1024
x = torch.rand((C, T), device="cuda") * 2 - 1
k = np.random.default_rng(0).standard_normal(K) * np.exp(-np.arange(K) / 200.0)
kf = torch.from_numpy((k / np.abs(k).sum()).astype(np.float32)[::-1].copy())


def timed(fn, name):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(9):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): fn()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e3)
    ts.sort(); print(f"{name:34s} min {ts[0]:.4f} med {ts[4]:.4f} ms", flush=True)


def staged():
    y, _, _ = E.sos_forward(x, None, sos, None, None)
    return E.fft_conv_forward(y, kf, (K - 1, 0))


timed(staged, "two launches")
timed(lambda: E.sos_fft_conv_forward(x, sos, kf, (K - 1, 0), force_block=3), "one launch (fused)")
for wg in (2, 3, 4):
    os.environ["TFX_OLS_LDS_SOS_WG_PER_CU"] = str(wg)
    E.env_reload()
    timed(lambda: E.sos_fft_conv_forward(x, sos, kf, (K - 1, 0), force_block=3), f"  {wg} workgroups per CU assumed")
ya = E.sos_fft_conv_forward(x, sos, kf, (K - 1, 0), force_block=3)
print("64 x 2.88 M: |fused - staged| =", float((ya - staged()).abs().max()))
