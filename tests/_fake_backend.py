"""TEST-ONLY stand-in for torchfx_amd.torchfx_ext built on the CPU oracle (see the
``oracle_backend`` fixture).  Lets the host-side logic run on a machine without a GPU."""
import numpy as np
import torch

from oracle import oracle as O

calls = []


def _np(t):
    return None if t is None else t.detach().cpu().numpy()


def _epilogue(y, ep):
    """Gain / clamp / raw statistic of an Epilogue on the oracle's output (the test-side twin of epilogue.h)."""
    if ep is None:
        return y
    if ep.gain != 1.0 or ep.clamp:
        y = torch.from_numpy(np.ascontiguousarray(O.gain(y.numpy(), ep.gain, "amplitude", ep.clamp)))
    if ep.stat is not None:
        a = y.numpy().astype(np.float64)
        a = a.reshape(-1, a.shape[-1]) if ep.per_row else a.reshape(1, -1)
        ep.stat_value = torch.from_numpy(np.abs(a).max(axis=1) if ep.stat == "absmax" else (a * a).sum(axis=1))
    return y


def sos_forward(x, sos, sos_cpu, state_x, state_y, *, out_dtype=None, precision=None, return_sections=False, epilogue=None):
    calls.append(("sos_forward", tuple(x.shape), int((sos_cpu if sos_cpu is not None else sos).shape[0])) + (("ep",) if epilogue else ()))
    s = _np(sos_cpu if sos_cpu is not None else sos)
    y, sx, sy, sec = O.sos_forward(_np(x), s, _np(state_x), _np(state_y), sections=True)
    odt = x.dtype if out_dtype is None else out_dtype
    out = (_epilogue(torch.from_numpy(y).to(odt), epilogue), torch.from_numpy(sx), torch.from_numpy(sy))
    return out + (torch.from_numpy(sec).to(odt),) if return_sections else out


def sos_bank_forward(x, sos_banks, state_x, state_y, *, out_dtype=None, precision=None):
    banks = _np(sos_banks) if hasattr(sos_banks, "detach") else np.asarray(sos_banks)
    nb, k = banks.shape[0], banks.shape[1]
    c = x.shape[0]
    calls.append(("sos_bank_forward", tuple(x.shape), nb))
    ys, sxs, sys_ = [], [], []
    for b in range(nb):
        sx = None if state_x is None else _np(state_x)[:, b * c:(b + 1) * c]
        sy = None if state_y is None else _np(state_y)[:, b * c:(b + 1) * c]
        y, nx, ny = O.sos_forward(_np(x), banks[b], sx, sy)
        ys.append(y), sxs.append(nx), sys_.append(ny)
    odt = x.dtype if out_dtype is None else out_dtype
    return (torch.from_numpy(np.stack(ys)).to(odt), torch.from_numpy(np.concatenate(sxs, axis=1)),
            torch.from_numpy(np.concatenate(sys_, axis=1)))


def sos_bank_sum_forward(x, sos_banks, state_x, state_y, *, precision=None):
    banks = _np(sos_banks) if hasattr(sos_banks, "detach") else np.asarray(sos_banks)
    nb, c = banks.shape[0], x.shape[0]
    calls.append(("sos_bank_sum_forward", tuple(x.shape), nb))
    acc = torch.zeros_like(x)
    sxs, sys_ = [], []
    for b in range(nb):
        sx = None if state_x is None else _np(state_x)[:, b * c:(b + 1) * c]
        sy = None if state_y is None else _np(state_y)[:, b * c:(b + 1) * c]
        y, nx, ny = O.sos_forward(_np(x), banks[b], sx, sy)
        acc += torch.from_numpy(y).to(x.dtype)
        sxs.append(nx), sys_.append(ny)
    return acc, torch.from_numpy(np.concatenate(sxs, axis=1)), torch.from_numpy(np.concatenate(sys_, axis=1))


def biquad_forward(x, b, a1, a2, state_x, state_y, *, out_dtype=None, precision=None):
    calls.append(("biquad_forward", tuple(x.shape)))
    y, sx, sy = O.biquad_forward(_np(x), _np(b), a1, a2, _np(state_x), _np(state_y))
    odt = x.dtype if out_dtype is None else out_dtype
    return torch.from_numpy(y).to(odt), torch.from_numpy(sx), torch.from_numpy(sy)


def fir_direct_forward(x, kernel):
    calls.append(("fir_direct_forward", tuple(x.shape), int(kernel.numel())))
    k = _np(kernel).reshape(-1).astype(_np(x).dtype)
    return torch.from_numpy(O.fir_direct(_np(x), k))


def fft_conv_forward(x, kernel, padding=(0, 0), epilogue=None):
    calls.append(("fft_conv_forward", tuple(x.shape), int(kernel.numel())) + (("ep",) if epilogue else ()))
    k = _np(kernel).reshape(-1).astype(_np(x).dtype)
    return _epilogue(torch.from_numpy(np.ascontiguousarray(O.fft_conv1d(_np(x), k, padding))), epilogue)


def normalize_apply(x, stat, peak, mode=0, per_row=False):
    calls.append(("normalize_apply", tuple(x.shape), int(mode), bool(per_row)))
    a = _np(x)
    rows = a.reshape(-1, a.shape[-1]) if per_row else a.reshape(1, -1)
    st = _np(stat).astype(np.float64)
    s = (st if mode == 0 else np.sqrt(st / rows.shape[1])).astype(a.dtype)
    out = rows.copy()
    for r in range(rows.shape[0]):
        if s[r] > 0:
            out[r] = (rows[r] / s[r]) * a.dtype.type(peak)
    return torch.from_numpy(out.reshape(a.shape))


from torchfx_amd.torchfx_ext import Epilogue  # noqa: E402,F401  (plain Python class, no device code)


def fir_stream_forward(x, kernel, hist, direct=False):
    calls.append(("fir_stream_forward", tuple(x.shape), int(kernel.numel()), bool(direct)))
    xn = _np(x)
    k = _np(kernel).reshape(-1).astype(xn.dtype)
    h = np.zeros((xn.shape[0], k.size - 1), xn.dtype) if hist is None else _np(hist).astype(xn.dtype)
    xv = np.concatenate([h, xn], axis=1)                       # the oracle has no two-pointer form: concatenate here
    y = O.fir_direct(xv, k)[:, k.size - 1:] if direct else O.fft_conv1d(xv, k, (0, 0))
    return torch.from_numpy(np.ascontiguousarray(y)), torch.from_numpy(np.ascontiguousarray(xv[:, xv.shape[1] - (k.size - 1):]))


def quantile_abs(x, q):
    calls.append(("quantile_abs", tuple(x.shape)))
    return torch.quantile(torch.abs(x.reshape(-1)), float(q), interpolation="linear").to(torch.float64).reshape(1)


def chunk_supported(C, T, K, taps):
    return C >= 1 and 1 <= T <= 4096 and 0 <= K <= 64 and 1 <= taps <= 4096 and T * taps <= (1 << 22)


def chunk_forward(x, sos, state_x, state_y, kernel, hist, gain=None, clamp=False, *, precision=None):
    """The fused per-chunk launch, staged on the oracle: cascade -> stateful direct FIR -> gain / clip."""
    calls.append(("chunk_forward", tuple(x.shape), int(np.asarray(_np(sos)).shape[0]), int(kernel.numel())))
    K = int(_np(sos).shape[0])
    if K:
        u, nsx, nsy = sos_forward(x, None, sos, state_x, state_y)[:3]
    else:
        u, nsx, nsy = x, torch.zeros(0, x.shape[0], 2, dtype=torch.float64), torch.zeros(0, x.shape[0], 2, dtype=torch.float64)
    calls.pop() if K else None
    if kernel.numel() > 1:
        y, nh = fir_stream_forward(u, kernel, hist, True)
        calls.pop()
    else:
        y, nh = u * float(_np(kernel).reshape(-1)[0]), torch.zeros(x.shape[0], 0)
    if gain is not None:
        y = y * np.float32(gain)
    if clamp:
        y = y.clamp(-1.0, 1.0)
    return y, nsx, nsy, nh


def sum_forward(tensors):
    calls.append(("sum_forward", len(tensors)))
    out = torch.zeros_like(tensors[0])
    for t in tensors:
        out += t
    return out


def delay_line_forward(x, delay_samples, decay, mix):
    if x.shape[-1] <= delay_samples:
        return x
    return torch.from_numpy(O.delay_line(_np(x).reshape(-1, x.shape[-1]), delay_samples, decay, mix)).reshape(x.shape)


STAT_ABSMAX, STAT_RMS = 0, 1


def gain_forward(x, gain, clamp=False):
    calls.append(("gain_forward", tuple(x.shape), float(gain), bool(clamp)))
    return torch.from_numpy(np.ascontiguousarray(O.gain(_np(x), gain, "amplitude", clamp)))


def stat_forward(x, mode=STAT_ABSMAX, per_row=False):
    a = _np(x).astype(np.float64)
    a = a.reshape(-1, a.shape[-1]) if per_row else a.reshape(1, -1)
    out = np.abs(a).max(axis=1) if mode == STAT_ABSMAX else np.sqrt((a * a).mean(axis=1))
    return torch.from_numpy(out)


def normalize_forward(x, peak, mode=STAT_ABSMAX, per_row=False):
    calls.append(("normalize_forward", tuple(x.shape), int(mode), bool(per_row)))
    strat = "per_channel" if per_row else ("peak" if mode == STAT_ABSMAX else "rms")
    return torch.from_numpy(np.ascontiguousarray(O.normalize(_np(x), peak, strat)))
