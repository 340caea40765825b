"""``Wave``: the signal container with the lazy ``|`` pipeline and the fusion planner.

Reference: ``src/torchfx/wave.py`` -- only the hot-path part is mirrored: construction
(:100-140), lazy ``ys`` (:196-205), ``_materialize`` (:207-239: consecutive IIR/Biquad steps
become one ``FusedSOSCascade``; a *single* IIR step runs the module itself and therefore
keeps its state across waves), ``_deferred`` (:241-257), ``to`` (:275-332), ``__or__``
(:578-695: fs propagation + eager coefficient design, ``nn.Sequential`` flattened into
steps), ``__len__`` / ``channels``.  ``from_file`` / ``save`` (:406-576) keep ``soundfile`` as the codec
(imported lazily, as there) but, for a ROCm target, hand the decoder's interleaved buffer to
``torchfx_amd.io`` -- chunked pinned upload + device-side de-interleave.

Beyond the reference, the planner merges consecutive FFT-mode ``FIR`` steps into one overlap-save
pass (``fuse_fir``): convolution is associative, so ``FIR(b1) | FIR(b2)`` == ``FIR(b1 * b2)``; the
merged taps are computed on the host in float64.  A fresh IIR cascade in front of such a run joins it
in one of two ways.  ``fuse_recursive`` (default): the cascade keeps the reference's arithmetic -- float64
recursion, one rounding to float32 -- and runs INSIDE the overlap-save pipeline's forward column pass
(``filter.fused.CascadeFIR`` -> ``tfx_sos_fft_conv_forward``): no pass over the signal of its own, every
section's output still readable.  ``fuse_spectral`` (opt-in, ``TORCHFX_AMD_FUSE_SPECTRAL=1``): the cascade is
folded into the FIR run as its impulse response (``_spectral_plan``) -- one pass less again, but the IIR part
then runs in float32 FFT arithmetic; taken only when a host-side replay bounds the extra error by 2e-6 of the
output scale (``_fold_error_estimate``).  ``TORCHFX_AMD_FUSION=reference`` stages every step exactly as the
reference does; under every policy results stay within the FIR/FFT tolerance of the reference's staged
output (``tests/golden/chain*.npz``, 1e-5).

``fuse_gain=True`` (env ``TORCHFX_AMD_FUSE_GAIN=1``, also opt-in) folds a clamp-free ``Gain`` into
the coefficients of the filter run it touches -- scaling is linear, so ``iir | gain | iir`` stays one
cascade launch (the b-row of the next section is multiplied by g) and ``gain | fir`` is one FIR
with scaled taps; by default a ``Gain`` is its own streaming pass and splits IIR runs exactly as in
the reference (``tests/test_chain_fusion.py:102-121``).
"""
from __future__ import annotations

import os
import threading
import typing as tp
from collections import OrderedDict

import numpy as np
import torch
from torch import Tensor, nn

from torchfx_amd.effect import FX
from torchfx_amd.filter._base import AbstractFilter


def _fusion_defaults() -> tuple[bool, bool, bool, bool]:
    """(fuse_fir, fuse_spectral, fuse_gain, fuse_epilogue) of a new ``Wave``.

    ``TORCHFX_AMD_FUSION`` = ``auto`` (default) turns on the two fusions whose result stays within the
    FIR/FFT tolerance of the staged reference chain (1e-5 relative, checked against the reference's
    staged output in ``tests/golden/chain*.npz`` and at full size): merging runs of FFT-mode FIRs and
    folding a fresh IIR cascade into the FIR run that follows it.  ``reference`` stages every step
    exactly as ``src/torchfx/wave.py:207-239`` does.  The per-feature variables
    ``TORCHFX_AMD_FUSE_{FIR,SPECTRAL,GAIN,EPILOGUE}`` (0/1) override either way.  ``auto`` also attaches a
    ``Gain`` / ``Normalize`` that follows a filter to that filter's kernel as an epilogue (bit-identical for
    the gain and the clamp; ``effect.Epilogued``); folding a gain into the coefficients stays opt-in."""
    auto = os.environ.get("TORCHFX_AMD_FUSION", "auto").lower() != "reference"

    def flag(name: str, dflt: bool) -> bool:
        v = os.environ.get(name)
        return dflt if v is None or v == "" else v == "1"
    return (flag("TORCHFX_AMD_FUSE_FIR", auto), flag("TORCHFX_AMD_FUSE_SPECTRAL", False), flag("TORCHFX_AMD_FUSE_GAIN", False),
            flag("TORCHFX_AMD_FUSE_EPILOGUE", auto))


def _recursive_default() -> bool:
    """``fuse_recursive`` of a new ``Wave``: on under ``TORCHFX_AMD_FUSION=auto`` (``TORCHFX_AMD_FUSE_RECURSIVE`` overrides)."""
    v = os.environ.get("TORCHFX_AMD_FUSE_RECURSIVE")
    if v is None or v == "":
        return os.environ.get("TORCHFX_AMD_FUSION", "auto").lower() != "reference"
    return v == "1"


# The spectral fold is taken only when this bound holds (relative to max(1, max|y|) of the replay)
FOLD_ERROR_LIMIT = 2e-6
_FOLD_ERR: "OrderedDict[tuple, float]" = OrderedDict()


def _fold_error_estimate(sos: np.ndarray, fir_flipped: np.ndarray, merged_flipped: np.ndarray) -> float:
    """What folding a cascade into the FIR run costs in accuracy, measured on the host once per (cascade, FIR) pair:
    2^16 pseudo-random samples through (a) the reference's staging -- float64 recursion, rounded to float32, float64
    convolution with the float32 taps -- and (b) the folded form as the device runs it -- the merged taps rounded to
    float32, one float32 FFT convolution (numpy's single-precision FFT; block length >= taps + 2^16, the device's
    2^16 ... 2^20-point blocks behave alike: the error grows with log N).  Returns 2 x max|a - b| / max(1, max|a|)."""
    import scipy.signal as sg

    n = 1 << 16
    x = np.random.default_rng(20240605).uniform(-1.0, 1.0, n).astype(np.float32)
    y_iir = sg.sosfilt(sos, x.astype(np.float64)).astype(np.float32)
    h_fir = fir_flipped[::-1].astype(np.float32).astype(np.float64)
    ref = sg.fftconvolve(y_iir.astype(np.float64), h_fir)[:n]
    h = merged_flipped[::-1].astype(np.float32)
    nfft = 1 << int(np.ceil(np.log2(n + h.size)))
    xf = np.zeros(nfft, np.float32); xf[:n] = x
    hf = np.zeros(nfft, np.float32); hf[:h.size] = h
    got = np.fft.irfft(np.fft.rfft(xf) * np.fft.rfft(hf), nfft)[:n]
    assert got.dtype == np.float32, "numpy >= 2.0: single-precision FFT"
    return 2.0 * float(np.abs(got.astype(np.float64) - ref).max()) / max(1.0, float(np.abs(ref).max()))


def _planner_fir(taps_flipped64: np.ndarray) -> nn.Module:
    """A stateless FFT-mode FIR around float64 FLIPPED taps (rounded once, when the kernel casts to x.dtype)."""
    from torchfx_amd.filter.fir import FIR

    fir = FIR.__new__(FIR)
    nn.Module.__init__(fir)
    fir._conv_mode, fir.a = "fft", [1.0]
    fir.register_buffer("kernel", torch.from_numpy(np.ascontiguousarray(taps_flipped64)).reshape(1, 1, -1))
    return fir


def _convolve64(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Float64 linear convolution; long operands go through one real FFT (error ~1e-16 of |a|·|b|, far
    below the float32 rounding the taps get at launch) instead of an O(Ka·Kb) ``np.convolve``: merging a
    2 419-tap impulse response into 66 559 taps costs 6 ms instead of 110."""
    if min(a.size, b.size) < 64:
        return np.convolve(a, b)
    n = a.size + b.size - 1
    nfft = 1 << (n - 1).bit_length()
    return np.fft.irfft(np.fft.rfft(a, nfft) * np.fft.rfft(b, nfft), nfft)[:n]


# merged FIRs by the identity + version of the member kernels (strong references keep the ids unique)
_MERGED: "OrderedDict[tuple, tuple[list, nn.Module]]" = OrderedDict()
# impulse responses of fresh cascades by SOS content
_IIR_FIR: "OrderedDict[bytes, nn.Module | None]" = OrderedDict()
_CACHE_MAX = 32
_CACHE_LOCK = threading.RLock()          # waves may be materialised from several host threads


def _lru_get(cache: OrderedDict, key):
    with _CACHE_LOCK:
        hit = cache.get(key)
        if hit is not None:
            cache.move_to_end(key)
        return hit


def _lru_put(cache: OrderedDict, key, value) -> None:
    with _CACHE_LOCK:
        cache[key] = value
        while len(cache) > _CACHE_MAX:
            cache.popitem(last=False)


def _merge_fir_run(run: list) -> nn.Module:
    """[FIR, FIR, ...] -> one FIR whose taps are the float64 convolution of the members'.  Cached per set of
    member kernels (tensor identity + version counter), so a pipeline that is planned again -- another wave,
    another length -- does not convolve again."""
    kernels = [f.kernel for f in run]
    key = tuple((id(k), k._version) for k in kernels)
    hit = _lru_get(_MERGED, key)
    if hit is not None:
        return hit[1]
    taps = None
    for k in kernels:
        # numpy, not torch: a flip + cast of 65 536 values through torch's CPU kernels costs 10-30 ms a piece (intra-op thread pool)
        b = np.ascontiguousarray(k.detach().cpu().reshape(-1).numpy()[::-1], dtype=np.float64)
        taps = b if taps is None else _convolve64(taps, b)
    merged = _planner_fir(taps[::-1])
    _lru_put(_MERGED, key, (kernels, merged))
    return merged


def _iir_as_fir(sos_t: Tensor, max_taps: int = 1 << 17) -> nn.Module | None:
    """A *fresh* (stateless) SOS cascade as an equivalent FIR: its impulse response truncated where the
    kernel planner says the filter has forgotten its past to float64 round-off (`warmup`:
    max|A^W| < 2^-60).  None when the memory is too long.  Cached by coefficient content."""
    import scipy.signal as sg

    from torchfx_amd import torchfx_ext

    sos = np.ascontiguousarray(sos_t.detach().cpu().numpy(), dtype=np.float64)
    key = sos.tobytes()
    with _CACHE_LOCK:
        if key in _IIR_FIR:
            _IIR_FIR.move_to_end(key)
            return _IIR_FIR[key]
    w = torchfx_ext.sos_plan_info(sos)["warmup"]
    fir = None
    if 0 <= w <= max_taps:
        imp = np.zeros(int(w) + 1)
        imp[0] = 1.0
        h = sg.sosfilt(sos, imp)                       # float64 impulse response, |tail| < 1e-18
        fir = _planner_fir(h[::-1])
    _lru_put(_IIR_FIR, key, fir)
    return fir


# ---- plan cache -----------------------------------------------------------------------------------
# The reference's materialisation is a cheap ``torch.cat`` of SOS rows per ``.ys`` (wave.py:207-239).  Ours
# derives merged taps and impulse responses, so the planned module list is cached per *pipeline*: the key is
# the identity of every member plus what the planner reads from it (coefficient tensor identity + version
# counter, fs, conv mode, gain settings, strategy type), the four fusion flags and the row length (the
# spectral fold asks the overlap-save geometry whether it pays).  Redesigning a filter
# (``compute_coefficients`` / ``set_parameter`` assign a new tensor), editing coefficients in place (version
# counter) or changing a Gain therefore miss; ``reset_state`` never matters because cached plans hold no
# state: planner-built cascades are re-instantiated around their cached table on every hit.  Entries keep
# strong references to the members and their coefficient tensors, so an ``id`` cannot be recycled while its
# entry lives; the cache is a 32-entry LRU.
_PLANS: "OrderedDict[tuple, tuple[list, list]]" = OrderedDict()


def plan_cache_clear() -> None:
    with _CACHE_LOCK:
        _PLANS.clear()
        _MERGED.clear()
        _IIR_FIR.clear()
        _FOLD_ERR.clear()


def _member_key(m: nn.Module, guard: list) -> tuple:
    from torchfx_amd.effect import Gain, Normalize

    guard.append(m)
    k: tuple = (id(m),)
    sos = getattr(m, "_sos", None)
    if isinstance(sos, Tensor):
        guard.append(sos)
        k += (id(sos), sos._version, getattr(m, "fs", None))
    ker = getattr(m, "kernel", None)
    if isinstance(ker, Tensor):
        guard.append(ker)
        k += (id(ker), ker._version, getattr(m, "_conv_mode", None))
    if isinstance(m, Gain):
        k += (m.gain, m.gain_type, m.clamp)
    elif isinstance(m, Normalize):
        guard.append(m.strategy)
        k += (id(m.strategy),)
    return k


def _lacks_coefficients(m: nn.Module) -> bool:
    return (hasattr(m, "_sos") and not isinstance(getattr(m, "_sos"), Tensor)) or \
           (hasattr(m, "kernel") and not isinstance(getattr(m, "kernel"), Tensor))


def _instantiate(cached: list) -> list:
    """A runnable plan from a cached one: planner-built cascades carry state while they run, so every
    materialisation gets its own (the reference builds a fresh ``FusedSOSCascade`` per ``.ys`` as well)."""
    from torchfx_amd.effect import Epilogued
    from torchfx_amd.filter.fused import FusedSOSCascade

    def fresh(m):
        if getattr(m, "_planner_built", False) and isinstance(m, FusedSOSCascade):
            c = FusedSOSCascade.from_table(m._stream.table)
            c._planner_built = True
            if hasattr(m, "fold_refused"):
                c.fold_refused = m.fold_refused
            if hasattr(m, "recursive_refused"):
                c.recursive_refused = m.recursive_refused
            return c
        if isinstance(m, Epilogued) and getattr(m.producer, "_planner_built", False):
            return Epilogued(fresh(m.producer), m.gain, m.norm)
        return m
    return [fresh(m) for m in cached]


def _plain_fir(m) -> bool:
    """FIR steps the planner may merge, fold into or attach an epilogue to: the stock stateless classes only.  A
    subclass with its own ``forward`` (``realtime.StatefulFIR`` carries history and takes no ``epilogue``) is
    staged as it is."""
    from torchfx_amd.filter.fir import FIR, DesignableFIR

    return isinstance(m, FIR) and type(m).forward in (FIR.forward, DesignableFIR.forward)


class Wave:
    """Discrete-time signal ``ys [C,T]`` sampled at ``fs`` with a deferred filter pipeline."""

    fs: int

    def __init__(self, ys, fs: int, device="cpu", metadata: dict[str, tp.Any] | None = None) -> None:
        self.fs = fs
        self._pipeline: list[nn.Module] = []
        self._ys = ys if isinstance(ys, Tensor) else Tensor(ys)   # Tensor(ys): float32, as wave.py:137
        self.metadata = metadata or {}
        self.fuse_fir, self.fuse_spectral, self.fuse_gain, self.fuse_epilogue = _fusion_defaults()
        self.fuse_recursive = _recursive_default()
        self.to(device)

    # ------------------------------------------------------------------ lazy data
    @property
    def ys(self) -> Tensor:
        self._materialize()
        return self._ys

    @ys.setter
    def ys(self, value: Tensor) -> None:
        self._ys = value
        self._pipeline = []

    def plan(self) -> list[nn.Module]:
        """The fused execution plan of the pending pipeline (``wave.py:216-233``), from the plan cache when
        this pipeline -- same members, same coefficients, same flags, same row length -- was planned before."""
        flags = (self.fuse_fir, getattr(self, "fuse_spectral", False), getattr(self, "fuse_gain", False),
                 getattr(self, "fuse_epilogue", False), getattr(self, "fuse_recursive", False))
        length = int(self._ys.shape[-1]) if self._ys.dim() else 0
        dtype = self._ys.dtype                       # the overlap-save path (and so the fold decision) depends on it
        guard: list = []
        key = (tuple(_member_key(m, guard) for m in self._pipeline), flags, length, dtype)
        # a member without coefficients yet (IIR.reset_state drops `_sos`; a FIR without a kernel) is keyed by identity alone:
        # such a pipeline is planned afresh every time instead of being looked up (advisor, round 3)
        cacheable = all(not _lacks_coefficients(m) for m in self._pipeline)
        hit = _lru_get(_PLANS, key) if cacheable else None
        if hit is not None:
            return _instantiate(hit[1])
        if self._ys.is_cuda and any(_plain_fir(m) for m in self._pipeline):
            from torchfx_amd import torchfx_ext
            torchfx_ext.prewarm(self._ys.device)     # the device-side one-time set-up runs while the taps are merged here
        built = self._build_plan(length, dtype)
        if cacheable:
            _lru_put(_PLANS, key, (guard, built))
        return _instantiate(built)

    def _build_plan(self, length: int, dtype: torch.dtype = torch.float32) -> list[nn.Module]:
        from torchfx_amd.filter.biquad import Biquad
        from torchfx_amd.filter.fir import FIR
        from torchfx_amd.filter._sos import CascadeTable
        from torchfx_amd.filter.fused import FusedSOSCascade
        from torchfx_amd.filter.iir import IIR

        from torchfx_amd.effect import Gain

        plan: list[nn.Module] = []
        items: list = []          # the open run in pipeline order: filters of one kind (+ folded Gains)
        lead: list = []           # folded Gains seen while no run is open: they go into the next run
        kind = None
        fold = getattr(self, "fuse_gain", False)

        def scaled_fir(m, g):
            if g == 1.0:
                return m
            f = FIR.__new__(FIR)
            nn.Module.__init__(f)
            f._conv_mode, f.a = m._conv_mode, [1.0]
            f.register_buffer("kernel", m.kernel.detach().to(torch.float64) * g)   # rounded once, at launch
            return f

        def flush() -> None:
            nonlocal items, kind
            if kind is None:
                return
            filters = [m for m in items if not isinstance(m, Gain)]
            if kind == "iir" and len(filters) == 1:
                plan.extend(items)                   # a lone IIR is stateful across waves: keep it staged
            else:
                # every folded gain scales the filter that follows it (trailing ones: the last filter)
                scales, nxt = [1.0] * len(filters), 0
                for m in items:
                    if isinstance(m, Gain):
                        g = m.linear_gain()
                        scales[min(nxt, len(filters) - 1)] *= 1.0 if g is None else g
                    else:
                        nxt += 1
                if kind == "iir":
                    cascade = FusedSOSCascade.from_table(CascadeTable.gather(filters, scales))
                    cascade._planner_built = True     # fresh per materialisation; the only cascades folded spectrally
                    plan.append(cascade)
                else:
                    mem = [scaled_fir(m, g) for m, g in zip(filters, scales)]
                    plan.append(_merge_fir_run(mem) if len(mem) >= 2 else mem[0])
            items, kind = [], None

        for m in self._pipeline:
            if fold and isinstance(m, Gain) and not m.clamp:
                (items if kind is not None else lead).append(m)
                continue
            k = "iir" if isinstance(m, (IIR, Biquad)) else (
                "fir" if (_plain_fir(m) and m._conv_mode != "direct" and (self.fuse_fir or fold)) else None)
            if k is None or (kind is not None and k != kind) or (k == "fir" and kind == "fir" and not self.fuse_fir):
                flush()
            if k is None:
                plan.extend(lead)
                lead = []
                plan.append(m)
            else:
                if kind is None:
                    items, lead = lead + [m], []
                else:
                    items.append(m)
                kind = k
        flush()
        plan.extend(lead)
        if getattr(self, "fuse_spectral", False):
            plan = self._spectral_plan(plan, length, dtype)
        if getattr(self, "fuse_recursive", False):
            plan = self._recursive_plan(plan, length, dtype)
        if getattr(self, "fuse_epilogue", False):
            plan = self._epilogue_plan(plan)
        return plan

    @staticmethod
    def _epilogue_plan(plan: list[nn.Module]) -> list[nn.Module]:
        """``fuse_epilogue``: ``filter | Gain`` , ``filter | Normalize`` and ``filter | Gain | Normalize`` run as
        the filter's kernel with an epilogue (``effect.Epilogued``) when the filter is an SOS module / cascade or
        an FFT-mode FIR and the normalisation strategy is one with a streaming reduction (peak, RMS, per
        channel)."""
        from torchfx_amd.effect import Epilogued, Gain, Normalize
        from torchfx_amd.filter.biquad import Biquad
        from torchfx_amd.filter.fir import FIR
        from torchfx_amd.filter.fused import CascadeFIR, FusedSOSCascade
        from torchfx_amd.filter.iir import IIR

        out: list[nn.Module] = []
        i = 0
        while i < len(plan):
            m = plan[i]
            producer = isinstance(m, (IIR, Biquad, FusedSOSCascade, CascadeFIR)) or (_plain_fir(m) and m._conv_mode != "direct")
            gain = norm = None
            j = i + 1
            if producer and j < len(plan) and isinstance(plan[j], Gain):
                gain, j = plan[j], j + 1
            if producer and j < len(plan) and isinstance(plan[j], Normalize) and Epilogued.norm_kind(plan[j]) is not None:
                norm, j = plan[j], j + 1
            if gain is not None or norm is not None:
                out.append(Epilogued(m, gain, norm))
                i = j
            else:
                out.append(m)
                i += 1
        return out

    @staticmethod
    def _ols_bytes_per_sample(taps: int, length: int, dtype: torch.dtype = torch.float32) -> float:
        """HBM bytes per output sample of one overlap-save pass with `taps` taps on rows of `length` samples of
        `dtype`, for the path that would run (``torchfx_ext.ols_plan_info``): the one-launch LDS kernel
        e N / S + e (e = element size), the three-pass pipeline 20 N / S + 4, rocFFT ~95 (float64: 190)."""
        from torchfx_amd import torchfx_ext

        return float(torchfx_ext.ols_plan_info(taps, length, (taps - 1, 0), dtype)["bytes_per_sample"])

    @staticmethod
    def _spectral_plan(plan: list[nn.Module], length: int = 0, dtype: torch.dtype = torch.float32) -> list[nn.Module]:
        """``fuse_spectral``: an LTI run  IIR-cascade | FIR...  is ONE linear system, so a freshly
        created (stateless) cascade that is followed by an FFT-mode FIR is folded into it as its
        impulse response, truncated where the cascade has forgotten its past to float64 round-off --
        the whole run becomes a single overlap-save pass (8 B/sample less HBM traffic, the recursive
        kernel is not launched).  The IIR part then runs in float32 FFT arithmetic like the FIR it
        joins (error ~1e-6 of the output scale instead of 1 ulp).  Only cascades the planner built itself in
        this plan are folded: a lone IIR step and a user-held ``FusedSOSCascade`` carry state from wave to
        wave (``fused.py:120-131``) and stay staged, whether they have run yet or not.  Not applied when the
        FIR is in direct mode or has its own ``forward`` (``StatefulFIR``), or when the longer taps would cost the overlap-save pass more HBM bytes per sample (block
        efficiency) than the 8 B/sample the recursive pass takes."""
        from torchfx_amd.filter.fir import FIR
        from torchfx_amd.filter.fused import FusedSOSCascade

        out: list[nn.Module] = []
        i = 0
        while i < len(plan):
            m = plan[i]
            nxt = plan[i + 1] if i + 1 < len(plan) else None
            if (isinstance(m, FusedSOSCascade) and getattr(m, "_planner_built", False) and m._state_x is None
                    and _plain_fir(nxt) and nxt._conv_mode != "direct"):
                eq = _iir_as_fir(m._sos)
                if eq is not None:
                    k0, k1 = int(nxt.kernel.numel()), int(nxt.kernel.numel()) + int(eq.kernel.numel()) - 1
                    try:
                        esz = 8.0 if dtype == torch.float64 else 4.0        # the recursive pass reads and writes the signal once
                        pays = (Wave._ols_bytes_per_sample(k1, length, dtype)
                                <= Wave._ols_bytes_per_sample(k0, length, dtype) + 2.0 * esz)
                    except RuntimeError:          # signal shorter than the taps: nothing to gain
                        pays = False
                    if pays:
                        merged = _merge_fir_run([eq, nxt])
                        # ... and only when the float32 FFT arithmetic the IIR part then runs in is measured to be harmless
                        ek = (m._sos.numpy().tobytes(), id(nxt.kernel), nxt.kernel._version)
                        err = _lru_get(_FOLD_ERR, ek)
                        if err is None:
                            err = _fold_error_estimate(m._sos.numpy(), nxt.kernel.detach().cpu().reshape(-1).numpy().astype(np.float64),
                                                       merged.kernel.detach().cpu().reshape(-1).numpy())
                            _lru_put(_FOLD_ERR, ek, (err, nxt.kernel))
                        else:
                            err = err[0]
                        if err <= FOLD_ERROR_LIMIT:
                            merged.fold_error_estimate = err
                            out.append(merged)
                            i += 2
                            continue
                        m.fold_refused = err                  # staged (or run inside the column pass): the plan shows why
            out.append(m)
            i += 1
        return out

    @staticmethod
    def _recursive_plan(plan: list[nn.Module], length: int = 0, dtype: torch.dtype = torch.float32) -> list[nn.Module]:
        """``fuse_recursive``: a cascade the planner built itself (fresh, zero state, dropped after the
        materialisation -- ``wave.py:221-233``) followed by an FFT-mode FIR runs as ONE overlap-save pipeline
        with the float64 recursion inside the forward column pass (``filter.fused.CascadeFIR``): the reference's
        arithmetic, one pass over the signal less.  Only where the kernel serves the geometry
        (``torchfx_ext.sos_fft_conv_supported``: float32 rows of any length, <= 8 sections whose memory fades within a
        4096-sample row, taps long enough for the 2^20-point block); a lone IIR step, a user-held ``FusedSOSCascade``
        (stateful across waves) and direct-mode FIRs stay staged.  A cascade that stays staged carries the reason
        (``recursive_refused``), which :meth:`explain` prints."""
        from torchfx_amd import torchfx_ext
        from torchfx_amd.filter.fused import CascadeFIR, FusedSOSCascade

        if dtype != torch.float32 or length <= 0:
            return plan
        out: list[nn.Module] = []
        i = 0
        while i < len(plan):
            m = plan[i]
            nxt = plan[i + 1] if i + 1 < len(plan) else None
            if (isinstance(m, FusedSOSCascade) and getattr(m, "_planner_built", False) and m._state_x is None
                    and _plain_fir(nxt) and nxt._conv_mode != "direct"):
                k = int(nxt.kernel.numel())
                if torchfx_ext.sos_fft_conv_supported(length, m._sos, k, (k - 1, 0)):
                    out.append(CascadeFIR(m._stream.table, nxt))
                    i += 2
                    continue
                m.recursive_refused = Wave._recursive_refusal(m._sos, k, length)
            out.append(m)
            i += 1
        return out

    @staticmethod
    def _recursive_refusal(sos: Tensor, taps: int, length: int) -> str:
        """Why ``tfx_sos_fft_conv_supported`` said no (host-only queries, the same ones the library asks itself)."""
        from torchfx_amd import torchfx_ext

        if int(sos.shape[0]) > 8:
            return f"{int(sos.shape[0])} sections > 8"
        warm = torchfx_ext.sos_fft_conv_warmup(sos)
        if warm < 0 or warm > 4096:
            return f"the cascade's memory ({warm} samples to 2^-40) is longer than a 4096-sample row of the transform"
        try:
            info = torchfx_ext.ols_plan_info(taps, length, (taps - 1, 0), torch.float32)
            return f"{taps} taps on rows of {length} samples run the {info['path']} path, not the 2^20 / 2^21-point three-pass pipeline"
        except RuntimeError as e:
            return str(e)

    def explain(self) -> list[str]:
        """One line per step of :meth:`plan` for THIS tensor: the step, the route it will take (``CascadeFIR``: the fused
        recursion-in-pass-A pipeline or the staged pair of launches) and why."""
        from torchfx_amd.effect import Epilogued
        from torchfx_amd.filter.fused import CascadeFIR, FusedSOSCascade

        lines = []
        for m in self.plan():
            inner = m.producer if isinstance(m, Epilogued) else m
            line = type(m).__name__ + (f"[{type(inner).__name__}]" if inner is not m else "")
            if isinstance(inner, CascadeFIR):
                path, why, _ = inner.route(self._ys)
                line += f": {path} -- {why}"
            elif isinstance(inner, FusedSOSCascade):
                if getattr(inner, "recursive_refused", None):
                    line += f": staged -- {inner.recursive_refused}"
                elif getattr(inner, "fold_refused", None) is not None:
                    line += f": staged -- spectral fold refused (error estimate {inner.fold_refused:.1e})"
            lines.append(line)
        return lines

    def _materialize(self) -> None:
        if not self._pipeline:
            return
        data = self._ys
        for step in self.plan():
            data = step(data)
        self._ys = data
        self._pipeline = []

    @classmethod
    def _deferred(cls, ys: Tensor, fs: int, device, metadata, pipeline: list[nn.Module],
                  fuse_fir: bool = False, fuse_spectral: bool = False, fuse_gain: bool = False,
                  fuse_epilogue: bool = False, fuse_recursive: bool = False) -> "Wave":
        w = object.__new__(cls)
        w._ys, w.fs, w._device, w.metadata, w._pipeline, w.fuse_fir = ys, fs, device, metadata, pipeline, fuse_fir
        w.fuse_spectral = fuse_spectral
        w.fuse_gain = fuse_gain
        w.fuse_epilogue = fuse_epilogue
        w.fuse_recursive = fuse_recursive
        return w

    # ------------------------------------------------------------------ device
    @property
    def device(self):
        return self._device

    @device.setter
    def device(self, device) -> None:
        self.to(device)

    def to(self, device) -> "Wave":
        self._device = device
        self._materialize()
        self._ys = self._ys.to(device)
        return self

    # ------------------------------------------------------------------ pipeline
    def __or__(self, f: nn.Module) -> "Wave":
        if not isinstance(f, nn.Module):
            raise TypeError(f"Expected nn.Module, but got {type(f).__name__} instead.")
        for m in f.modules():                       # includes f itself and nested containers
            if isinstance(m, FX):
                if getattr(m, "fs", 0) is None:
                    m.fs = self.fs
                if isinstance(m, AbstractFilter) and not m._has_computed_coeff:
                    m.compute_coefficients()
        steps = list(f.children()) if isinstance(f, nn.Sequential) else [f]
        return Wave._deferred(self._ys, self.fs, self._device, self.metadata,
                              self._pipeline + steps, self.fuse_fir, getattr(self, "fuse_spectral", False),
                              getattr(self, "fuse_gain", False), getattr(self, "fuse_epilogue", False),
                              getattr(self, "fuse_recursive", False))

    # ------------------------------------------------------------------ files
    _SUBTYPE_BY_ENCODING = {"PCM_S": lambda b: f"PCM_{b}", "PCM_U": lambda b: "PCM_U8" if b == 8 else f"PCM_{b}",
                            "PCM_F": lambda b: "FLOAT" if b == 32 else "DOUBLE"}

    @classmethod
    def from_file(cls, path, frame_offset: int = 0, num_frames: int = -1, device="cpu",
                  pcm16_on_device: bool = False) -> "Wave":
        """Read an audio file through ``soundfile`` (``wave.py:406-462``): float32 ``[channels, frames]``,
        ``fs`` and the ``num_frames / num_channels / subtype / format`` metadata.  With a ROCm ``device``
        the transposition runs on the GPU behind a chunked pinned upload; ``pcm16_on_device`` additionally
        reads 16-bit PCM files as int16 and converts on the GPU (same values, half the PCIe bytes)."""
        import soundfile as _sf

        stop = None if num_frames == -1 else frame_offset + num_frames
        try:
            info = _sf.info(str(path))
            metadata = {"num_frames": info.frames, "num_channels": info.channels,
                        "subtype": info.subtype, "format": info.format}
        except Exception:
            metadata = {}
        on_gpu = torch.device(device).type == "cuda"
        as_i16 = on_gpu and pcm16_on_device and metadata.get("subtype") == "PCM_16"
        data_np, fs = _sf.read(str(path), start=frame_offset, stop=stop, dtype="int16" if as_i16 else "float32",
                               always_2d=True)
        if on_gpu:
            from torchfx_amd import io as _io

            w = object.__new__(cls)
            w._ys = _io.upload_interleaved(data_np, device)
            w.fs, w._device, w.metadata, w._pipeline = fs, device, metadata, []
            w.fuse_fir, w.fuse_spectral, w.fuse_gain, w.fuse_epilogue = _fusion_defaults()
            w.fuse_recursive = _recursive_default()
            return w
        return cls(torch.from_numpy(np.ascontiguousarray(data_np.T)), fs, metadata=metadata)

    def save(self, path, format: str | None = None, encoding: str | None = None,  # noqa: A002
             bits_per_sample: int | None = None) -> None:
        """Write through ``soundfile.write`` (``wave.py:481-576``): format from the extension (WAV when
        unknown), ``encoding`` / ``bits_per_sample`` mapped to a libsndfile subtype, parent directories
        created.  Device tensors are interleaved on the GPU and downloaded in chunks."""
        import pathlib

        import soundfile as _sf

        out = pathlib.Path(path)
        out.parent.mkdir(parents=True, exist_ok=True)
        if format is None:
            format = {".wav": "WAV", ".flac": "FLAC", ".ogg": "OGG"}.get(out.suffix.lower(), "WAV")  # noqa: A001
        subtype = None
        if encoding is not None and bits_per_sample is not None:
            rule = self._SUBTYPE_BY_ENCODING.get(encoding)
            subtype = rule(bits_per_sample) if rule else f"{encoding}{bits_per_sample}"
        elif bits_per_sample is not None:
            subtype = f"PCM_{bits_per_sample}"
        elif encoding == "PCM_F":
            subtype = "FLOAT"
        ys = self.ys
        if ys.is_cuda and ys.dim() == 2 and ys.dtype == torch.float32:
            from torchfx_amd import io as _io

            frames = _io.download_interleaved(ys)
        else:
            frames = ys.cpu().numpy().T
        _sf.write(str(out), frames, self.fs, format=format, subtype=subtype)

    def transform(self, func, *args, **kwargs) -> "Wave":
        """Apply ``func`` to the (materialised) samples and wrap the result (``wave.py:334-360``)."""
        self._materialize()
        return Wave(func(self._ys, *args, **kwargs), self.fs)

    @classmethod
    def merge(cls, waves: tp.Sequence["Wave"], split_channels: bool = False) -> "Wave":
        """Mix (sum, zero-padded to the longest) or stack (``split_channels``) several waves of one
        sample rate (``wave.py:760-830``)."""
        if not waves:
            raise ValueError("No waves to merge. Provide at least one wave.")
        fs = waves[0].fs
        for w in waves:
            if w.fs != fs:
                raise ValueError(f"Sampling frequency mismatch: {w.fs} != {fs}. "
                                 "All waves must have the same sampling frequency.")
        if split_channels:
            return Wave(torch.cat([w.ys for w in waves], dim=0), fs)
        longest = max(len(w) for w in waves)
        first = waves[0].ys
        mix = torch.zeros((first.shape[0], longest), dtype=first.dtype, device=first.device)
        for w in waves:
            mix[:, : len(w)] += w.ys
        return Wave(mix, fs)

    def __len__(self) -> int:
        return self.ys.shape[1]

    def channels(self) -> int:
        return self.ys.shape[0]

    def get_channel(self, index: int) -> "Wave":
        return Wave(self.ys[index], self.fs)

    def duration(self, unit: str) -> float:
        return len(self) / self.fs * (1000 if unit == "ms" else 1)
