"""``torchfx_amd.filter`` -- the filter classes of ``torchfx.filter`` (reference
``src/torchfx/filter/__init__.py:38-72``) on the HIP backend.  ``LogFilterBank`` (a SURVEY.md 8f "next" row) runs as one filter-bank launch."""
from torchfx_amd.filter._base import AbstractFilter, ParallelFilterCombination
from torchfx_amd.filter.biquad import (
    Biquad, BiquadAllPass, BiquadBPF, BiquadBPFPeak, BiquadHPF, BiquadLPF, BiquadNotch,
)
from torchfx_amd.filter.filterbank import LogFilterBank
from torchfx_amd.filter.fir import FIR, DesignableFIR
from torchfx_amd.filter.fused import FusedSOSCascade
from torchfx_amd.filter.iir import (
    IIR, AllPass, Butterworth, Chebyshev1, Chebyshev2, Elliptic, HiButterworth, HiChebyshev1,
    HiChebyshev2, HiElliptic, HiLinkwitzRiley, HiShelving, LinkwitzRiley, LoButterworth,
    LoChebyshev1, LoChebyshev2, LoElliptic, LoLinkwitzRiley, LoShelving, Notch, ParametricEQ,
    Peaking, Shelving,
)

__all__ = [
    "AbstractFilter", "ParallelFilterCombination",
    "AllPass", "Biquad", "BiquadAllPass", "BiquadBPF", "BiquadBPFPeak", "BiquadHPF", "BiquadLPF",
    "BiquadNotch", "Butterworth", "Chebyshev1", "Chebyshev2", "DesignableFIR", "Elliptic", "FIR",
    "FusedSOSCascade", "HiButterworth", "HiChebyshev1", "HiChebyshev2", "HiElliptic",
    "HiLinkwitzRiley", "HiShelving", "IIR", "LinkwitzRiley", "LoButterworth", "LoChebyshev1",
    "LoChebyshev2", "LoElliptic", "LoLinkwitzRiley", "LoShelving", "LogFilterBank", "Notch", "ParametricEQ",
    "Peaking", "Shelving",
]
