"""vadc_amd -- MI355X-native Silero VAD v3.1 forward pass behind vadc's backend boundary.

The product is libvadc_amd.so (vadc_amd/csrc, C-ABI in include/vadc_amd.h).  This package holds the ctypes
binding, the host-side mirror of the reference's backend interface, the .testtensor container reader and
the synthetic-input generator used by the benchmark.
"""
from . import synth, testtensor  # noqa: F401

__all__ = ["synth", "testtensor"]
