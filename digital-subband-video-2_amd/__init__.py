"""dsv2_amd: MI355X-native DSV2 (Digital Subband Video 2, bitstream v2.8) hot path.

The product is the C-ABI shared library ``libdsv2hip.so`` (HIP kernels for gfx950 +
a C++ host controller, drop-in for the reference's dsv_encoder.h / dsv_decoder.h
API).  This Python package only holds the thin ctypes host binding used by the
tests and bench, and the deterministic synthetic-video generator.
"""
from . import sharding, synth  # noqa: F401
