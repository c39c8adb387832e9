"""acarsdec_amd -- MI355X-native (gfx950) implementation of acarsdec's per-channel DSP hot path:
the rtl.c down-converter and the msk.c MSK demodulator with its acars.c framing feedback.

The product is lib/libacarsdec_amd.so (hand-written HIP kernels behind the C ABI declared in
include/acarsdec_amd.h).  The Python in this package is a thin ctypes binding plus host-side
signal generators; it contains no compute path and no CPU fallback.
"""
from . import _capi  # noqa: F401

__all__ = ["_capi", "decoder", "synth"]
