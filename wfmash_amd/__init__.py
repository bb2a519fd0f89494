"""wfmash_amd -- MI355X-native wfmash hot path (align: BiWFA gap-affine-2p; map: mashmap3 sketching).

The product is `libwfmash_hip.so` (C ABI in include/wfmash_hip.h, HIP kernels in
wfmash_amd/csrc).  This package only holds the build recipe and ctypes plumbing.
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
