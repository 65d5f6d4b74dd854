"""isegmi -- MI355X-native RoI/mask inference hot path (Mask R-CNN / Yolact), Python host side.

Thin ctypes host over libisegmi.so (hand-written gfx950 HIP kernels).  No torch, no triton.
"""
from . import _ffi  # noqa: F401

__all__ = ["_ffi"]
