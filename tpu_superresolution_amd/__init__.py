"""MI355X-native SwinIR path (drop-in for tpu_superresolution's modules/network_swinir.py).

The arithmetic lives in libsrk.so (hand-written HIP for gfx950, C ABI in include/srk.h); this package
is the host side: ctypes binding, the nn.Module mirror of the reference's constructor/state_dict,
the fused optimizer and the data-parallel wrapper.
"""
from .network_swinir import SwinIR, window_partition, window_reverse  # noqa: F401

__all__ = ["SwinIR", "window_partition", "window_reverse"]
