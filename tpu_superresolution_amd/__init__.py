"""MI355X-native SwinIR path (drop-in for tpu_superresolution's modules/network_swinir.py).

The arithmetic lives in libsrk.so (hand-written HIP for gfx950, C ABI in include/srk.h); this package
is the host side: ctypes binding, the nn.Module mirror of the reference's constructor/state_dict,
the fused optimizer and the data-parallel wrapper -- plus the reference's other entry points
(train.py / evaluate.py / finetune_swinir.py) and the MS_ResUNet plumbing model (stock torch operators).
"""
from .dat_arch import DAT  # noqa: F401
from .hat_arch import HAT  # noqa: F401
from .ms_resunet import MS_ResUNet, MSResUNet  # noqa: F401
from .network_swinir import SwinIR, window_partition, window_reverse  # noqa: F401

__all__ = ["SwinIR", "window_partition", "window_reverse", "MS_ResUNet", "MSResUNet", "HAT", "DAT"]
