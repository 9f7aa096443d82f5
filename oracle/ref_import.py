"""Import the reference's model files READ-ONLY from /root/reference (build container only).

TEST INFRASTRUCTURE.  Used by ``oracle/make_golden.py`` and by the CPU tests that pin the oracle
against the live reference when ``/root/reference`` is present.  Never used on the GPU box (the
reference does not travel) and never by the product package.

``timm`` is not installed here; the reference needs three symbols from it
(network_swinir.py:11, hat_arch.py:6, dat_arch.py:7).  They are provided by an in-memory stand-in
registered in ``sys.modules``.  The stand-in only affects weight-init RNG and
train-mode DropPath, neither of which is used for parity.
"""
from __future__ import annotations

import os
import sys
import types
import warnings

import torch
import torch.nn as nn

REFERENCE_DIR = "/root/reference/modules"


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_DIR, "network_swinir.py"))


def _to_2tuple(x):
    if isinstance(x, (tuple, list)):
        return tuple(x)
    return (x, x)


class _DropPath(nn.Module):
    def __init__(self, drop_prob: float = 0.0, scale_by_keep: bool = True):
        super().__init__()
        self.drop_prob = drop_prob
        self.scale_by_keep = scale_by_keep

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        m = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            m.div_(keep)
        return x * m


def install_timm_standin() -> None:
    if "timm" in sys.modules and not getattr(sys.modules["timm"], "_srk_standin", False):
        return  # a real timm is importable; use it
    for name in ("timm", "timm.layers", "timm.models", "timm.models.layers"):
        mod = types.ModuleType(name)
        mod._srk_standin = True
        sys.modules[name] = mod
    for name in ("timm.layers", "timm.models.layers"):
        mod = sys.modules[name]
        mod.to_2tuple = _to_2tuple
        mod.trunc_normal_ = torch.nn.init.trunc_normal_
        mod.DropPath = _DropPath


def import_reference(module: str = "network_swinir"):
    """Return the reference module object (e.g. ``network_swinir``)."""
    if not reference_available():
        raise RuntimeError("reference sources are not present at " + REFERENCE_DIR)
    sys.dont_write_bytecode = True
    install_timm_standin()
    if REFERENCE_DIR not in sys.path:
        sys.path.insert(0, REFERENCE_DIR)
    warnings.filterwarnings("ignore", message="torch.meshgrid")
    return __import__(module)
