"""MS_ResUNet of the reference (modules/ms_resunet.py:96-264): a RefineNet-style CNN, 1 -> 1 channel, same-size in/out.

BASELINE config 1 is "evaluate.py on PyTorch CPU (plumbing, no GPU)": this network is a conv/BatchNorm/MaxPool graph
with no custom kernel in scope, so it runs on stock torch operators -- on the CPU, or on the
GPU through torch's own ROCm operators.  What is kept verbatim is the drop-in surface: the factory ``MS_ResUNet()``
(alias ``MSResUNet``), ``RefineNet(block, layers)``, and the state_dict schema (360 keys, 24 918 369 parameters;
pinned by golden G11, tests/test_cfg1_plumbing.py).  The graph is described by small tables instead of the
reference's hand-unrolled forward:

    encoder   conv5x5(pad 1)+BN+ReLU -> 4 bottleneck stages (planes 32/64/128/256, strides 1/2/2/2)      :102-115
    decoder   per level k = 1..4 (deepest first): dimred conv3x3 -> RCU(2x2) [-> joint conv, + upsampled deeper
              level, ReLU] -> chained residual pooling (4 stages of MaxPool5 + conv3x3) -> RCU(3x2)
              [-> joint conv -> ConvTranspose(4, 2, 1) -> centre crop to the next skip]                    :117-147, :210-256
    head      conv5x5(pad 2) -> conv3x3(pad 2)   (pad 2 on a 3x3 is in the reference: together with the pad-1 5x5 stem it
              restores the input size)                                                                   :148-149, :258-260
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


def conv3x3(in_planes, out_planes, stride=1, bias=False):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=bias)


class RCPB(nn.Module):
    """Chained residual pooling (ms_resunet.py:12-31): x += conv_i(maxpool5(top)), top chained through the stages."""

    def __init__(self, in_planes, out_planes, n_stages):
        super().__init__()
        for i in range(n_stages):
            self.add_module(f"{i + 1}_outvar_dimred", conv3x3(in_planes if i == 0 else out_planes, out_planes))
        self.stride, self.n_stages = 1, n_stages
        self.maxpool = nn.MaxPool2d(kernel_size=5, stride=1, padding=2)

    def forward(self, x):
        top = x
        for i in range(self.n_stages):
            top = getattr(self, f"{i + 1}_outvar_dimred")(self.maxpool(top))
            x = top + x
        return x


class RCUBlock(nn.Module):
    """Residual conv units (ms_resunet.py:35-55): n_blocks x [ (ReLU, conv3x3) x n_stages, + residual ]; only the first
    conv of a unit has a bias."""
    _suffix = ("_conv", "_conv_relu_varout_dimred")

    def __init__(self, in_planes, out_planes, n_blocks, n_stages):
        super().__init__()
        if n_stages > len(self._suffix):
            raise ValueError("RCUBlock supports at most 2 stages per unit")
        for i in range(n_blocks):
            for j in range(n_stages):
                self.add_module(f"{i + 1}{self._suffix[j]}",
                                conv3x3(in_planes if i == 0 and j == 0 else out_planes, out_planes, bias=(j == 0)))
        self.stride, self.n_blocks, self.n_stages = 1, n_blocks, n_stages

    def forward(self, x):
        for i in range(self.n_blocks):
            y = x
            for j in range(self.n_stages):
                y = getattr(self, f"{i + 1}{self._suffix[j]}")(F.relu(y))
            x = y + x
        return x


class Bottleneck(nn.Module):
    """1x1 -> 3x3(stride) -> 1x1(x4) with BatchNorm, projection shortcut when shapes change (ms_resunet.py:57-93)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        skip = x if self.downsample is None else self.downsample(x)
        return self.relu(y + skip)


# decoder levels, deepest first: (level k, encoder channels, width of this level, width handed to the next level or None)
_DECODER = ((1, 1024, 256, 128), (2, 512, 128, 128), (3, 256, 128, 128), (4, 128, 128, None))


class RefineNet(nn.Module):
    def __init__(self, block, layers):
        super().__init__()
        self.inplanes = 32
        self.conv1 = nn.Conv2d(1, 32, kernel_size=5, stride=1, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(32)
        self.relu = nn.ReLU(inplace=True)
        for k in (4, 3, 2):
            self.add_module(f"upCT{k}", nn.ConvTranspose2d(128, 128, kernel_size=4, stride=2, padding=1))
        for i, (planes, stride) in enumerate(((32, 1), (64, 2), (128, 2), (256, 2))):
            self.add_module(f"layer{i + 1}", self._make_layer(block, planes, layers[i], stride))
        for k, enc_c, width, out_w in _DECODER:
            self.add_module(f"p_ims1d2_outl{k}_dimred", conv3x3(enc_c, width))
            self.add_module(f"adapt_stage{k}_b", self._make_rcu(width, width, 2, 2))
            if k > 1:
                self.add_module(f"adapt_stage{k}_b2_joint_varout_dimred", conv3x3(width, width))
            self.add_module(f"mflow_conv_g{k}_pool", self._make_crp(width, width, 4))
            self.add_module(f"mflow_conv_g{k}_b", self._make_rcu(width, width, 3, 2))
            if out_w is not None:
                self.add_module(f"mflow_conv_g{k}_b3_joint_varout_dimred", conv3x3(width, out_w))
        self.clf_conv1 = nn.Conv2d(128, 64, kernel_size=5, stride=1, padding=2, bias=True)
        self.clf_conv2 = nn.Conv2d(64, 1, kernel_size=3, stride=1, padding=2, bias=True)

    @staticmethod
    def _crop_like(x, ref):
        """Centre-crop x to ref's spatial size (ConvTranspose may overshoot an odd-sized skip; ms_resunet.py:151-170)."""
        h, w = x.shape[-2:]
        hr, wr = ref.shape[-2:]
        if (h, w) == (hr, wr):
            return x
        dh, dw = h - hr, w - wr
        return x[:, :, dh // 2:h - (dh - dh // 2), dw // 2:w - (dw - dw // 2)]

    def _make_crp(self, in_planes, out_planes, stages):
        return nn.Sequential(RCPB(in_planes, out_planes, stages))

    def _make_rcu(self, in_planes, out_planes, blocks, stages):
        return nn.Sequential(RCUBlock(in_planes, out_planes, blocks, stages))

    def _make_layer(self, block, planes, blocks, stride=1):
        out_c = planes * block.expansion
        down = None
        if stride != 1 or self.inplanes != out_c:
            down = nn.Sequential(nn.Conv2d(self.inplanes, out_c, kernel_size=1, stride=stride, bias=False), nn.BatchNorm2d(out_c))
        seq = [block(self.inplanes, planes, stride, down)]
        self.inplanes = out_c
        seq += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*seq)

    def forward(self, x):
        x = self.relu(self.bn1(self.conv1(x)))
        skips = []
        for i in range(1, 5):
            x = getattr(self, f"layer{i}")(x)
            skips.append(x)
        carry = None                                   # upsampled output of the deeper level
        for k, _, _, out_w in _DECODER:
            skip = skips[4 - k]
            y = getattr(self, f"adapt_stage{k}_b")(getattr(self, f"p_ims1d2_outl{k}_dimred")(skip))
            if k == 1:
                y = self.relu(y)
            else:
                y = F.relu(getattr(self, f"adapt_stage{k}_b2_joint_varout_dimred")(y) + carry)
            y = getattr(self, f"mflow_conv_g{k}_b")(getattr(self, f"mflow_conv_g{k}_pool")(y))
            if out_w is not None:
                y = getattr(self, f"mflow_conv_g{k}_b3_joint_varout_dimred")(y)
                carry = self._crop_like(getattr(self, f"upCT{5 - k}")(y), skips[3 - k])
        return self.clf_conv2(self.clf_conv1(y))


def MS_ResUNet():
    """ms_resunet.py:262-264."""
    return RefineNet(Bottleneck, [3, 4, 3, 3])


MSResUNet = MS_ResUNet        # the spelling BASELINE.json uses
