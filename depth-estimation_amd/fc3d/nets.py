"""The two reconstruction U-Nets of the reference (network_pytorch.py:56-211),
restated with a table-driven builder.  They run on stock PyTorch-ROCm (MIOpen);
this project accelerates the image formation around them, not the networks.

Architecture: 5 levels, 32..512 channels, every conv is 3x3 with reflect
padding + BatchNorm + ReLU, max-pool 2 down, nearest x2 + 3x3 conv up, skip
concatenation, 1x1 conv + BatchNorm + sigmoid head scaled to an output range.
Input [B,H,W,1] -> output [B,H,W,C] like the reference."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

Z_MIN, Z_MAX = 0.695714566, 0.941062289
XY_MIN = np.array([-768 / 2 / 1666.67, -512 / 2 / 1666.67], dtype=np.float32)
XY_MAX = np.array([768 / 2 / 1666.67, 512 / 2 / 1666.67], dtype=np.float32)
WIDTHS = (32, 64, 128, 256, 512)


class ConvBN(nn.Module):
    def __init__(self, cin, cout, k=3, relu=True):
        super().__init__()
        self.pad = (k - 1) // 2
        self.conv = nn.Conv2d(cin, cout, k, padding=0)
        self.bn = nn.BatchNorm2d(cout, momentum=0.1)
        self.relu = relu

    def forward(self, x):
        if self.pad:
            x = F.pad(x, (self.pad,) * 4, mode="reflect")
        x = self.bn(self.conv(x))
        return F.relu(x, inplace=True) if self.relu else x


class Up(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 3, padding=0)

    def forward(self, x):
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        return self.conv(F.pad(x, (1, 1, 1, 1), mode="reflect"))


class UNet(nn.Module):
    """Sub-modules carry the reference's attribute names (down<l>_<i>, up<l>_0 for the up-convolution,
    up<l>_1/2 for the decoder pair, up1_3 for the head; network_pytorch.py:64-101, 143-180), so a
    state_dict written by the reference loads here with strict=True and the other way round.  The
    output range is a pair of persistent buffers under the names given in `range_names`
    (xy_min / xy_max, network_pytorch.py:104-105) or, with range_names=None, kept out of the
    state_dict like the plain floats of network_pytorch.py:183-184."""

    def __init__(self, out_channels: int, lo, hi, range_names=None):
        super().__init__()
        self.levels = len(WIDTHS)
        cin = 1
        for lvl, w in enumerate(WIDTHS, 1):
            self.add_module(f"down{lvl}_1", ConvBN(cin, w))
            self.add_module(f"down{lvl}_2", ConvBN(w, w))
            cin = w
        for lvl in range(self.levels - 1, 0, -1):
            w = WIDTHS[lvl - 1]
            self.add_module(f"up{lvl}_0", Up(2 * w, w))
            self.add_module(f"up{lvl}_1", ConvBN(2 * w, w))
            self.add_module(f"up{lvl}_2", ConvBN(w, w))
        self.add_module("up1_3", ConvBN(WIDTHS[0], out_channels, k=1, relu=False))
        self._range = range_names or ("_lo", "_hi")
        for name, v in zip(self._range, (lo, hi)):
            self.register_buffer(name, torch.as_tensor(np.asarray(v), dtype=torch.float32).reshape(1, -1, 1, 1),
                                 persistent=range_names is not None)

    def forward(self, x):
        x = x.permute(0, 3, 1, 2)
        m = self._modules
        skips = []
        for lvl in range(1, self.levels + 1):
            if lvl > 1:
                x = F.max_pool2d(x, 2)
            x = m[f"down{lvl}_2"](m[f"down{lvl}_1"](x))
            skips.append(x)
        for lvl in range(self.levels - 1, 0, -1):
            x = torch.cat([m[f"up{lvl}_0"](x), skips[lvl - 1]], dim=1)
            x = m[f"up{lvl}_2"](m[f"up{lvl}_1"](x))
        lo, hi = (self._buffers[n] for n in self._range)
        out = torch.sigmoid(m["up1_3"](x)) * (hi - lo) + lo
        return out.permute(0, 2, 3, 1)


def UNet_xy() -> UNet:
    """x/z, y/z regressor; network_pytorch.py:56-132."""
    return UNet(2, XY_MIN, XY_MAX, range_names=("xy_min", "xy_max"))


def UNet_z() -> UNet:
    """depth regressor; network_pytorch.py:135-211."""
    return UNet(1, [Z_MIN], [Z_MAX])
