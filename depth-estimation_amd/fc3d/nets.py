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
    def __init__(self, out_channels: int, lo, hi):
        super().__init__()
        self.enc = nn.ModuleList()
        cin = 1
        for w in WIDTHS:
            self.enc.append(nn.Sequential(ConvBN(cin, w), ConvBN(w, w)))
            cin = w
        self.up = nn.ModuleList()
        self.dec = nn.ModuleList()
        for w in reversed(WIDTHS[:-1]):
            self.up.append(Up(2 * w, w))
            self.dec.append(nn.Sequential(ConvBN(2 * w, w), ConvBN(w, w)))
        self.head = ConvBN(WIDTHS[0], out_channels, k=1, relu=False)
        self.register_buffer("lo", torch.as_tensor(lo, dtype=torch.float32).reshape(1, -1, 1, 1))
        self.register_buffer("hi", torch.as_tensor(hi, dtype=torch.float32).reshape(1, -1, 1, 1))

    def forward(self, x):
        x = x.permute(0, 3, 1, 2)
        skips = []
        for i, blk in enumerate(self.enc):
            if i:
                x = F.max_pool2d(x, 2)
            x = blk(x)
            skips.append(x)
        for up, dec, skip in zip(self.up, self.dec, reversed(skips[:-1])):
            x = dec(torch.cat([up(x), skip], dim=1))
        out = torch.sigmoid(self.head(x)) * (self.hi - self.lo) + self.lo
        return out.permute(0, 2, 3, 1)


def UNet_xy() -> UNet:
    """x/z, y/z regressor; network_pytorch.py:56-132."""
    return UNet(2, XY_MIN, XY_MAX)


def UNet_z() -> UNet:
    """depth regressor; network_pytorch.py:135-211."""
    return UNet(1, [Z_MIN], [Z_MAX])
