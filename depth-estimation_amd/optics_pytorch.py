"""Drop-in for the reference's optics_pytorch.py: same module globals, function
names, argument order and defaults, backed by the gfx950 kernels in fc3d.

Put this directory on sys.path ahead of the reference checkout and
`from optics_pytorch import codePattern` (train_pytorch.py:29) picks this up.

Differences a caller can observe:
  * tensors must be on a GPU (the reference's default device is 'cuda' too);
  * codePattern never materialises the [B,H,W,L] layer mask or the L-channel
    convolution: each pixel is blurred with the PSF of its own layer only;
  * the PSF stack is uploaded once per distinct `PSFs_np` array object, not on
    every call (rebind `optics_pytorch.PSFs_np` to switch stacks, or call
    invalidate_psf_cache() after editing the array in place);
  * an optional keyword `n_layers` lets a single-channel pattern be passed.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

from fc3d import ops as _ops

#### optical parameters (optics_pytorch.py:13-19)
wvl = 530e-9
z_min = 0.695714566
z_max = 0.941062289
z0 = 0.8
D = 21 / 2 * 1.4 * 1e-3

# experimentally calibrated PSFs if present in the working directory, else the
# reference's all-ones fallback stack (optics_pytorch.py:22-27)
fn_psfs = 'psfs_exp.npy'
if os.path.exists(fn_psfs):
    PSFs_np = np.load(fn_psfs)
else:
    print(f"Warning: {fn_psfs} not found. Using dummy PSFs.", file=sys.stderr)   # the reference prints to stdout
    PSFs_np = np.ones((21, 5, 5, 1), dtype=np.float32)

_psf_cache: dict = {}
_grid_cache: dict = {}


def invalidate_psf_cache() -> None:
    _psf_cache.clear()


def _device_psfs(device: torch.device) -> torch.Tensor:
    """[L,Kh,Kw] fp32 copy of the current PSFs_np on `device`, cached per array object."""
    arr = PSFs_np
    key = (id(arr), arr.__array_interface__['data'][0], arr.shape, str(arr.dtype), str(device))
    hit = _psf_cache.get(key)
    if hit is None:
        if arr.ndim != 4 or arr.shape[-1] != 1:
            raise RuntimeError(f"PSFs_np must be [N_layers, H_psf, W_psf, 1], got {arr.shape}")
        dev = torch.from_numpy(np.ascontiguousarray(arr[..., 0], dtype=np.float32)).to(device)
        if len(_psf_cache) > 16:
            _psf_cache.clear()
        hit = (arr, dev)          # holding `arr` keeps its id from being reused
        _psf_cache[key] = hit
    return hit[1]


def _layer_grid(n_layers: int):
    key = (n_layers, wvl, D, z0, z_min, z_max)
    g = _grid_cache.get(key)
    if g is None:
        g = _ops.layer_grid(n_layers, wvl, D, z0, z_min, z_max)
        _grid_cache[key] = g
    return g


def z2Wm(z):
    """Depth -> wavefront error; python floats, numpy arrays and tensors alike
    (optics_pytorch.py:58-68)."""
    k = 2 * np.pi / wvl
    return k * (D / 2) ** 2 / 2 * (1 / z - 1 / z0)


def gen_WmMask(WmMap, Wm_layers, device='cuda'):
    """One-hot layer mask [B,H,W,N_layers] (optics_pytorch.py:30-55).  codePattern
    does not use it (the layer is decided per pixel inside the kernel); it is kept
    for callers that want the mask itself and runs as stock tensor ops on WmMap's
    device."""
    n = len(Wm_layers)
    step = (Wm_layers[-1] - Wm_layers[0]) / (n - 1)
    centres = torch.as_tensor(np.asarray(Wm_layers / step, dtype=np.float32), device=WmMap.device)
    r = torch.round(WmMap / float(np.float32(step)))
    return (torch.abs(r - centres.view(1, 1, 1, -1)) < 0.01).float()


def _channel0(pattern: torch.Tensor) -> torch.Tensor:
    """[B,H,W,1] view of channel 0 that keeps autograd cheap.

    Slicing an L-channel tensor that requires grad makes autograd allocate an
    L-channel zero gradient; when the pattern is a stride-0 expansion of a
    single-channel tensor the gradient can go to that tensor directly."""
    if pattern.shape[-1] == 1:
        return pattern
    if pattern.requires_grad and pattern.stride(-1) == 0 and pattern._base is not None:
        base = pattern._base
        if base.dim() == 4 and base.shape[-1] == 1 and base.shape[1:3] == pattern.shape[1:3] \
                and base.shape[0] in (1, pattern.shape[0]) and base.data_ptr() == pattern.data_ptr():
            return base.expand(pattern.shape[0], -1, -1, 1)
    return pattern[:, :, :, 0:1]


def codePattern(pattern, zMap, device='cuda', is_conv_psf=True, n_layers=None):
    """Blur the projected pattern with the PSF of each pixel's depth layer and
    normalise every image to max 1 (optics_pytorch.py:71-121).

    pattern [B,H,W,N_layers] (only channel 0 is read), zMap [B,H,W,1]
    -> [B,H,W,1].  `device` is accepted for signature compatibility; the work
    runs where the tensors are."""
    if pattern.dim() != 4 or zMap.dim() != 4:
        raise RuntimeError(f"codePattern expects [B,H,W,C] tensors, got {tuple(pattern.shape)}, {tuple(zMap.shape)}")
    if pattern.shape[0] == 0:                     # empty batch: empty result, as the reference's tensor ops give
        return pattern[..., 0:1].clone()
    L = int(pattern.shape[-1]) if n_layers is None else int(n_layers)
    if n_layers is not None and pattern.shape[-1] not in (1, L):
        raise RuntimeError(f"pattern has {pattern.shape[-1]} channels but n_layers={L}")
    psf = _device_psfs(pattern.device)
    if is_conv_psf and psf.shape[0] != L:
        raise RuntimeError(f"The size of tensor a ({psf.shape[0]}) must match the size of tensor b ({L}) "
                           "at non-singleton dimension 3")
    out, _layer, _mx, _am = _ops.CodePatternFn.apply(_channel0(pattern), zMap, psf, _layer_grid(L), bool(is_conv_psf))
    return out


def codePattern_simple(pattern, zMap):
    """No blur: channel 0 of the pattern (optics_pytorch.py:124-137)."""
    if pattern.shape[-1] > 1:
        return pattern[:, :, :, 0:1]
    return pattern
