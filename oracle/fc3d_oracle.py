"""CPU oracle for the FreeCam3D image-formation path.

THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  It restates, with stock
torch-CPU / numpy operators, the arithmetic of the reference functions named in
each docstring (paths are relative to the reference checkout).  Only the test
suite, the smoke check and the CPU-baseline leg of the benchmark may import it;
the shipped operators never do.

Pinned by tests/golden/*.npz, which were produced by importing the reference's
own modules on CPU (tests/golden/make_golden.py) -- the reference has no test
suite or golden vectors of its own.

All functions take and return fp32 torch tensors on the CPU unless noted.
Layouts follow the reference: images/depth/grids are [B,H,W,C]; homogeneous
scene coordinates are [B,4,H,W]; poses are [B,4,4].
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------- #
# optics  (optics_pytorch.py:13-19)
# --------------------------------------------------------------------------- #
WVL = 530e-9
Z_MIN = 0.695714566
Z_MAX = 0.941062289
Z0 = 0.8
APERTURE_D = 21 / 2 * 1.4 * 1e-3


def defocus_coeff() -> float:
    """float64 constant multiplying (1/z - 1/z0); optics_pytorch.py:66-67."""
    k = 2 * np.pi / WVL
    return k * (APERTURE_D / 2) ** 2 / 2


def z2Wm(z):
    """Depth -> defocus wavefront error; optics_pytorch.py:58-68.

    Works for python floats, numpy arrays and torch tensors like the
    reference; for an fp32 tensor every step is an fp32 op (reciprocal,
    subtract 1.25, multiply by the fp32-cast coefficient)."""
    return defocus_coeff() * (1 / z - 1 / Z0)


def layer_grid(n_layers: int):
    """(centres/step as float64 array, step as float64); optics_pytorch.py:82-84, 40-44."""
    lo = z2Wm(Z_MAX)
    hi = z2Wm(Z_MIN)
    layers = np.linspace(lo, hi, n_layers)
    step = (layers[-1] - layers[0]) / (n_layers - 1)
    return layers / step, step


def gen_WmMask(WmMap: torch.Tensor, n_layers: int) -> torch.Tensor:
    """One-hot depth-layer mask [B,H,W,L]; optics_pytorch.py:30-55.

    The reference loops over layers; here the L comparisons are broadcast.
    Scalars meet the fp32 tensor as fp32, as torch does for python scalars."""
    centres, step = layer_grid(n_layers)
    centres32 = torch.tensor(centres.astype(np.float32))
    r = torch.round(WmMap / float(np.float32(step)))           # [B,H,W,1]
    return (torch.abs(r - centres32.view(1, 1, 1, -1)) < 0.01).float()


def layer_index(zMap: torch.Tensor, n_layers: int) -> torch.Tensor:
    """Integer layer per pixel, -1 where no layer matches (output is 0 there).

    Derived from gen_WmMask; the one-hot never has two ones because layer
    centres are one step apart."""
    m = gen_WmMask(z2Wm(zMap), n_layers)
    idx = m.argmax(dim=-1, keepdim=True)
    return torch.where(m.sum(dim=-1, keepdim=True) > 0, idx, torch.full_like(idx, -1))


def codePattern(pattern: torch.Tensor, zMap: torch.Tensor, psfs, is_conv_psf: bool = True,
                return_parts: bool = False):
    """Depth-layered PSF blur + per-image max normalisation; optics_pytorch.py:71-121.

    pattern [B,H,W,L] (only channel 0 is used), zMap [B,H,W,1],
    psfs [L,Kh,Kw,1] (numpy or tensor; the reference reads a module global).
    """
    L = pattern.shape[-1]
    mask = gen_WmMask(z2Wm(zMap), L)
    p0 = pattern[..., 0:1]
    if is_conv_psf:
        w = torch.as_tensor(np.asarray(psfs), dtype=torch.float32).permute(0, 3, 1, 2)  # [L,1,Kh,Kw]
        conv = F.conv2d(p0.permute(0, 3, 1, 2), w, padding='same').permute(0, 2, 3, 1)
    else:
        conv = p0
    raw = torch.sum(conv * mask, dim=-1, keepdim=True)
    col_max = torch.max(raw, dim=1, keepdim=True)[0]
    img_max = torch.max(col_max, dim=2, keepdim=True)[0]
    out = raw / (img_max + 1e-10)
    if return_parts:
        return out, raw, img_max
    return out


def argmax_first_w_then_h(raw: torch.Tensor) -> torch.Tensor:
    """Pixel that receives the max's gradient in the reference's two-stage
    torch.max (dim=1 then dim=2, optics_pytorch.py:118): smallest w among the
    maxima, then smallest h in that column.  Returns [B,2] = (h, w)."""
    B, H, W, _ = raw.shape
    r = raw[..., 0]
    col_max, col_arg = torch.max(r, dim=1)        # [B,W]
    w = torch.argmax(col_max, dim=1)              # first occurrence
    h = col_arg[torch.arange(B), w]
    return torch.stack([h, w], dim=1)


# --------------------------------------------------------------------------- #
# geometry
# --------------------------------------------------------------------------- #
FOCAL = 1666.67


def pixel_to_ray_array(width: int = 1200, height: int = 800) -> np.ndarray:
    """float64 [H,W,3] ray table; geometry_pytorch.py:15-41 without the
    per-pixel python loop (same expression evaluated on index grids)."""
    x = np.arange(width, dtype=np.float64)
    y = np.arange(height, dtype=np.float64)
    xv = width / (2 * FOCAL) * ((2.0 * ((x + 0.5) / width)) - 1.0)
    yv = height / (2 * FOCAL) * ((2.0 * ((y + 0.5) / height)) - 1.0)
    out = np.empty((height, width, 3), dtype=np.float64)
    out[..., 0] = xv[None, :]
    out[..., 1] = yv[:, None]
    out[..., 2] = 1.0
    return out


def points_in_camera_coords(z_map: torch.Tensor, rays, allBatch: bool = False) -> torch.Tensor:
    """[z*rx, z*ry, z*rz, 1] as [B,H,W,4]; geometry_pytorch.py:44-65."""
    if allBatch:
        r = rays.float()
    else:
        r = torch.from_numpy(np.asarray(rays)).float().unsqueeze(0)
    xyz = z_map * r                                   # broadcast == repeat-then-multiply
    ones = torch.ones_like(z_map)
    return torch.cat([xyz.expand(z_map.shape[0], -1, -1, -1), ones], dim=-1)


def z2pointcloud(z: torch.Tensor, width=None, height=None) -> torch.Tensor:
    """[B,4,H,W]; geometry_pytorch.py:225-235.  The reference always uses the
    1200x800 table; width/height let tests use other sizes."""
    H, W = z.shape[1], z.shape[2]
    rays = pixel_to_ray_array(W if width is None else width, H if height is None else height)
    return points_in_camera_coords(z, rays).permute(0, 3, 1, 2)


def get_intrinsic(B: int, cx: float = 1200 / 2, cy: float = 800 / 2) -> torch.Tensor:
    """geometry_pytorch.py:238-252 (principal point hard-coded to 600,400 there)."""
    K = np.diag([FOCAL, FOCAL, 1, 1]).astype(np.float64)
    K[0, 2] = cx
    K[1, 2] = cy
    return torch.from_numpy(K).float().unsqueeze(0).repeat(B, 1, 1)


def coord_transform(coord: torch.Tensor, transform: torch.Tensor) -> torch.Tensor:
    """Project [B,4,H,W] by [B,4,4] and divide by depth; geometry_pytorch.py:68-90."""
    B, _, H, W = coord.shape
    u = torch.matmul(transform, coord.reshape(B, 4, H * W))
    zden = u[:, 2:3] + 1e-10
    g = torch.cat([u[:, 0:1] / zden, u[:, 1:2] / zden], dim=1)
    return g.reshape(B, 2, H, W).permute(0, 2, 3, 1)


def coord0TOgrid1(coord0: torch.Tensor, pose01: torch.Tensor, cx: float = 600.0, cy: float = 400.0) -> torch.Tensor:
    """geometry_pytorch.py:93-106."""
    T = torch.matmul(get_intrinsic(coord0.shape[0], cx, cy), pose01)
    return coord_transform(coord0, T)


def bilinear_sampler(imgs: torch.Tensor, coords: torch.Tensor) -> torch.Tensor:
    """geometry_pytorch.py:109-177.  Weights come from the CLAMPED corners,
    flat gather indices are formed in fp32 exactly as the reference does
    (so they lose integer precision above 2**24 elements, like the reference)."""
    B, Hs, Ws, C = imgs.shape
    _, Ht, Wt, _ = coords.shape
    cx = coords[..., 0:1]
    cy = coords[..., 1:2]
    fx = torch.floor(cx)
    fy = torch.floor(cy)
    lim_x = float(Ws - 1)
    lim_y = float(Hs - 1)
    x0 = fx.clamp(0.0, lim_x)
    x1 = (fx + 1).clamp(0.0, lim_x)
    y0 = fy.clamp(0.0, lim_y)
    y1 = (fy + 1).clamp(0.0, lim_y)
    wx0 = x1 - cx
    wx1 = cx - x0
    wy0 = y1 - cy
    wy1 = cy - y0
    base = (torch.arange(B).float() * float(Ws * Hs)).view(B, 1, 1, 1).expand(B, Ht, Wt, 1)
    row0 = base + y0 * float(Ws)
    row1 = base + y1 * float(Ws)
    flat = imgs.reshape(-1, C)

    def tap(xc, row):
        return flat[(xc + row).reshape(-1).long()].reshape(B, Ht, Wt, C)

    i00 = tap(x0, row0)
    i01 = tap(x0, row1)
    i10 = tap(x1, row0)
    i11 = tap(x1, row1)
    return (wx0 * wy0) * i00 + (wx0 * wy1) * i01 + (wx1 * wy0) * i10 + (wx1 * wy1) * i11


def visible_indices(grid: torch.Tensor, method: str) -> torch.Tensor:
    """int64 flat indices [B,n] marked by gen_visible_mask; geometry_pytorch.py:195-205."""
    B, H, W, _ = grid.shape
    gx = grid[..., 0]
    gy = grid[..., 1]
    if method == 'all':
        fx, cx_ = torch.floor(gx), torch.ceil(gx)
        fy, cy_ = torch.floor(gy), torch.ceil(gy)
        idx = torch.stack([fx + fy * W, fx + cy_ * W, cx_ + fy * W, cx_ + cy_ * W], dim=3)
    elif method == 'round':
        idx = torch.round(gx) + torch.round(gy) * W
    else:
        raise UnboundLocalError("local variable 'grid_idx_1d' referenced before assignment")
    return idx.reshape(B, -1).long()


def gen_visible_mask(grid: torch.Tensor, method: str = 'round') -> torch.Tensor:
    """Forward-splat visibility mask [B,H,W,1]; geometry_pytorch.py:180-222.
    Only the flat index is range-checked, so out-of-row x wraps into
    neighbouring rows exactly as in the reference."""
    B, H, W, _ = grid.shape
    idx = visible_indices(grid, method)
    ok = (idx >= 0) & (idx < H * W)
    out = torch.zeros(B, H * W + 1, dtype=torch.bool)
    out.scatter_(1, torch.where(ok, idx, torch.full_like(idx, H * W)), True)
    return out[:, :H * W].float().reshape(B, H, W, 1)


def warp_p2c(Ip, coord_p, coord_c, pose_p2c, cx: float = 600.0, cy: float = 400.0, pose_c2p=None):
    """geometry_pytorch.py:255-281 -> (Ic*mask, grid_p2c, grid_c2p).

    pose_c2p: optional precomputed inverse.  LAPACK's fp32 inverse differs in the
    last bit from host to host, so bit-exact comparisons against a fixture pass the
    inverse that was recorded with it."""
    if pose_c2p is None:
        pose_c2p = torch.linalg.inv(pose_p2c)
    grid_c2p = coord0TOgrid1(coord_c, pose_c2p, cx, cy)
    Ic = bilinear_sampler(Ip, grid_c2p)
    grid_p2c = coord0TOgrid1(coord_p, pose_p2c, cx, cy)
    vis = gen_visible_mask(grid_p2c, 'all')
    return Ic * vis, grid_p2c, grid_c2p


def zp_cView_to_zc(zp_cView, pose_p2c, xy, pose_c2p=None):
    """geometry_pytorch.py:284-308."""
    A = torch.linalg.inv(pose_p2c) if pose_c2p is None else pose_c2p
    a = A[:, 2, :].reshape(-1, 1, 1, 4)
    den = a[..., 0:1] * xy[..., 0:1] + a[..., 1:2] * xy[..., 1:2] + a[..., 2:3]
    return (zp_cView - a[..., 3:4]) / den


# --------------------------------------------------------------------------- #
# the two callers of the path, reduced to their tensor math
# --------------------------------------------------------------------------- #
def image_formation(pattern, z_p, coord_p, coord_c, pose_p2c, psfs, cx, cy, is_conv_psf=True, pose_c2p=None):
    """codePattern -> warp_p2c, the part of forward_model that the captures/s
    metric counts; train_pytorch.py:94-97."""
    Ip_coded = codePattern(pattern, z_p, psfs, is_conv_psf)
    Ic, grid_p2c, grid_c2p = warp_p2c(Ip_coded, coord_p, coord_c, pose_p2c, cx, cy, pose_c2p)
    return Ip_coded, Ic, grid_p2c, grid_c2p


def reproject_chain(zp_cView, pose_p2c, xy, IpGT, cx, cy, pose_c2p=None):
    """zp -> zc -> scene points -> projector pixel -> sampled pattern; the
    differentiable part of reProj_IpGT, train_pytorch.py:189-202.  zp_cView / xy may hold B_sub maps per
    sample: the inverse pose and the pattern are then repeated (:198, :202), and zp_cView_to_zc broadcasts
    the pose the way its own expression does (one sample only, geometry_pytorch.py:301-306)."""
    if pose_c2p is None:
        pose_c2p = torch.linalg.inv(pose_p2c)
    B_sub = zp_cView.shape[0] // pose_p2c.shape[0]
    zc = zp_cView_to_zc(zp_cView, pose_p2c, xy, pose_c2p)
    ray = torch.cat([xy, torch.ones_like(zc)], dim=-1)
    coord_c = points_in_camera_coords(zc, ray, allBatch=True).permute(0, 3, 1, 2)
    grid_c2p = coord0TOgrid1(coord_c, pose_c2p.repeat(B_sub, 1, 1), cx, cy)
    return zc, grid_c2p, bilinear_sampler(IpGT.repeat(B_sub, 1, 1, 1), grid_c2p)


# --------------------------------------------------------------------------- #
# the two callers as the training script composes them (pinned by g9_callers.npz, which holds what
# the reference's own forward_model / reProj_IpGT returned at 800x1200)
# --------------------------------------------------------------------------- #
def image_gradients(img):
    """Forward differences, zero in the last row / column; utils_pytorch.py:206-222."""
    dy = F.pad(img[:, 1:] - img[:, :-1], (0, 0, 0, 0, 0, 1))
    dx = F.pad(img[:, :, 1:] - img[:, :, :-1], (0, 0, 0, 1, 0, 0))
    return dy, dx


def withReflectance(I, depth):
    """I * (0.8 cos(atan(|grad z| / 100 um)) + 0.2); utils_pytorch.py:251-280."""
    dy, dx = image_gradients(depth)
    theta = torch.atan(torch.sqrt(dy ** 2 + dx ** 2) / 100e-6)
    return I * (torch.cos(theta) * 0.8 + 0.2)


def cost_rms_mask(GT, hat, mask, eps=1e-6):
    """utils_pytorch.py:148-163."""
    return torch.sqrt(torch.sum(torch.square(mask * (GT - hat))) / torch.sum(mask).clamp(min=eps) + eps)


def cost_L1_mask(GT, hat, mask):
    """utils_pytorch.py:166-177."""
    return torch.sum(mask * torch.abs(GT - hat)) / (torch.sum(mask) + 1)


def erode(mask, k):
    """1 where the whole k x k neighbourhood is 1; train_pytorch.py:184-186, :222-224."""
    m = F.avg_pool2d(mask.permute(0, 3, 1, 2), kernel_size=k, stride=1, padding=k // 2).permute(0, 2, 3, 1)
    return (m > 0.999).float()


def centre_crop(t, N):
    """multi_rand_crop with the centre offset forward_model passes; train_pytorch.py:56-73, :142-152."""
    H, W = t.shape[1:3]
    h0, w0 = (H - N) // 2, (W - N) // 2
    return t[:, h0:h0 + N, w0:w0 + N, :]


def forward_model(z_p, z_c, pose_p2c, pattern, psfs, noise, is_conv_psf=True, crop=512, cx=600.0, cy=400.0,
                  pose_c2p=None):
    """The simulated capture, train_pytorch.py:79-162, with the pattern (gen_pattern's result, :90) and
    the three random draws (:100, :102: offset [B,1,1,1] and scale [B,1,1,1] uniform in [0,1), gauss
    [B,H,W,1] standard normal) passed in.  Returns the reference's eleven outputs in its order."""
    B, H, W, _ = z_p.shape
    coord_p = z2pointcloud(z_p)                                               # :86-87
    coord_c = z2pointcloud(z_c)
    Ip_ref = withReflectance(pattern, z_p)                                    # :91
    Ip_coded = codePattern(Ip_ref, z_p, psfs, is_conv_psf)                    # :94
    Ic, grid_p2c, grid_c2p = warp_p2c(Ip_coded, coord_p, coord_c, pose_p2c, cx, cy, pose_c2p)   # :97
    Ic_exp = Ic + noise["offset"] * 0.05 + noise["gauss"] * 0.005             # :100
    Ic_max = torch.max(torch.max(Ic_exp, dim=1, keepdim=True)[0], dim=2, keepdim=True)[0]       # :101
    Ic_scaled = torch.clamp(Ic_exp / Ic_max * (noise["scale"] * 0.3 + 0.7), 0.0, 1.0)           # :102-104
    vis = gen_visible_mask(grid_p2c, 'all')                                   # :107
    xy = torch.from_numpy(pixel_to_ray_array(W, H)[:, :, 0:2]).float().unsqueeze(0).repeat(B, 1, 1, 1)   # :110-112
    xyz_cView = bilinear_sampler(torch.cat([xy, z_p], dim=-1), grid_c2p) * vis                  # :115-116
    Ic_mask = (Ic_scaled > 0.05).float() * vis                                # :118
    c = (lambda t: centre_crop(t, crop)) if crop else (lambda t: t)           # :138-160
    return (c(Ic_scaled), c(Ic_mask), c(xyz_cView), c(xy), c(z_c), c(z_p), pose_p2c, z_p, c(Ic), Ip_coded, Ic_scaled)


def reProj_IpGT(zp_cView, pose_p2c, xy, Ic, zcGT, maskC, IpGT, cx=600.0, cy=400.0, pose_c2p=None):
    """The re-projection loss, train_pytorch.py:179-211 (one crop per sample, B_sub = 1)."""
    me = erode(maskC, 5)                                                      # :184-186
    zc, _, Ic_reProj = reproject_chain(zp_cView * me, pose_p2c, xy, IpGT, cx, cy, pose_c2p)     # :188-202
    loss_zc_rms = cost_rms_mask(zcGT, zc, me)                                 # :205-207
    loss_Ic_rms = cost_rms_mask(Ic, Ic_reProj, me)
    loss_Ic_L1 = cost_L1_mask(Ic, Ic_reProj, me)
    return loss_Ic_L1 + loss_zc_rms, loss_zc_rms, loss_Ic_rms, loss_Ic_L1     # :209-211
