"""Drop-in for the reference's geometry_pytorch.py: same function names,
argument order and defaults, backed by the gfx950 kernels in fc3d.

Put this directory on sys.path ahead of the reference checkout and the import
at train_pytorch.py:23-25 picks this module up unchanged.

Differences a caller can observe:
  * tensors must be on a GPU;
  * pixel_to_ray_array is vectorised and cached (the reference's per-pixel
    python loop costs >1 s per call at 1200x800) and returns a read-only array;
  * inverse(pose) is computed in float64 and rounded once instead of by LAPACK
    in fp32, so grids that involve the inverse can differ in the last bit;
  * bilinear_sampler indexes with integers; the reference builds flat indices
    in fp32, which goes wrong once B*H*W exceeds 2**24 (B > 64 at 512x512);
  * gradients: everything the reference's autograd differentiates -- images, coordinates, depths, rays
    (per-sample maps), the pose through coord_transform, coord0TOgrid1, warp_p2c and zp_cView_to_zc (xy
    too); the intrinsic matrix is a constant as in the reference.
"""
from __future__ import annotations

import weakref

import numpy as np
import torch

from fc3d import ops as _ops

H = 800
W = 1200

_FOCAL = 1666.67


def pixel_to_ray(pixel, pixel_width=1200, pixel_height=800):
    """(x, y) pixel -> ray direction with z = 1 (geometry_pytorch.py:15-27)."""
    x, y = pixel
    x_vect = pixel_width / (2 * _FOCAL) * ((2.0 * ((x + 0.5) / pixel_width)) - 1.0)
    y_vect = pixel_height / (2 * _FOCAL) * ((2.0 * ((y + 0.5) / pixel_height)) - 1.0)
    return (x_vect, y_vect, 1.0)


_ray_tables: dict = {}
# id(table) -> (weakref to table, width, height): lets points_in_camera_coords recognise a
# table made here and use its separable form instead of uploading H*W*3 floats per call
_ray_registry: dict = {}
_ray_device: dict = {}


def pixel_to_ray_array(width=1200, height=800):
    """float64 [height, width, 3] table of pixel_to_ray over the image
    (geometry_pytorch.py:30-41), evaluated on index vectors and cached."""
    key = (int(width), int(height))
    tab = _ray_tables.get(key)
    if tab is None:
        xs = np.arange(width, dtype=np.float64)
        ys = np.arange(height, dtype=np.float64)
        rx = pixel_to_ray((xs, 0.0), pixel_width=width, pixel_height=height)[0]
        ry = pixel_to_ray((0.0, ys), pixel_width=width, pixel_height=height)[1]
        tab = np.empty((height, width, 3), dtype=np.float64)
        tab[..., 0] = rx[None, :]
        tab[..., 1] = ry[:, None]
        tab[..., 2] = 1.0
        tab.flags.writeable = False
        _ray_tables[key] = tab
        _ray_registry[id(tab)] = (weakref.ref(tab), int(width), int(height))
    return tab


def _separable_rays(width: int, height: int, device):
    key = (width, height, str(device))
    hit = _ray_device.get(key)
    if hit is None:
        tab = pixel_to_ray_array(width, height)
        rx = torch.from_numpy(np.ascontiguousarray(tab[0, :, 0])).float().to(device)
        ry = torch.from_numpy(np.ascontiguousarray(tab[:, 0, 1])).float().to(device)
        hit = (rx, ry)
        _ray_device[key] = hit
    return hit


def points_in_camera_coords(z_map, pixel_to_ray_array, allBatch=False):
    """z_map [B,H,W,1] x rays -> homogeneous scene points [B,H,W,4]
    (geometry_pytorch.py:44-65).  rays: numpy [H,W,3], or with allBatch a
    tensor [B,H,W,3]."""
    rays = pixel_to_ray_array
    if z_map.shape[0] == 0:
        return z_map.new_zeros((0, z_map.shape[1], z_map.shape[2], 4))
    if allBatch:
        return _ops.PointsFn.apply(z_map, None, None, rays.float())
    reg = _ray_registry.get(id(rays))
    if reg is not None and reg[0]() is rays:
        rx, ry = _separable_rays(reg[1], reg[2], z_map.device)
        pts = _ops.PointsFn.apply(z_map, rx, ry, None)
        if z_map.dim() == 4 and z_map.is_contiguous() and z_map.dtype == torch.float32:
            # what warp_p2c needs to read the depth plane instead of these points (see fc3d.ops)
            _ops.note_points_from_depth(pts, z_map.detach(), rx, ry)
        return pts
    dense = torch.from_numpy(np.ascontiguousarray(rays)).float().to(z_map.device)
    return _ops.PointsFn.apply(z_map, None, None, dense)


def get_intrinsic(B, device='cuda'):
    """Intrinsic matrix [B,4,4] of the 1200x800 renders (geometry_pytorch.py:238-252)."""
    return _intrinsic_cached(B, torch.device(device)).clone()


_intrinsic_cache: dict = {}


def _intrinsic_cached(B: int, device: torch.device) -> torch.Tensor:
    key = (int(B), str(device))
    K = _intrinsic_cache.get(key)
    if K is None:
        M = np.diag([_FOCAL, _FOCAL, 1, 1])
        M[0][2] = 1200 / 2
        M[1][2] = 800 / 2
        K = torch.from_numpy(M).float().to(device).unsqueeze(0).repeat(B, 1, 1)
        _intrinsic_cache[key] = K
    return K


_get_intrinsic_builtin = get_intrinsic


def _intrinsics_for(B: int, device: torch.device) -> torch.Tensor:
    """Honours a monkey-patched get_intrinsic (callers rebind it to move the principal
    point); the built-in one is served from the cache without a copy."""
    fn = globals()["get_intrinsic"]
    if fn is _get_intrinsic_builtin:
        return _intrinsic_cached(B, device)
    return fn(B, device=device)


def coord_transform(coord, transform):
    """[B,4,H,W] scene points x [B,4,4] -> pixel coordinates [B,H,W,2]
    (geometry_pytorch.py:68-90)."""
    if coord.shape[0] == 0:
        return coord.new_zeros((0, coord.shape[2], coord.shape[3], 2))
    return _ops.ProjectFn.apply(coord, transform)


def coord0TOgrid1(coord0, pose01):
    """Scene points of view 0 -> pixel coordinates in view 1 (geometry_pytorch.py:93-106)."""
    return _coord0TOgrid1(coord0, pose01, None)


def _coord0TOgrid1(coord0, pose01, K):
    """coord0TOgrid1 with the intrinsic matrix as an argument (None: get_intrinsic's)."""
    if coord0.shape[0] == 0:
        return coord0.new_zeros((0, coord0.shape[2], coord0.shape[3], 2))
    if K is None:
        K = _intrinsics_for(coord0.shape[0], coord0.device)
    if pose01.requires_grad and torch.is_grad_enabled():
        T, _, _ = _ops.PoseTransformsFn.apply(pose01, K)       # K @ pose with the gradient K^T g
    else:
        T, _, _ = _ops.pose_transforms(pose01.detach(), K, want_fwd=True)
    return _ops.ProjectFn.apply(coord0, T)


def bilinear_sampler(imgs, coords):
    """Bilinear sampling with the reference's clamp-then-weight rule
    (geometry_pytorch.py:109-177): imgs [B,Hs,Ws,C], coords [B,Ht,Wt,2] (x, y)."""
    if imgs.shape[0] == 0:
        return imgs.new_zeros((0, coords.shape[1], coords.shape[2], imgs.shape[3]))
    return _ops.BilinearFn.apply(imgs, coords)


def gen_visible_mask(grid, method='round'):
    """Forward-splat visibility mask [B,H,W,1] (geometry_pytorch.py:180-222)."""
    if method == 'all':
        m = 0
    elif method == 'round':
        m = 1
    else:
        # the reference falls through both branches and trips over an unset local
        raise UnboundLocalError("local variable 'grid_idx_1d' referenced before assignment")
    if grid.shape[0] == 0:
        return grid.new_zeros((0, grid.shape[1], grid.shape[2], 1))
    return _ops.visible_mask(grid, m)


def z2pointcloud(z):
    """Depth map [B,H,W,1] -> scene points [B,4,H,W] using the 1200x800 ray table
    (geometry_pytorch.py:225-235), so like the reference it only accepts 800x1200 input."""
    return points_in_camera_coords(z, pixel_to_ray_array()).permute(0, 3, 1, 2)


def warp_p2c(Ip, coord_p, coord_c, pose_p2c):
    """Projector image -> camera view with occluded pixels blacked out
    (geometry_pytorch.py:255-281).  Returns (Ic_masked, grid_p2c, grid_c2p)."""
    return _warp_p2c(Ip, coord_p, coord_c, pose_p2c, None)


def _warp_p2c(Ip, coord_p, coord_c, pose_p2c, K):
    """warp_p2c with the intrinsic matrix as an argument (None: get_intrinsic's)."""
    if Ip.shape[0] == 0:                          # empty batch
        e = coord_p.new_zeros((0, coord_p.shape[2], coord_p.shape[3], 2))
        return Ip.clone(), e, e.clone()
    if torch.is_grad_enabled() and (coord_p.requires_grad or coord_c.requires_grad or pose_p2c.requires_grad):
        # rare differentiable-geometry case: compose the individual operators
        if K is None:
            K = _intrinsics_for(coord_p.shape[0], coord_p.device)
        if pose_p2c.requires_grad:
            T_p2c, T_c2p, _ = _ops.PoseTransformsFn.apply(pose_p2c, K)
        else:
            T_p2c, T_c2p, _ = _ops.pose_transforms(pose_p2c, K, want_fwd=True, want_inv=True)
        grid_c2p = _ops.ProjectFn.apply(coord_c, T_c2p)
        Ic = _ops.BilinearFn.apply(Ip, grid_c2p)
        grid_p2c = _ops.ProjectFn.apply(coord_p, T_p2c)
        return Ic * _ops.visible_mask(grid_p2c, 0), grid_p2c, grid_c2p
    if K is None:
        K = _intrinsics_for(coord_p.shape[0], coord_p.device)
    # two launches: the projection kernel forms K*pose and K*inverse(pose) itself
    return _ops.WarpP2CPoseFn.apply(Ip, coord_p, coord_c, pose_p2c, K)


def warp_p2c_begin(coord_p, pose_p2c, K=None):
    """First half of warp_p2c for callers that want to overlap it with the production of Ip: the forward
    warp of coord_p and the visibility splat (geometry_pytorch.py:276-277) depend on neither Ip nor
    coord_c.  Issue it on another stream, order that stream before warp_p2c_finish.  No gradients
    w.r.t. coord_p / pose on this route (use warp_p2c for those)."""
    if K is None:
        K = _intrinsics_for(coord_p.shape[0], coord_p.device)
    return _ops.warp_p2c_begin(coord_p, pose_p2c, K)


def warp_p2c_finish(Ip, coord_c, begun):
    """Second half: (Ic_masked, grid_p2c, grid_c2p) exactly as warp_p2c returns them."""
    Ic, grid_c2p = _ops.WarpSampleFn.apply(Ip, coord_c, begun)
    return Ic, begun.grid_p2c, grid_c2p


def zp_cView_to_zc(zp_cView, pose_p2c, xy):
    """Projector-frame depth seen from the camera -> camera-frame depth
    (geometry_pytorch.py:284-308)."""
    if zp_cView.shape[0] == 0:
        return zp_cView.clone()
    if pose_p2c.requires_grad and torch.is_grad_enabled():
        inv = _ops.PoseInverseFn.apply(pose_p2c)
    else:
        _, _, inv = _ops.pose_transforms(pose_p2c.detach(), None, want_a=True)
    B = zp_cView.shape[0]
    if inv.shape[0] != B:
        # the reference's expression broadcasts [B,1,1,1] pose entries against the maps (:301-306): one pose for
        # several maps (reProj_IpGT with B_sub crops of one sample, train_pytorch.py:189-194) is the case that works
        if inv.shape[0] != 1:
            raise RuntimeError(f"The size of tensor a ({inv.shape[0]}) must match the size of tensor b ({B}) at "
                               "non-singleton dimension 0")
        inv = inv.expand(B, 4, 4)
    return _ops.ZpToZcFn.apply(zp_cView, xy, inv)
