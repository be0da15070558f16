"""Host-side counterpart of the two callers of the image-formation path in the
reference's training script: forward_model (train_pytorch.py:79-162) and
reProj_IpGT (:179-211), plus the small loss / normalisation helpers they use
(utils_pytorch.py:148-222, 251-301).  Everything heavy goes through the
drop-in operators of optics_pytorch / geometry_pytorch (HIP kernels); the
glue around them is stock tensor arithmetic.

Unlike the reference these functions take the image size from their inputs
(the reference hard-codes 800x1200 and a 512 centre crop), accept the pattern
and the noise tensors as arguments (so runs are reproducible on any device),
and build the ray table once per size instead of with a per-pixel Python loop.
The reference's intrinsic matrix puts the principal point at (600, 400) whatever
the image size (geometry_pytorch.py:247-249): it is only right for 800x1200, so
every function here takes `intrinsics` and refuses other sizes without it.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

import geometry_pytorch as G
import optics_pytorch as O

from . import synth

_consts: dict = {}


def _rays_xy(H: int, W: int, device) -> torch.Tensor:
    key = ("xy", H, W, str(device))
    if key not in _consts:
        ray = G.pixel_to_ray_array(W, H)[:, :, 0:2]
        _consts[key] = torch.from_numpy(np.ascontiguousarray(ray)).float().to(device)
    return _consts[key]


def dot_pattern(H: int, W: int, device) -> torch.Tensor:
    """[1,H,W,1] binary dot array ('dotArray' mode of utils_pytorch.py:85-137), cached per size."""
    key = ("dots", H, W, str(device))
    if key not in _consts:
        _consts[key] = torch.from_numpy(synth.dot_pattern(H, W)).to(device).view(1, H, W, 1)
    return _consts[key]


def image_gradients(img):
    """Forward differences padded with a zero last row / column; utils_pytorch.py:206-222."""
    dy = F.pad(img[:, 1:] - img[:, :-1], (0, 0, 0, 0, 0, 1))
    dx = F.pad(img[:, :, 1:] - img[:, :, :-1], (0, 0, 0, 1, 0, 0))
    return dy, dx


def with_reflectance(I, depth):
    """I * (0.8 cos(atan(|grad z| / 100um)) + 0.2); utils_pytorch.py:251-280."""
    dy, dx = image_gradients(depth)
    theta = torch.atan(torch.sqrt(dy ** 2 + dx ** 2) / 100e-6)
    return I * (torch.cos(theta) * 0.8 + 0.2)


def _native_box(t, k):
    """The box-filter kernel serves one-channel GPU images that carry no gradient."""
    return (t.is_cuda and t.dim() == 4 and t.shape[3] == 1 and t.shape[0] > 0 and k % 2 == 1 and k <= 33
            and t.dtype == torch.float32 and not (torch.is_grad_enabled() and t.requires_grad))


def local_norm(Im, kernelsize=17, constant=0.01):
    """Im / (box mean + c); utils_pytorch.py:285-301."""
    if _native_box(Im, kernelsize):
        from . import ops
        return ops.box_filter(Im, kernelsize, "local_norm", constant)
    mean = F.avg_pool2d(Im.permute(0, 3, 1, 2), kernelsize, stride=1, padding=kernelsize // 2).permute(0, 2, 3, 1)
    return Im / (mean + constant)


def _same(v):
    return v


def cost_rms_mask(GT, hat, mask, eps=1e-6, reduce=_same):
    """utils_pytorch.py:148-163.  reduce: applied to the numerator sum and to the mask count
    (fc3d.dist.sum_over_ranks gives the loss over the global batch of a data-parallel step)."""
    return torch.sqrt(reduce(torch.sum(torch.square(mask * (GT - hat)))) / reduce(torch.sum(mask)).clamp(min=eps) + eps)


def cost_L1_mask(GT, hat, mask, reduce=_same):
    """utils_pytorch.py:166-177."""
    return reduce(torch.sum(mask * torch.abs(GT - hat))) / (reduce(torch.sum(mask)) + 1)


def erode(mask, k):
    """1 where the whole k x k neighbourhood is 1 (avg-pool > 0.999); train_pytorch.py:184-186."""
    if _native_box(mask, k):
        from . import ops
        return ops.box_filter(mask, k, "erode")
    m = F.avg_pool2d(mask.permute(0, 3, 1, 2), kernel_size=k, stride=1, padding=k // 2).permute(0, 2, 3, 1)
    return (m > 0.999).float()


def cost_grad_mask(GT, hat, mask, reduce=_same):
    me = erode(mask, 9)
    GTy, GTx = image_gradients(GT)
    hy, hx = image_gradients(hat)
    return cost_rms_mask(GTx, hx, me, reduce=reduce) + cost_rms_mask(GTy, hy, me, reduce=reduce)


def centre_crop(t, N):
    H, W = t.shape[1:3]
    if N is None or (N >= H and N >= W):
        return t
    h0, w0 = (H - N) // 2, (W - N) // 2
    return t[:, h0:h0 + N, w0:w0 + N, :]


def resolve_intrinsics(intrinsics, B, H, W, dev):
    """[B,4,4] intrinsic matrices for images of H x W.  intrinsics: None, a (cx, cy) principal point
    (focal length 1666.67 as in the reference), or a [4,4] / [B,4,4] tensor.  None means the
    reference's matrix and is accepted only for 800x1200 images -- or when the caller has rebound
    geometry_pytorch.get_intrinsic, whose result is then used as it would be by the drop-in operators."""
    if intrinsics is None:
        if G.get_intrinsic is not G._get_intrinsic_builtin or (H, W) == (800, 1200):
            return G._intrinsics_for(B, dev)
        raise ValueError(f"images of {H}x{W}: the reference's intrinsic matrix (principal point 600, 400) is for 800x1200 "
                         "only; pass intrinsics=(cx, cy) or a 4x4 matrix")
    if isinstance(intrinsics, torch.Tensor):
        K = intrinsics.to(device=dev, dtype=torch.float32)
        if K.dim() == 2:
            K = K.unsqueeze(0)
        if tuple(K.shape[1:]) != (4, 4) or K.shape[0] not in (1, B):
            raise ValueError(f"intrinsics must be [4,4] or [{B},4,4], got {tuple(intrinsics.shape)}")
        return K.expand(B, 4, 4).contiguous()
    cx, cy = intrinsics
    key = ("K", float(cx), float(cy), B, str(dev))
    if key not in _consts:
        M = np.diag([1666.67, 1666.67, 1, 1])
        M[0][2], M[1][2] = cx, cy
        _consts[key] = torch.from_numpy(M).float().to(dev).unsqueeze(0).repeat(B, 1, 1)
    return _consts[key]


def _draw_noise(B, H, W, dev):
    return dict(offset=torch.rand(B, 1, 1, 1, device=dev), gauss=torch.randn(B, H, W, 1, device=dev),
                scale=torch.rand(B, 1, 1, 1, device=dev))


def forward_model(z_p, z_c, pose_p2c, n_layers=21, is_conv_psf=True, crop=None, noise=None, pattern=None, fused=True,
                  intrinsics=None):
    """Simulated capture; train_pytorch.py:79-162.  noise: optional dict(offset [B,1,1,1] in
    [0,1), gauss [B,H,W,1] ~ N(0,1), scale [B,1,1,1] in [0,1)); drawn on the device if absent.
    fused=True (used when nothing asks for a gradient, as in the reference's loop) runs the
    capture kernels: shading, codePattern, warp straight from the depth maps, sensor model;
    fused=False composes the drop-in operators with tensor arithmetic as the reference does."""
    B, H, W, _ = z_p.shape
    dev = z_p.device
    K = resolve_intrinsics(intrinsics, B, H, W, dev) if B > 0 else None
    Ip = dot_pattern(H, W, dev) if pattern is None else pattern
    if noise is None:
        noise = _draw_noise(B, H, W, dev)
    c = lambda t: centre_crop(t, crop)  # noqa: E731
    xy = _rays_xy(H, W, dev).unsqueeze(0).expand(B, -1, -1, -1)
    if fused and B > 0 and not (torch.is_grad_enabled() and any(t.requires_grad for t in (z_p, z_c, pose_p2c, Ip))):
        from . import ops
        rx, ry = G._separable_rays(W, H, dev)
        Ip0 = Ip if Ip.shape[3] == 1 else Ip[..., 0:1]          # codePattern reads channel 0 only
        psf = O._device_psfs(dev) if is_conv_psf else None
        if is_conv_psf and not (ops.code_pattern_shaded_supported(psf) and psf.shape[0] == n_layers):
            Ip_coded = O.codePattern(ops.reflectance(Ip0, z_p), z_p, is_conv_psf=True, n_layers=n_layers)
        else:          # shading folded into the PSF kernel (3x3..7x7) or, in the training mode, into the gating pass
            Ip_coded = ops.code_pattern_shaded(Ip0, z_p, O._layer_grid(n_layers), psf)
        # three launches: forward warp (forms K*pose itself), inverse warp (+ the sensor's peak), sensor model
        Ic, xyz_cView, Ic_scaled, Ic_mask = ops.capture_fwd(Ip_coded, z_p, z_c, rx, ry, pose_p2c, K, noise["offset"],
                                                            noise["gauss"], noise["scale"])
        return (c(Ic_scaled), c(Ic_mask), c(xyz_cView), c(xy), c(z_c), c(z_p), pose_p2c, z_p, c(Ic), Ip_coded,
                Ic_scaled)
    rays = G.pixel_to_ray_array(W, H)
    coord_p = G.points_in_camera_coords(z_p, rays).permute(0, 3, 1, 2)
    coord_c = G.points_in_camera_coords(z_c, rays).permute(0, 3, 1, 2)
    Ip_ref = with_reflectance(Ip, z_p)
    Ip_coded = O.codePattern(Ip_ref, z_p, is_conv_psf=is_conv_psf, n_layers=n_layers)
    Ic, grid_p2c, grid_c2p = G._warp_p2c(Ip_coded, coord_p, coord_c, pose_p2c, K)
    Ic_exp = Ic + noise["offset"] * 0.05 + noise["gauss"] * 0.005
    Ic_max = Ic_exp.amax(dim=(1, 2), keepdim=True)
    Ic_scaled = torch.clamp(Ic_exp / Ic_max * (noise["scale"] * 0.3 + 0.7), 0.0, 1.0)
    vis = G.gen_visible_mask(grid_p2c, 'all')
    xyz_cView = G.bilinear_sampler(torch.cat([xy, z_p], dim=-1), grid_c2p) * vis
    Ic_mask = (Ic_scaled > 0.05).float() * vis
    return (c(Ic_scaled), c(Ic_mask), c(xyz_cView), c(xy), c(z_c), c(z_p), pose_p2c, z_p, c(Ic), Ip_coded, Ic_scaled)


def _per_crop(t, B_sub):
    """pose / pattern of B samples for B_sub crops of each: the reference's .repeat(B_sub, ...) order
    (train_pytorch.py:198, :202), i.e. crop i belongs to sample i % B."""
    return t if B_sub == 1 else t.repeat(B_sub, *([1] * (t.dim() - 1)))


def _crops_per_sample(zp, pose_p2c):
    B_all, B = zp.shape[0], pose_p2c.shape[0]
    if B == 0 or B_all % B:
        raise ValueError(f"{B_all} depth maps for {B} poses: the number of maps must be a multiple of the number of samples")
    return B_all // B


def reproject(zp, pose_p2c, xy, IpGT, fused=True, intrinsics=None):
    """(zc, Ic_reProj) of the re-projection chain, train_pytorch.py:194-202: camera-view projector
    depth -> camera depth -> scene points -> projector pixel -> sampled pattern.  fused=True runs
    the chain as one kernel per direction when only zp asks for a gradient; otherwise (or with
    fused=False) the drop-in operators are composed as the reference does, which differentiates
    everything the reference's autograd does: zp, pose, xy and IpGT.  zp / xy may hold B_sub crops
    per sample (zp.shape[0] = B_sub * pose.shape[0], :189-191).  intrinsics: of the full-size image
    IpGT (zp and xy may be crops of it)."""
    from . import ops
    B_sub = _crops_per_sample(zp, pose_p2c)
    B_all = zp.shape[0]
    K = resolve_intrinsics(intrinsics, B_all, IpGT.shape[1], IpGT.shape[2], zp.device)
    grads_elsewhere = torch.is_grad_enabled() and (xy.requires_grad or IpGT.requires_grad or pose_p2c.requires_grad)
    if fused and not grads_elsewhere:
        _, T_c2p, pose_c2p = ops.pose_transforms(_per_crop(pose_p2c.detach(), B_sub), K, want_inv=True, want_a=True)
        return ops.ReprojectFn.apply(zp, xy, pose_c2p, T_c2p, _per_crop(IpGT, B_sub))
    if B_sub > 1 and pose_p2c.shape[0] > 1:
        zc = G.zp_cView_to_zc(zp, _per_crop(pose_p2c, B_sub), xy)        # the reference's own call only broadcasts one pose
    else:
        zc = G.zp_cView_to_zc(zp, pose_p2c, xy)
    ray = torch.cat([xy, torch.ones_like(zc)], dim=-1)
    coord_c = G.points_in_camera_coords(zc, ray, allBatch=True).permute(0, 3, 1, 2)
    if torch.is_grad_enabled() and pose_p2c.requires_grad:
        pose_c2p = ops.PoseInverseFn.apply(pose_p2c)
    else:          # inverse(pose) once, through the library's own kernel, as a constant
        _, _, pose_c2p = ops.pose_transforms(pose_p2c.detach(), None, want_a=True)
    grid_c2p = G._coord0TOgrid1(coord_c, _per_crop(pose_c2p, B_sub), K)
    return zc, G.bilinear_sampler(_per_crop(IpGT, B_sub), grid_c2p)


def reProj_IpGT(zp_cView, pose_p2c, xy, Ic, zcGT, maskC, IpGT, fused=True, intrinsics=None, reduce=None):
    """Re-projection loss; train_pytorch.py:179-211, B_sub crops per sample included (:189-202).  fused=True:
    erosion, chain, masked sums and the scalar arithmetic of the losses run in the loss kernels when only
    zp_cView asks for a gradient; anything else that does goes through the composed operators.  reduce: None,
    or a callable applied to the masked sums and the mask count before the losses are formed
    (fc3d.dist.sum_over_ranks: the losses of the global batch)."""
    me = erode(maskC, 5)
    grads_elsewhere = torch.is_grad_enabled() and any(
        t.requires_grad for t in (pose_p2c, xy, Ic, zcGT, maskC, IpGT))
    if fused and zp_cView.is_cuda and zp_cView.shape[0] > 0 and not grads_elsewhere:
        from . import ops
        B_sub = _crops_per_sample(zp_cView, pose_p2c)
        K = resolve_intrinsics(intrinsics, zp_cView.shape[0], IpGT.shape[1], IpGT.shape[2], zp_cView.device)
        _, T_c2p, pose_c2p = ops.pose_transforms(_per_crop(pose_p2c.detach(), B_sub), K, want_inv=True, want_a=True)
        L = ops.ReprojLossFn.apply(zp_cView, me, xy, pose_c2p, T_c2p, _per_crop(IpGT, B_sub), Ic, zcGT,
                                   None if reduce is None else (lambda s: reduce(s).detach()))
        return L[0], L[1], L[2], L[3]
    zp = zp_cView * me
    zc, Ic_reProj = reproject(zp, pose_p2c, xy, IpGT, fused, intrinsics)
    r = _same if reduce is None else reduce
    loss_zc_rms = cost_rms_mask(zcGT, zc, me, reduce=r)
    loss_Ic_rms = cost_rms_mask(Ic, Ic_reProj, me, reduce=r)
    loss_Ic_L1 = cost_L1_mask(Ic, Ic_reProj, me, reduce=r)
    return loss_Ic_L1 + loss_zc_rms, loss_zc_rms, loss_Ic_rms, loss_Ic_L1


def _rms_sums(GT, hat, mask):
    """The two sums of cost_rms_mask (utils_pytorch.py:161-163): masked squared error and mask count."""
    return torch.sum(torch.square(mask * (GT - hat))), torch.sum(mask)


def cal_cost(xyz, xyz_hat, mask, reduce=_same):
    """Reconstruction loss; train_pytorch.py:214-230.  reduce (fc3d.dist.sum_over_ranks for the loss over the
    global batch of a data-parallel step) is applied ONCE, to the vector of the eight masked sums and mask counts
    the four RMS terms are made of, not term by term: one collective instead of eight."""
    me = erode(erode(mask, 5), 9)                       # cost_grad_mask erodes the already eroded mask again
    GTy, GTx = image_gradients(xyz[..., 2:3])
    hy, hx = image_gradients(xyz_hat[..., 2:3])
    terms = (_rms_sums(xyz[..., :2], xyz_hat[..., :2], mask), _rms_sums(xyz[..., 2:3], xyz_hat[..., 2:3], mask),
             _rms_sums(GTx, hx, me), _rms_sums(GTy, hy, me))
    S = reduce(torch.stack([v for nd in terms for v in nd]))
    rms = [torch.sqrt(S[2 * i] / S[2 * i + 1].clamp(min=1e-6) + 1e-6) for i in range(4)]
    l_xy, l_z, l_g = rms[0], rms[1], rms[2] + rms[3]
    return l_xy + l_z + l_g, l_xy, l_z, l_g


class TrainStep:
    """One optimisation step of the reference's loop (train_pytorch.py:283-304) on a per-rank batch.
    The gradients of the two U-Nets live in one flat buffer cut into three ranges; each range's
    all-reduce (RCCL) is started from an autograd hook as soon as backward has filled it, so the
    transfers overlap the rest of the backward pass.

    loss: "rank_mean" -- every rank forms the reference's losses on its own batch and the gradients
    are averaged (the usual data-parallel approximation); "global" -- the masked sums and mask
    counts are added over ranks before the losses are formed (utils_pytorch.py:161-163, :177), so
    the step equals a single-process step on the concatenated batch up to BatchNorm statistics,
    which stay per rank."""

    def __init__(self, device, n_layers=21, is_conv_psf=True, crop=None, lr=1e-4, weight_reproj=1.0, intrinsics=None,
                 loss="rank_mean", n_buckets=3, overlap=True):
        from . import dist as fdist
        from . import nets
        if loss not in ("rank_mean", "global"):
            raise ValueError(f"unknown loss mode {loss!r}")
        self.model_xy = nets.UNet_xy().to(device)
        self.model_z = nets.UNet_z().to(device)
        self.intrinsics = intrinsics
        self.params = list(self.model_xy.parameters()) + list(self.model_z.parameters())
        # every rank starts from rank 0's weights and BatchNorm statistics, whatever its own seed was
        fdist.broadcast_module_state([self.model_xy, self.model_z])
        self.bucket = fdist.FlatGradBucket(self.params, n_buckets=n_buckets, overlap=overlap)
        self.opt = torch.optim.Adam(self.params, lr=lr)
        self.n_layers, self.is_conv_psf, self.crop, self.w = n_layers, is_conv_psf, crop, weight_reproj
        self.loss_mode = loss
        self.keep_capture = False          # True: the simulated capture of the last step stays in self.last_capture
        self.last_capture = None
        self._reduce = fdist.sum_over_ranks if loss == "global" else None
        self._world = fdist.dist.get_world_size() if (fdist.dist.is_available() and fdist.dist.is_initialized()) else 1

    def __call__(self, z_p, z_c, pose_p2c, noise=None):
        self.model_xy.train()
        self.model_z.train()
        with torch.no_grad():
            (Ic_s, Ic_m, xyz, xy, zc, zp, pose, _zp_full, Ic, Ip_coded, _) = forward_model(
                z_p, z_c, pose_p2c, self.n_layers, self.is_conv_psf, self.crop, noise, intrinsics=self.intrinsics)
        if self.keep_capture:
            self.last_capture = dict(Ic_scaled=Ic_s, Ic_mask=Ic_m, xyz_cView=xyz, Ic=Ic, Ip_coded=Ip_coded)
        inp = local_norm(Ic_s, 17, 0.01)
        xyz_hat = torch.cat([self.model_xy(inp), self.model_z(inp)], dim=-1)
        red = self._reduce
        loss_recon, *_ = cal_cost(xyz, xyz_hat, Ic_m, reduce=_same if red is None else red)
        loss_reproj, *_ = reProj_IpGT(xyz_hat[..., 2:3], pose, xy, Ic, zc, Ic_m, Ip_coded, intrinsics=self.intrinsics,
                                      reduce=red)
        loss = loss_recon + loss_reproj * self.w
        self.bucket.zero()
        # global mode: every rank holds its share of the gradient of the same loss, the shares add up;
        # the bucket averages, hence the factor
        (loss * self._world if red is not None else loss).backward()
        self.bucket.average(async_op=True)
        self.bucket.wait()
        self.opt.step()
        return loss.detach()
