"""torch.autograd wrappers around the C ABI of libfc3d_hip.so.

PyTorch is used here for device memory, streams and the autograd tape only;
all arithmetic happens in the HIP kernels.  Tensors must live on a GPU: there
is deliberately no CPU path.
"""
from __future__ import annotations

import ctypes
import threading

import numpy as np
import torch

from . import _native
from ._native import LayerGrid

_lib = _native.load()
_check = _native.check


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    # the raw handle of the current stream of the current device; the private accessor skips the
    # Stream object torch.cuda.current_stream() builds (a dozen microseconds per call)
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


class _on_device:
    """`with torch.cuda.device(dev)` only when dev is not the current device already."""
    __slots__ = ("ctx",)

    def __init__(self, dev):
        self.ctx = None if dev.index is None or dev.index == torch.cuda.current_device() else torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            return self.ctx.__exit__(*exc)
        return False


def _need_gpu(*tensors):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("fc3d operators run on the GPU only (got a tensor on %s); "
                               "there is no CPU implementation" % t.device)
        if t.dtype != torch.float32:
            raise TypeError(f"fc3d operators are fp32 only (got {t.dtype})")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"tensors on different devices: {dev} and {t.device}")
    return dev


def _dense(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


# --------------------------------------------------------------------------- #
# optics
# --------------------------------------------------------------------------- #
def layer_grid(n_layers: int, wvl: float, D: float, z0: float, z_min: float, z_max: float) -> LayerGrid:
    """The four fp32 scalars of the depth->layer decision, derived in float64 the
    way optics_pytorch.py:66-67, 82-84 and 40-44 derive them."""
    coeff = 2 * np.pi / wvl * (D / 2) ** 2 / 2
    lo = coeff * (1 / z_max - 1 / z0)
    hi = coeff * (1 / z_min - 1 / z0)
    layers = np.linspace(lo, hi, n_layers)
    step = (layers[-1] - layers[0]) / (n_layers - 1) if n_layers > 1 else float("nan")
    centres = (layers / step).astype(np.float32)
    c0 = np.float32(centres[0])
    # The kernel reconstructs centre l as centre0 + l.  r is integer-valued, so the reference's
    # test |r - centre_l| < 0.01 is the same decision as long as the two differ by far less
    # than 0.01 (linspace makes them differ by rounding noise only, e.g. 1e-9 at the centre).
    if n_layers > 1 and not np.all(np.abs(centres.astype(np.float64) - (float(c0) + np.arange(n_layers))) < 1e-4):
        raise NotImplementedError("layer centres are not one step apart; unsupported optical constants")
    return LayerGrid(np.float32(coeff), np.float32(1 / z0), np.float32(step), c0, n_layers)


# A by-product of a backward pass, handed from the operator that produced a gradient tensor to the
# operator that consumes it: the sampler's image-gradient kernel can sum (gradient * sampled image)
# over its tiles while it writes the gradient, which is exactly the sum(g * out) codePattern's
# backward needs for the gradient through its per-image maximum.  It travels as an attribute of the
# gradient tensor (autograd hands the very same Python object on) and is only honoured if that
# tensor is unmodified and the consumer's own forward result is the image that was sampled;
# anything else (gradients accumulated from several consumers -- a new tensor without the
# attribute --, hooks that rewrite them, other images) takes the ordinary path.
# The hand-over is watched: handover_stats counts offers, sums taken, sums refused (the gradient was modified
# or belongs to another image) and sums lost -- offered for an image whose codePattern backward then ran
# without seeing any offer, which is what a PyTorch that copies gradients between backward functions
# would look like.  The first loss is reported with a warning; the result is right either way (the
# ordinary two-launch backward runs), only a launch slower.
handover_stats = {"offered": 0, "taken": 0, "refused": 0, "lost": 0}
_pending_offers: dict = {}
_warned_lost = False


def _offer_grad_dot(grad: torch.Tensor, img: torch.Tensor, partials: torch.Tensor) -> None:
    grad._fc3d_dot = (grad._version, img.data_ptr(), img._version, tuple(img.shape), partials)
    handover_stats["offered"] += 1
    if len(_pending_offers) > 64:
        _pending_offers.clear()
    _pending_offers[img.data_ptr()] = (img._version, tuple(img.shape))


def _take_grad_dot(grad: torch.Tensor, out: torch.Tensor):
    global _warned_lost
    hit = getattr(grad, "_fc3d_dot", None)
    pending = _pending_offers.pop(out.data_ptr(), None)
    if hit is None:
        if pending is not None and pending == (out._version, tuple(out.shape)) and grad.is_contiguous():
            # a sum was offered for exactly this image, and the gradient that arrived does not carry it: either
            # autograd summed gradients of several consumers (a new tensor: legitimate) or it no longer passes
            # the producer's tensor object on
            handover_stats["lost"] += 1
            if not _warned_lost:
                _warned_lost = True
                import warnings
                warnings.warn("fc3d: the sum(g * out) by-product offered by the sampler's backward did not reach "
                              "codePattern's backward (several consumers of Ip_coded, a gradient hook, or a PyTorch "
                              "that copies gradients between backward functions); using the two-launch backward",
                              RuntimeWarning, stacklevel=3)
        return None
    version, img_ptr, img_version, img_shape, partials = hit
    if (grad._version != version or not grad.is_contiguous() or img_ptr != out.data_ptr()
            or img_version != out._version or img_shape != tuple(out.shape)):
        handover_stats["refused"] += 1
        return None
    handover_stats["taken"] += 1
    return partials


class CodePatternFn(torch.autograd.Function):
    """codePattern on channel 0 of the pattern; optics_pytorch.py:71-121."""

    @staticmethod
    def forward(ctx, pat0, zMap, psf, grid: LayerGrid, conv: bool):
        # pat0: [B,H,W,1] (any strides), zMap: [B,H,W,1], psf: [L,Kh,Kw] dense on the device
        dev = _need_gpu(pat0, zMap, psf)
        B, H, W, _ = pat0.shape
        if tuple(zMap.shape) != (B, H, W, 1):
            raise RuntimeError(f"zMap shape {tuple(zMap.shape)} does not match pattern {tuple(pat0.shape)}")
        z = _dense(zMap)
        L = grid.n_layers
        Kh, Kw = (int(psf.shape[1]), int(psf.shape[2])) if conv else (1, 1)
        out = torch.empty((B, H, W, 1), dtype=torch.float32, device=dev)
        layer = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
        keys = torch.empty(_lib.fc3d_code_pattern_fwd_workspace(B, H, W) // 8, dtype=torch.int64, device=dev)
        mx = torch.empty(B, dtype=torch.float32, device=dev)
        am = torch.empty(B, dtype=torch.int32, device=dev)
        nkeys = ctypes.c_int(0)
        with _on_device(dev):
            s = _stream()
            _check(_lib.fc3d_code_pattern_fwd(s, pat0.data_ptr(), pat0.stride(0), pat0.stride(1), pat0.stride(2),
                                              z.data_ptr(), psf.data_ptr() if conv else None, B, H, W, Kh, Kw,
                                              int(conv), grid, out.data_ptr(), layer.data_ptr(), keys.data_ptr(),
                                              ctypes.byref(nkeys)), "fc3d_code_pattern_fwd")
            _check(_lib.fc3d_code_pattern_norm(s, out.data_ptr(), keys.data_ptr(), nkeys.value, B, H, W,
                                               out.data_ptr(), mx.data_ptr(), am.data_ptr()),
                   "fc3d_code_pattern_norm")
        ctx.save_for_backward(out, layer, mx, am, psf)
        ctx.meta = (B, H, W, L, Kh, Kw, conv)
        ctx.mark_non_differentiable(layer, mx, am)
        ctx.set_materialize_grads(False)      # no zero tensors for the auxiliary outputs
        return out, layer, mx, am

    @staticmethod
    def backward(ctx, g_out, _gl, _gm, _ga):
        if g_out is None:
            return None, None, None, None, None
        out, layer, mx, am, psf = ctx.saved_tensors
        B, H, W, L, Kh, Kw, conv = ctx.meta
        g = _dense(g_out)
        dev = g.device
        gp = torch.empty((B, H, W, 1), dtype=torch.float32, device=dev)
        # did the operator that produced g (the sampler's backward) already form sum(g * out)?
        dot = _take_grad_dot(g_out, out) if _lib.fc3d_code_pattern_bwd_dot_supported(Kh, Kw, int(conv)) else None
        if dot is not None:
            with _on_device(dev):
                _check(_lib.fc3d_code_pattern_bwd_dot(_stream(), g.data_ptr(), dot.data_ptr(), dot.shape[1],
                                                      layer.data_ptr(), mx.data_ptr(), am.data_ptr(),
                                                      psf.data_ptr() if conv else None, B, H, W, L, Kh, Kw, int(conv),
                                                      gp.data_ptr()),
                       "fc3d_code_pattern_bwd_dot")
            return gp, None, None, None, None
        ws = torch.empty(max(1, _lib.fc3d_code_pattern_bwd_workspace(B, H, W) // 4), dtype=torch.float32, device=dev)
        with _on_device(dev):
            _check(_lib.fc3d_code_pattern_bwd(_stream(), g.data_ptr(), out.data_ptr(), layer.data_ptr(),
                                              mx.data_ptr(), am.data_ptr(), psf.data_ptr() if conv else None,
                                              B, H, W, L, Kh, Kw, int(conv), ws.data_ptr(), gp.data_ptr()),
                   "fc3d_code_pattern_bwd")
        return gp, None, None, None, None


# --------------------------------------------------------------------------- #
# geometry
# --------------------------------------------------------------------------- #
def pose_transforms(pose: torch.Tensor, K: torch.Tensor | None, want_fwd=False, want_inv=False, want_a=False,
                    clear: torch.Tensor | None = None):
    """(K@pose, K@inv(pose), inv(pose)) as [B,4,4] tensors (None where not asked for).
    `clear`: a dense tensor the same launch fills with zeros (warp_p2c's visibility mask)."""
    dev = _need_gpu(pose, K)
    if pose.dim() != 3 or tuple(pose.shape[1:]) != (4, 4):
        raise RuntimeError(f"pose must be [B,4,4], got {tuple(pose.shape)}")
    B = pose.shape[0]
    p = _dense(pose)
    k_ptr, k_sb = None, 0
    if K is not None:
        K = _dense(K)
        if K.dim() == 3:
            if K.shape[0] not in (1, B) or tuple(K.shape[1:]) != (4, 4):
                raise RuntimeError(f"intrinsics must be [B,4,4], got {tuple(K.shape)}")
            k_sb = 16 if K.shape[0] == B and B > 1 else 0
        elif tuple(K.shape) != (4, 4):
            raise RuntimeError(f"intrinsics must be [4,4] or [B,4,4], got {tuple(K.shape)}")
        k_ptr = K.data_ptr()
    tf = torch.empty((B, 4, 4), dtype=torch.float32, device=dev) if want_fwd else None
    ti = torch.empty((B, 4, 4), dtype=torch.float32, device=dev) if want_inv else None
    a = torch.empty((B, 4, 4), dtype=torch.float32, device=dev) if want_a else None
    with _on_device(dev):
        _check(_lib.fc3d_pose_transforms(_stream(), p.data_ptr(), k_ptr, k_sb, B,
                                         tf.data_ptr() if want_fwd else None, ti.data_ptr() if want_inv else None,
                                         a.data_ptr() if want_a else None,
                                         clear.data_ptr() if clear is not None else None,
                                         clear.numel() * clear.element_size() if clear is not None else 0),
               "fc3d_pose_transforms")
    return tf, ti, a


def _coord_args(coord: torch.Tensor):
    """Pointer + element strides of a [B,4,H,W] coordinate tensor; broadcast (stride-0)
    or otherwise exotic views are densified first."""
    if coord.dim() != 4 or coord.shape[1] != 4:
        raise RuntimeError(f"coord must be [B,4,H,W], got {tuple(coord.shape)}")
    st = coord.stride()
    if min(st) <= 0 and coord.numel() > 0:
        coord = coord.contiguous()
        st = coord.stride()
    return coord, st


# Scene points that are known to be points_in_camera_coords(z, separable pixel rays) carry a note that says
# so: (the dense depth tensor, its version, ray_x, ray_y, the points' own version).  warp_p2c then reads the
# 4-byte depth instead of the 16-byte stored point and forms the point as it was stored (bit-identical).  The
# note is honoured only while neither tensor has been written in place since (version counters) and only for
# the plain permuted view of the points that the reference's callers pass (geometry_pytorch.py:225-235).
_POINTS_FROM_DEPTH = True
# Under stream capture the note cannot be checked again at replay: a graph whose inputs are refreshed in place
# (coords.copy_(new points)) would go on reading the depth map of the capture.  So a capture uses the stored
# points unless its owner says that the depth maps are its inputs (GraphedImageFormation(points_from_depth=True)).
_POINTS_FROM_DEPTH_IN_CAPTURE = False


def allow_points_from_depth_in_capture(on: bool) -> bool:
    global _POINTS_FROM_DEPTH_IN_CAPTURE
    old, _POINTS_FROM_DEPTH_IN_CAPTURE = _POINTS_FROM_DEPTH_IN_CAPTURE, bool(on)
    return old


def set_points_from_depth(on: bool) -> bool:
    """Switches the depth-plane form of warp_p2c's coordinate reads on or off (default on); returns the
    previous setting."""
    global _POINTS_FROM_DEPTH
    old, _POINTS_FROM_DEPTH = _POINTS_FROM_DEPTH, bool(on)
    return old


def note_points_from_depth(points: torch.Tensor, z_dense: torch.Tensor, ray_x: torch.Tensor, ray_y: torch.Tensor):
    points._fc3d_points = (z_dense, z_dense._version, ray_x, ray_y, points._version)


def _points_note(coord: torch.Tensor):
    """(z [B,H,W,1] dense, ray_x, ray_y) if coord is the untouched [B,4,H,W] view of noted points, else None."""
    if not _POINTS_FROM_DEPTH or coord.requires_grad:
        return None
    if not _POINTS_FROM_DEPTH_IN_CAPTURE and coord.is_cuda and torch.cuda.is_current_stream_capturing():
        return None
    base = coord._base if coord._base is not None else coord
    note = getattr(base, "_fc3d_points", None)
    if note is None:
        return None
    z, zv, rx, ry, pv = note
    if base._version != pv or z._version != zv or base.dim() != 4 or not base.is_contiguous():
        return None
    B, H, W, C = base.shape
    if C != 4 or tuple(coord.shape) != (B, 4, H, W) or coord.stride() != (H * W * 4, 1, W * 4, 4):
        return None
    if coord.data_ptr() != base.data_ptr() or tuple(z.shape) != (B, H, W, 1):
        return None
    return z, rx, ry


class ProjectFn(torch.autograd.Function):
    """coord_transform; geometry_pytorch.py:68-90."""

    @staticmethod
    def forward(ctx, coord, T):
        dev = _need_gpu(coord, T)
        coord, st = _coord_args(coord)
        B, _, H, W = coord.shape
        if tuple(T.shape) != (B, 4, 4):
            raise RuntimeError(f"transform must be [{B},4,4], got {tuple(T.shape)}")
        T = _dense(T)
        grid = torch.empty((B, H, W, 2), dtype=torch.float32, device=dev)
        with _on_device(dev):
            _check(_lib.fc3d_project_fwd(_stream(), coord.data_ptr(), st[0], st[1], st[2], st[3], T.data_ptr(),
                                         B, H, W, grid.data_ptr()), "fc3d_project_fwd")
        ctx.save_for_backward(coord, T)
        return grid

    @staticmethod
    def backward(ctx, g_grid):
        coord, T = ctx.saved_tensors
        B, _, H, W = coord.shape
        g = _dense(g_grid)
        st = coord.stride()
        gc = gT = None
        with _on_device(g.device):
            if ctx.needs_input_grad[0]:
                gc = torch.empty((B, 4, H, W), dtype=torch.float32, device=g.device)
                _check(_lib.fc3d_project_bwd(_stream(), g.data_ptr(), coord.data_ptr(), st[0], st[1], st[2], st[3],
                                             T.data_ptr(), B, H, W, gc.data_ptr()), "fc3d_project_bwd")
            if ctx.needs_input_grad[1]:
                ws = torch.empty(max(1, _lib.fc3d_project_bwd_transform_workspace(B, H, W) // 4), dtype=torch.float32,
                                 device=g.device)
                gT = torch.empty((B, 4, 4), dtype=torch.float32, device=g.device)
                _check(_lib.fc3d_project_bwd_transform(_stream(), g.data_ptr(), coord.data_ptr(), st[0], st[1], st[2],
                                                       st[3], T.data_ptr(), B, H, W, ws.data_ptr(), gT.data_ptr()),
                       "fc3d_project_bwd_transform")
        return gc, gT


class PoseTransformsFn(torch.autograd.Function):
    """(K@pose, K@inverse(pose), inverse(pose)) with gradients w.r.t. the pose (K is data): the
    forward values are those of pose_transforms; the backward is three batched 4x4 products,
    g_pose = K^T g_fwd - A^T (g_A + K^T g_inv) A^T with A = inverse(pose) (the gradient of a matrix
    inverse), evaluated with tensor ops on the [B,4,4] matrices."""

    @staticmethod
    def forward(ctx, pose, K):
        tf, ti, a = pose_transforms(pose, K, want_fwd=True, want_inv=True, want_a=True)
        Kb = K if K.dim() == 3 else K.unsqueeze(0)
        ctx.save_for_backward(a, Kb)
        ctx.set_materialize_grads(False)
        return tf, ti, a

    @staticmethod
    def backward(ctx, g_tf, g_ti, g_a):
        a, Kb = ctx.saved_tensors
        Kt = Kb.transpose(1, 2)
        gp = None
        if g_tf is not None:
            gp = Kt @ g_tf
        ga = g_a
        if g_ti is not None:
            ga = Kt @ g_ti if ga is None else ga + Kt @ g_ti
        if ga is not None:
            at = a.transpose(1, 2)
            term = -(at @ ga @ at)
            gp = term if gp is None else gp + term
        return gp, None


_IMG_GRAD_MODE = "ordered"


def set_image_gradient_mode(mode: str) -> str:
    """How d out / d img of the bilinear sampler (and of warp_p2c) is accumulated for one-channel
    images: "ordered" (default): gather form, no atomics, fixed summation order; "atomic": scatter
    form with global float atomics.  torch.use_deterministic_algorithms(True) always selects
    "ordered".  Returns the previous mode."""
    global _IMG_GRAD_MODE
    if mode not in ("ordered", "atomic"):
        raise ValueError(f"unknown image-gradient mode {mode!r}")
    old, _IMG_GRAD_MODE = _IMG_GRAD_MODE, mode
    return old


def _img_grad_ordered() -> bool:
    return _IMG_GRAD_MODE == "ordered" or torch.are_deterministic_algorithms_enabled()


def _tap_index_buffer(B, Hs, Ws, Ht, Wt, dev):
    """Storage for the tap index of a [B,Ht,Wt,2] grid into a one-channel [B,Hs,Ws] image, or None
    where the gather form does not apply."""
    nbytes = _lib.fc3d_bilinear_bwd_img_workspace(B, Hs, Ws, 1, Ht, Wt)
    return torch.empty(nbytes // 8, dtype=torch.int64, device=dev) if nbytes else None


def _bilinear_bwd_img(g, grid, mask, B, Hs, Ws, C, Ht, Wt, index=None, img=None):
    dev = g.device
    ready = index is not None
    if not ready and C == 1 and _img_grad_ordered():
        index = _tap_index_buffer(B, Hs, Ws, Ht, Wt, dev)
    gi = torch.empty((B, Hs, Ws, C), dtype=torch.float32, device=dev)
    if ready and C == 1 and img is not None and img.is_contiguous():
        # gather form + the per-tile sums of gi * img on the side (see _offer_grad_dot)
        n = _lib.fc3d_bilinear_bwd_img_dot_count(B, Hs, Ws)
        partials = torch.empty((B, n), dtype=torch.float32, device=dev)
        with _on_device(dev):
            _check(_lib.fc3d_bilinear_bwd_img_dot(_stream(), g.data_ptr(), grid.data_ptr(), B, Hs, Ws, Ht, Wt,
                                                  index.data_ptr(), img.data_ptr(), gi.data_ptr(), partials.data_ptr()),
                   "fc3d_bilinear_bwd_img_dot")
        _offer_grad_dot(gi, img, partials)
        return gi
    with _on_device(dev):
        _check(_lib.fc3d_bilinear_bwd_img(_stream(), g.data_ptr(), grid.data_ptr(),
                                          mask.data_ptr() if mask is not None else None, B, Hs, Ws, C, Ht, Wt,
                                          index.data_ptr() if index is not None else None, int(ready),
                                          gi.data_ptr()),
               "fc3d_bilinear_bwd_img")
    return gi


class BilinearFn(torch.autograd.Function):
    """bilinear_sampler; geometry_pytorch.py:109-177."""

    @staticmethod
    def forward(ctx, imgs, coords):
        dev = _need_gpu(imgs, coords)
        if imgs.dim() != 4 or coords.dim() != 4 or coords.shape[-1] != 2 or coords.shape[0] != imgs.shape[0]:
            raise RuntimeError(f"bilinear_sampler: imgs {tuple(imgs.shape)} / coords {tuple(coords.shape)}")
        img, grid = _dense(imgs), _dense(coords)
        B, Hs, Ws, C = img.shape
        _, Ht, Wt, _ = grid.shape
        out = torch.empty((B, Ht, Wt, C), dtype=torch.float32, device=dev)
        with _on_device(dev):
            _check(_lib.fc3d_bilinear_fwd(_stream(), img.data_ptr(), grid.data_ptr(), None, B, Hs, Ws, C, Ht, Wt,
                                          out.data_ptr()), "fc3d_bilinear_fwd")
        ctx.save_for_backward(img, grid)
        return out

    @staticmethod
    def backward(ctx, g_out):
        img, grid = ctx.saved_tensors
        B, Hs, Ws, C = img.shape
        _, Ht, Wt, _ = grid.shape
        g = _dense(g_out)
        gi = gg = None
        if ctx.needs_input_grad[0]:
            gi = _bilinear_bwd_img(g, grid, None, B, Hs, Ws, C, Ht, Wt)
        if ctx.needs_input_grad[1]:
            gg = torch.empty_like(grid)
            with _on_device(g.device):
                _check(_lib.fc3d_bilinear_bwd_grid(_stream(), g.data_ptr(), img.data_ptr(), grid.data_ptr(), None,
                                                   B, Hs, Ws, C, Ht, Wt, gg.data_ptr()), "fc3d_bilinear_bwd_grid")
        return gi, gg


def visible_mask(grid: torch.Tensor, method: int) -> torch.Tensor:
    """gen_visible_mask as a float [B,H,W,1] tensor; geometry_pytorch.py:180-222."""
    dev = _need_gpu(grid)
    if grid.dim() != 4 or grid.shape[-1] != 2:
        raise RuntimeError(f"grid must be [B,H,W,2], got {tuple(grid.shape)}")
    g = _dense(grid.detach())
    B, H, W, _ = g.shape
    out = torch.empty((B, H, W, 1), dtype=torch.float32, device=dev)
    with _on_device(dev):
        _check(_lib.fc3d_visible_mask(_stream(), g.data_ptr(), B, H, W, method, out.data_ptr(), None),
               "fc3d_visible_mask")
    return out


class WarpP2CFn(torch.autograd.Function):
    """warp_p2c with gradient w.r.t. Ip only (coords and pose are data);
    geometry_pytorch.py:255-281."""

    @staticmethod
    def forward(ctx, Ip, coord_p, coord_c, T_p2c, T_c2p, zero_mask):
        # zero_mask: a [B,H,W] uint8 buffer of zeros to use as the visibility mask (see warp_mask_buffer)
        dev = _need_gpu(Ip, coord_p, coord_c, T_p2c, T_c2p)
        coord_p, sp = _coord_args(coord_p)
        coord_c, sc = _coord_args(coord_c)
        B, _, H, W = coord_p.shape
        if tuple(coord_c.shape) != (B, 4, H, W) or tuple(Ip.shape) != (B, H, W, 1):
            raise RuntimeError(f"warp_p2c: Ip {tuple(Ip.shape)}, coord_p {tuple(coord_p.shape)}, "
                               f"coord_c {tuple(coord_c.shape)}")
        ip = _dense(Ip)
        Ic = torch.empty((B, H, W, 1), dtype=torch.float32, device=dev)
        gp2c = torch.empty((B, H, W, 2), dtype=torch.float32, device=dev)
        gc2p = torch.empty((B, H, W, 2), dtype=torch.float32, device=dev)
        have_zero = zero_mask is not None
        if have_zero and (tuple(zero_mask.shape) != (B, H, W) or zero_mask.dtype != torch.uint8 or
                          not zero_mask.is_contiguous() or zero_mask.device != dev):
            raise RuntimeError("warp_p2c: zero_mask must be a dense uint8 [B,H,W] tensor on the inputs' device")
        mask = zero_mask if have_zero else torch.empty((B, H, W), dtype=torch.uint8, device=dev)
        # the sampling kernel also notes where its pixels landed, for the gradient w.r.t. Ip
        index = _tap_index_buffer(B, H, W, H, W, dev) if (ctx.needs_input_grad[0] and _img_grad_ordered()) else None
        with _on_device(dev):
            _check(_lib.fc3d_warp_p2c_fwd(_stream(), ip.data_ptr(), coord_p.data_ptr(), sp[0], sp[1], sp[2], sp[3],
                                          coord_c.data_ptr(), sc[0], sc[1], sc[2], sc[3], T_p2c.data_ptr(),
                                          T_c2p.data_ptr(), B, H, W, Ic.data_ptr(), gp2c.data_ptr(), gc2p.data_ptr(),
                                          mask.data_ptr(), int(have_zero),
                                          index.data_ptr() if index is not None else None), "fc3d_warp_p2c_fwd")
        ctx.save_for_backward(gc2p, mask, index)
        ctx.mark_non_differentiable(gp2c, gc2p, mask)
        ctx.set_materialize_grads(False)      # no zero tensors for the two grids and the mask
        return Ic, gp2c, gc2p, mask

    @staticmethod
    def backward(ctx, g_Ic, _g1, _g2, _g3):
        if g_Ic is None:
            return None, None, None, None, None, None
        gc2p, mask, index = ctx.saved_tensors
        B, H, W, _ = gc2p.shape
        gi = _bilinear_bwd_img(_dense(g_Ic), gc2p, mask, B, H, W, 1, H, W, index=index)
        return gi, None, None, None, None, None


# Visibility-mask buffers that hold zeros between calls: fc3d_warp_p2c_pose_fwd's sampling kernel
# stores zeros back into the mask it has just consumed, so a buffer is cleared once, when it is
# made, and never again.  One buffer per (device, stream, shape, host thread): calls of one thread
# on one stream are ordered, and two host threads never share a buffer.  Where the two halves of a
# warp run on different streams (warp_p2c_begin / WarpSampleFn), the second half leaves an event
# behind and the next checkout of the buffer waits for it, so the next splat cannot overtake the
# kernel that is still reading and clearing the mask.  A buffer that a stream capture has used stays
# for good (the graph replays write to its address); a capture only uses a buffer that exists
# already, i.e. one made by eager calls on the same stream, and orders its streams itself.
# set_mask_pooling(False) opts out: every call then gets a fresh mask that the kernels clear.
_zero_masks: dict = {}
_zero_masks_pinned: set = set()
_zero_mask_events: dict = {}
_MASK_POOLING = True


def set_mask_pooling(on: bool) -> bool:
    """Whether warp_p2c / the capture simulator keep self-clearing visibility-mask buffers between calls
    (default) or allocate and clear one per call.  Returns the previous setting."""
    global _MASK_POOLING
    old, _MASK_POOLING = _MASK_POOLING, bool(on)
    return old


def _pooled_zero_mask(B, H, W, dev, stream, capturing):
    if not _MASK_POOLING:
        return None, None
    key = (dev.index, stream, B, H, W, threading.get_ident())
    m = _zero_masks.get(key)
    if m is None:
        if capturing:
            return key, None
        for k in [k for k in _zero_masks if k not in _zero_masks_pinned][:max(0, len(_zero_masks) - 7)]:
            del _zero_masks[k]
            _zero_mask_events.pop(k, None)
        m = torch.zeros((B, H, W), dtype=torch.uint8, device=dev)
        _zero_masks[key] = m
    if capturing:
        _zero_masks_pinned.add(key)
    else:
        ev = _zero_mask_events.pop(key, None)
        if ev is not None:          # last cleared by a kernel on another stream: order this stream behind it
            torch.cuda.current_stream(dev).wait_event(ev)
    return key, m


def _intrinsics_args(K, B):
    """(dense K, element stride between samples) for a [4,4] or [B,4,4] / [1,4,4] intrinsic matrix."""
    K = _dense(K)
    if K.dim() == 3:
        if K.shape[0] not in (1, B) or tuple(K.shape[1:]) != (4, 4):
            raise RuntimeError(f"intrinsics must be [B,4,4], got {tuple(K.shape)}")
        return K, (16 if K.shape[0] == B and B > 1 else 0)
    if tuple(K.shape) != (4, 4):
        raise RuntimeError(f"intrinsics must be [4,4] or [B,4,4], got {tuple(K.shape)}")
    return K, 0


class WarpP2CPoseFn(torch.autograd.Function):
    """warp_p2c from the pose and the intrinsic matrix, gradient w.r.t. Ip only (geometry_pytorch.py:255-281):
    two launches -- the projection kernel forms K*pose and K*inverse(pose) on the side.  With the
    ordered image gradient the visibility mask is dead once the sampling kernel has read it; the
    kernel then stores zeros back and the buffer is reused by the next call without a clear.
    Returns (Ic_masked, grid_p2c, grid_c2p)."""

    last_form = None      # "depth" / "points": which coordinate read the last forward used (for tests and benchmarks)

    @staticmethod
    def forward(ctx, Ip, coord_p, coord_c, pose, K):
        dev = _need_gpu(Ip, coord_p, coord_c, pose, K)
        note_p, note_c = _points_note(coord_p), _points_note(coord_c)
        depth_form = (note_p is not None and note_c is not None and note_p[1] is note_c[1] and note_p[2] is note_c[2])
        coord_p, sp = _coord_args(coord_p)
        coord_c, sc = _coord_args(coord_c)
        B, _, H, W = coord_p.shape
        if tuple(coord_c.shape) != (B, 4, H, W) or tuple(Ip.shape) != (B, H, W, 1):
            raise RuntimeError(f"warp_p2c: Ip {tuple(Ip.shape)}, coord_p {tuple(coord_p.shape)}, "
                               f"coord_c {tuple(coord_c.shape)}")
        if tuple(pose.shape) != (B, 4, 4):
            raise RuntimeError(f"pose must be [{B},4,4], got {tuple(pose.shape)}")
        ip, po = _dense(Ip), _dense(pose)
        K, k_sb = _intrinsics_args(K, B)
        Ic = torch.empty((B, H, W, 1), dtype=torch.float32, device=dev)
        gp2c = torch.empty((B, H, W, 2), dtype=torch.float32, device=dev)
        gc2p = torch.empty((B, H, W, 2), dtype=torch.float32, device=dev)
        tt = torch.empty((2, B, 4, 4), dtype=torch.float32, device=dev)
        want_grad = ctx.needs_input_grad[0]
        ordered = _img_grad_ordered()
        index = _tap_index_buffer(B, H, W, H, W, dev) if (want_grad and ordered) else None
        with _on_device(dev):
            s = _stream()
            # the mask outlives the call only where the atomic form of the gradient reads it again
            key = mask = None
            if not (want_grad and index is None):
                key, mask = _pooled_zero_mask(B, H, W, dev, s, torch.cuda.is_current_stream_capturing())
            pooled = mask is not None
            if not pooled:
                mask = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
            if depth_form:
                status = _lib.fc3d_warp_p2c_pose_depth_fwd(
                    s, ip.data_ptr(), note_p[0].data_ptr(), note_c[0].data_ptr(), note_p[1].data_ptr(),
                    note_p[2].data_ptr(), po.data_ptr(), K.data_ptr(), k_sb, B, H, W, Ic.data_ptr(), gp2c.data_ptr(),
                    gc2p.data_ptr(), mask.data_ptr(), int(pooled), int(pooled),
                    index.data_ptr() if index is not None else None, tt[0].data_ptr(), tt[1].data_ptr(), None)
            else:
                status = _lib.fc3d_warp_p2c_pose_fwd(
                    s, ip.data_ptr(), coord_p.data_ptr(), sp[0], sp[1], sp[2], sp[3], coord_c.data_ptr(), sc[0], sc[1],
                    sc[2], sc[3], po.data_ptr(), K.data_ptr(), k_sb, B, H, W, Ic.data_ptr(), gp2c.data_ptr(),
                    gc2p.data_ptr(), mask.data_ptr(), int(pooled), int(pooled),
                    index.data_ptr() if index is not None else None, tt[0].data_ptr(), tt[1].data_ptr(), None)
            if status != 0 and pooled:
                _zero_masks.pop(key, None)        # its contents are unknown now
            _check(status, "fc3d_warp_p2c_pose_depth_fwd" if depth_form else "fc3d_warp_p2c_pose_fwd")
        WarpP2CPoseFn.last_form = "depth" if depth_form else "points"
        ctx.save_for_backward(gc2p, None if pooled else mask, index, ip if index is not None else None)
        ctx.mark_non_differentiable(gp2c, gc2p)
        ctx.set_materialize_grads(False)
        return Ic, gp2c, gc2p

    @staticmethod
    def backward(ctx, g_Ic, _g1, _g2):
        if g_Ic is None:
            return None, None, None, None, None
        gc2p, mask, index, ip = ctx.saved_tensors
        B, H, W, _ = gc2p.shape
        gi = _bilinear_bwd_img(_dense(g_Ic), gc2p, mask, B, H, W, 1, H, W, index=index, img=ip)
        return gi, None, None, None, None


class WarpBegun:
    """What the first launch of warp_p2c (projection of coord_p + visibility splat) leaves for the second."""
    __slots__ = ("grid_p2c", "mask", "transforms", "pooled", "pool_key", "shape", "stream")


def warp_p2c_begin(coord_p: torch.Tensor, pose: torch.Tensor, K: torch.Tensor) -> WarpBegun:
    """The half of warp_p2c that needs neither the image nor coord_c (geometry_pytorch.py:276-277): grid_p2c,
    the visibility mask, and K*pose / K*inverse(pose) formed on the side.  It may be issued on another stream,
    beside the work that produces the image; the caller orders that stream before WarpSampleFn
    (`main.wait_stream(side)`), nothing more: the ordering of the NEXT first half behind this warp's second half
    (which reads and clears the pooled mask on the main stream) is kept by an event stored with the buffer."""
    dev = _need_gpu(coord_p, pose, K)
    coord_p, sp = _coord_args(coord_p.detach())
    B, _, H, W = coord_p.shape
    if tuple(pose.shape) != (B, 4, 4):
        raise RuntimeError(f"pose must be [{B},4,4], got {tuple(pose.shape)}")
    po = _dense(pose.detach())
    K, k_sb = _intrinsics_args(K, B)
    h = WarpBegun()
    h.shape = (B, H, W)
    h.grid_p2c = torch.empty((B, H, W, 2), dtype=torch.float32, device=dev)
    h.transforms = torch.empty((2, B, 4, 4), dtype=torch.float32, device=dev)
    with _on_device(dev):
        s = _stream()
        h.pool_key = h.mask = None
        if _img_grad_ordered():        # the atomic form of the image gradient reads the mask again: no pooling then
            h.pool_key, h.mask = _pooled_zero_mask(B, H, W, dev, s, torch.cuda.is_current_stream_capturing())
        h.pooled = h.mask is not None
        h.stream = s
        if h.pooled:
            # checked out until the second half has cleared it again: a first half that is never finished
            # must not leave a marked mask behind for the next call
            _zero_masks.pop(h.pool_key, None)
        if not h.pooled:
            h.mask = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
        _check(_lib.fc3d_warp_project_splat(s, coord_p.data_ptr(), sp[0], sp[1], sp[2], sp[3], po.data_ptr(),
                                            K.data_ptr(), k_sb, B, H, W, h.grid_p2c.data_ptr(), h.mask.data_ptr(),
                                            int(h.pooled), h.transforms[0].data_ptr(), h.transforms[1].data_ptr(),
                                            None), "fc3d_warp_project_splat")
    return h


class WarpSampleFn(torch.autograd.Function):
    """The second half of warp_p2c (geometry_pytorch.py:271-273, :279) after warp_p2c_begin: (Ic_masked,
    grid_c2p); gradient w.r.t. Ip only."""

    @staticmethod
    def forward(ctx, Ip, coord_c, begun: WarpBegun):
        dev = _need_gpu(Ip, coord_c)
        coord_c, sc = _coord_args(coord_c)
        B, H, W = begun.shape
        if tuple(coord_c.shape) != (B, 4, H, W) or tuple(Ip.shape) != (B, H, W, 1):
            raise RuntimeError(f"warp_p2c: Ip {tuple(Ip.shape)}, coord_c {tuple(coord_c.shape)} after a first half "
                               f"for {B}x{H}x{W}")
        if begun.mask is None:
            raise RuntimeError("this first half of warp_p2c has been finished already")
        ip = _dense(Ip)
        Ic = torch.empty((B, H, W, 1), dtype=torch.float32, device=dev)
        gc2p = torch.empty((B, H, W, 2), dtype=torch.float32, device=dev)
        want_grad = ctx.needs_input_grad[0]
        index = _tap_index_buffer(B, H, W, H, W, dev) if (want_grad and _img_grad_ordered()) else None
        if want_grad and index is None and begun.pooled:
            raise RuntimeError("the image-gradient mode changed between the two halves of warp_p2c")
        mask, begun.mask = begun.mask, None
        with _on_device(dev):
            s = _stream()
            capturing = torch.cuda.is_current_stream_capturing()
            cross = s != begun.stream
            if cross and not capturing:
                # the first half ran (and allocated) on another stream: tell the allocator who reads
                here = torch.cuda.current_stream()
                begun.transforms.record_stream(here)
                begun.grid_p2c.record_stream(here)
                if not begun.pooled:
                    mask.record_stream(here)
            status = _lib.fc3d_warp_sample(s, ip.data_ptr(), coord_c.data_ptr(), sc[0], sc[1], sc[2], sc[3],
                                           begun.transforms[1].data_ptr(), B, H, W, mask.data_ptr(), int(begun.pooled),
                                           Ic.data_ptr(), gc2p.data_ptr(),
                                           index.data_ptr() if index is not None else None)
            _check(status, "fc3d_warp_sample")
            if begun.pooled:
                # cleared again by the kernel just launched: back to the pool -- with an event if that kernel runs
                # on another stream than the one the buffer belongs to, for the next checkout to wait on
                if cross and not capturing:
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream())
                    _zero_mask_events[begun.pool_key] = ev
                _zero_masks[begun.pool_key] = mask
        ctx.save_for_backward(gc2p, None if begun.pooled else mask, index, ip if index is not None else None)
        ctx.mark_non_differentiable(gc2p)
        ctx.set_materialize_grads(False)
        return Ic, gc2p

    @staticmethod
    def backward(ctx, g_Ic, _g1):
        if g_Ic is None:
            return None, None, None
        gc2p, mask, index, ip = ctx.saved_tensors
        B, H, W, _ = gc2p.shape
        gi = _bilinear_bwd_img(_dense(g_Ic), gc2p, mask, B, H, W, 1, H, W, index=index, img=ip)
        return gi, None, None


class PointsFn(torch.autograd.Function):
    """points_in_camera_coords; geometry_pytorch.py:44-65.  Returns [B,H,W,4].  Differentiable in z_map and
    in the rays (per-sample ray maps as reProj_IpGT builds them from xy, train_pytorch.py:196-197; a
    shared table or the separable form get the sums over what they were broadcast over)."""

    @staticmethod
    def forward(ctx, z_map, ray_x, ray_y, ray_dense):
        dev = _need_gpu(z_map, ray_x, ray_y, ray_dense)
        if z_map.dim() != 4 or z_map.shape[-1] != 1:
            raise RuntimeError(f"z_map must be [B,H,W,1], got {tuple(z_map.shape)}")
        z = _dense(z_map)
        B, H, W, _ = z.shape
        sb = 0
        dense_shape = None
        if ray_dense is not None:
            dense_shape = tuple(ray_dense.shape)
            ray_dense = _dense(ray_dense)
            if tuple(ray_dense.shape[-3:]) != (H, W, 3):
                raise RuntimeError(f"The size of tensor a ({W}) must match the size of tensor b "
                                   f"({ray_dense.shape[-2]}) at non-singleton dimension 2")
            sb = H * W * 3 if (ray_dense.dim() == 4 and ray_dense.shape[0] == B and B > 1) else 0
        elif ray_x.numel() != W or ray_y.numel() != H:
            raise RuntimeError(f"The size of tensor a ({W}) must match the size of tensor b ({ray_x.numel()}) "
                               "at non-singleton dimension 2")
        out = torch.empty((B, H, W, 4), dtype=torch.float32, device=dev)
        ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        with _on_device(dev):
            _check(_lib.fc3d_points_fwd(_stream(), z.data_ptr(), ptr(ray_x), ptr(ray_y), ptr(ray_dense), sb, B, H, W,
                                        out.data_ptr()), "fc3d_points_fwd")
        ray_grad = any(ctx.needs_input_grad[1:])
        ctx.save_for_backward(ray_x, ray_y, ray_dense, z if ray_grad else None)
        ctx.shape = (B, H, W, sb, dense_shape)
        return out

    @staticmethod
    def backward(ctx, g_out):
        ray_x, ray_y, ray_dense, z = ctx.saved_tensors
        B, H, W, sb, dense_shape = ctx.shape
        g = _dense(g_out)
        gz = g_rx = g_ry = g_rd = None
        ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        with _on_device(g.device):
            if ctx.needs_input_grad[0]:
                gz = torch.empty((B, H, W, 1), dtype=torch.float32, device=g.device)
                _check(_lib.fc3d_points_bwd(_stream(), g.data_ptr(), ptr(ray_x), ptr(ray_y), ptr(ray_dense), sb, B, H, W,
                                            gz.data_ptr()), "fc3d_points_bwd")
            if any(ctx.needs_input_grad[1:]):
                gr = torch.empty((B, H, W, 3), dtype=torch.float32, device=g.device)       # per sample and pixel
                _check(_lib.fc3d_points_bwd_rays(_stream(), g.data_ptr(), z.data_ptr(), B, H, W, gr.data_ptr()),
                       "fc3d_points_bwd_rays")
                if ray_dense is not None and ctx.needs_input_grad[3]:
                    # a table shared by the batch collects the sum over the samples it was broadcast to
                    g_rd = gr if sb else gr.sum(0, keepdim=len(dense_shape) == 4)
                    g_rd = g_rd.reshape(dense_shape)
                if ray_dense is None:
                    if ctx.needs_input_grad[1]:
                        g_rx = gr[..., 0].sum((0, 1)).reshape(ray_x.shape)
                    if ctx.needs_input_grad[2]:
                        g_ry = gr[..., 1].sum((0, 2)).reshape(ray_y.shape)
        return gz, g_rx, g_ry, g_rd


class PoseInverseFn(torch.autograd.Function):
    """inverse(pose) [B,4,4] (fp64 Gauss-Jordan, rounded once) with its gradient -A^T g_A A^T."""

    @staticmethod
    def forward(ctx, pose):
        _, _, a = pose_transforms(pose, None, want_a=True)
        ctx.save_for_backward(a)
        return a

    @staticmethod
    def backward(ctx, g_a):
        a, = ctx.saved_tensors
        at = a.transpose(1, 2)
        return -(at @ g_a @ at)


class ZpToZcFn(torch.autograd.Function):
    """zp_cView_to_zc given inverse(pose); geometry_pytorch.py:284-308.  Differentiable in zp, xy and the
    inverse pose (PoseInverseFn carries that on to the pose)."""

    @staticmethod
    def forward(ctx, zp, xy, inv):
        dev = _need_gpu(zp, xy, inv)
        B, H, W, _ = zp.shape
        if tuple(xy.shape) != (B, H, W, 2) or tuple(inv.shape) != (B, 4, 4):
            raise RuntimeError(f"zp_cView_to_zc: zp {tuple(zp.shape)}, xy {tuple(xy.shape)}, pose {tuple(inv.shape)}")
        z, r, a = _dense(zp), _dense(xy), _dense(inv)
        out = torch.empty_like(z)
        with _on_device(dev):
            _check(_lib.fc3d_zp_to_zc_fwd(_stream(), z.data_ptr(), r.data_ptr(), a.data_ptr(), B, H, W,
                                          out.data_ptr()), "fc3d_zp_to_zc_fwd")
        full = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        ctx.save_for_backward(r, a, out if full else None)
        return out

    @staticmethod
    def backward(ctx, g):
        r, inv, zc = ctx.saved_tensors
        B, H, W, _ = r.shape
        g = _dense(g)
        dev = g.device
        if zc is None:
            gz = torch.empty_like(g)
            with _on_device(dev):
                _check(_lib.fc3d_zp_to_zc_bwd(_stream(), g.data_ptr(), r.data_ptr(), inv.data_ptr(), B, H, W,
                                              gz.data_ptr()), "fc3d_zp_to_zc_bwd")
            return gz, None, None
        gz = torch.empty_like(g) if ctx.needs_input_grad[0] else None
        gxy = torch.empty_like(r) if ctx.needs_input_grad[1] else None
        ginv = ws = None
        if ctx.needs_input_grad[2]:
            ginv = torch.empty((B, 4, 4), dtype=torch.float32, device=dev)
            ws = torch.empty(_lib.fc3d_zp_to_zc_bwd_full_workspace(B, H, W) // 4, dtype=torch.float32, device=dev)
        ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        with _on_device(dev):
            _check(_lib.fc3d_zp_to_zc_bwd_full(_stream(), g.data_ptr(), zc.data_ptr(), r.data_ptr(), inv.data_ptr(), B, H, W,
                                               ptr(gz), ptr(gxy), ptr(ws), ptr(ginv)), "fc3d_zp_to_zc_bwd_full")
        return gz, gxy, ginv


class ReprojectFn(torch.autograd.Function):
    """zp_cView -> (zc, pattern sampled at the re-projected position): the differentiated chain
    of reProj_IpGT (train_pytorch.py:194-202) in one kernel per direction.  inv = inverse(pose),
    T = K @ inverse(pose); gradient w.r.t. zp only."""

    @staticmethod
    def forward(ctx, zp, xy, inv, T, img):
        dev = _need_gpu(zp, xy, inv, T, img)
        B, H, W, _ = zp.shape
        if (tuple(zp.shape) != (B, H, W, 1) or tuple(xy.shape) != (B, H, W, 2) or img.dim() != 4
                or img.shape[0] != B or img.shape[3] != 1
                or tuple(inv.shape) != (B, 4, 4) or tuple(T.shape) != (B, 4, 4)):
            raise RuntimeError(f"reproject: zp {tuple(zp.shape)}, xy {tuple(xy.shape)}, img {tuple(img.shape)}, "
                               f"inverse pose {tuple(inv.shape)}, transform {tuple(T.shape)}")
        z, r, a, t, im = _dense(zp), _dense(xy), _dense(inv), _dense(T), _dense(img)
        zc = torch.empty_like(z)
        out = torch.empty_like(z)
        with _on_device(dev):
            _check(_lib.fc3d_reproject_fwd(_stream(), z.data_ptr(), r.data_ptr(), a.data_ptr(), t.data_ptr(),
                                           im.data_ptr(), B, H, W, im.shape[1], im.shape[2], zc.data_ptr(),
                                           out.data_ptr(), None),
                   "fc3d_reproject_fwd")
        ctx.save_for_backward(zc, r, a, t, im)
        ctx.set_materialize_grads(False)
        return zc, out

    @staticmethod
    def backward(ctx, g_zc, g_out):
        # the one-kernel chain differentiates the depth only; fc3d.capture.reproject routes calls whose pose, xy
        # or pattern ask for a gradient to the composed operators, which provide all of them
        assert not any(ctx.needs_input_grad[1:]), "use fc3d.capture.reproject: it composes the operators for these gradients"
        if g_zc is None and g_out is None:
            return None, None, None, None, None
        zc, r, a, t, im = ctx.saved_tensors
        B, H, W, _ = zc.shape
        gz = _dense(g_zc) if g_zc is not None else None
        go = _dense(g_out) if g_out is not None else None
        gp = torch.empty_like(zc)
        with _on_device(zc.device):
            _check(_lib.fc3d_reproject_bwd(_stream(), gz.data_ptr() if gz is not None else None,
                                           go.data_ptr() if go is not None else None, zc.data_ptr(), r.data_ptr(),
                                           a.data_ptr(), t.data_ptr(), im.data_ptr(), B, H, W, im.shape[1],
                                           im.shape[2], gp.data_ptr()),
                   "fc3d_reproject_bwd")
        return gp, None, None, None, None


class ReprojLossFn(torch.autograd.Function):
    """The re-projection loss (train_pytorch.py:188-211) from the predicted depth zp_hat and an
    already eroded mask: a [4] tensor (loss_Ic_L1 + loss_zc_rms, loss_zc_rms, loss_Ic_rms,
    loss_Ic_L1).  zc and the re-projected pattern never reach memory; gradient w.r.t. zp_hat only."""

    @staticmethod
    def forward(ctx, zp_hat, mask, xy, inv, T, img, Ic, zcGT, sum_reduce=None):
        # sum_reduce: optional callable applied to the four masked sums before the losses are formed
        # (data-parallel training with the reference's global-sum / global-count losses: the sums
        # are added over ranks, every rank then evaluates and differentiates the same global loss)
        dev = _need_gpu(zp_hat, mask, xy, inv, T, img, Ic, zcGT)
        B, H, W, _ = zp_hat.shape
        one = (B, H, W, 1)
        if (tuple(zp_hat.shape) != one or tuple(mask.shape) != one or tuple(Ic.shape) != one
                or tuple(zcGT.shape) != one or tuple(xy.shape) != (B, H, W, 2) or img.dim() != 4
                or img.shape[0] != B or img.shape[3] != 1 or tuple(inv.shape) != (B, 4, 4)
                or tuple(T.shape) != (B, 4, 4)):
            raise RuntimeError(f"reproj_loss: zp_hat {tuple(zp_hat.shape)}, mask {tuple(mask.shape)}, xy "
                               f"{tuple(xy.shape)}, Ic {tuple(Ic.shape)}, zcGT {tuple(zcGT.shape)}, img "
                               f"{tuple(img.shape)}, inverse pose {tuple(inv.shape)}, transform {tuple(T.shape)}")
        ts = tuple(_dense(t.detach()) for t in (zp_hat, mask, xy, inv, T, img, Ic, zcGT))
        z, m, r, a, t, im, ic, zg = ts
        sums = torch.empty(8, dtype=torch.float32, device=dev)       # [0:4] the masked sums, [4:8] the losses
        ws = torch.empty(max(4, _lib.fc3d_reproj_loss_workspace(B, H, W) // 4), dtype=torch.float32, device=dev)
        with _on_device(dev):
            _check(_lib.fc3d_reproj_loss_fwd(_stream(), z.data_ptr(), m.data_ptr(), r.data_ptr(), a.data_ptr(),
                                             t.data_ptr(), im.data_ptr(), ic.data_ptr(), zg.data_ptr(), B, H, W,
                                             im.shape[1], im.shape[2], ws.data_ptr(), sums.data_ptr(),
                                             sums.data_ptr() + 16),
                   "fc3d_reproj_loss_fwd")
        if sum_reduce is not None:
            S = sum_reduce(sums[0:4].clone())
            sums[0:4] = S
            # the reference's fp32 expressions (utils_pytorch.py:161-163, :177; train_pytorch.py:209)
            nc = S[3].clamp(min=1e-6)
            zc_rms, ic_rms = torch.sqrt(S[0] / nc + 1e-6), torch.sqrt(S[1] / nc + 1e-6)
            l1 = S[2] / (S[3] + 1)
            sums[4:8] = torch.stack([l1 + zc_rms, zc_rms, ic_rms, l1])
        ctx.save_for_backward(sums, *ts)
        return sums[4:8].clone()

    @staticmethod
    def backward(ctx, g_losses):
        # as for ReprojectFn: fc3d.capture.reProj_IpGT composes the operators when anything but the depth asks
        assert not any(ctx.needs_input_grad[1:]), "use fc3d.capture.reProj_IpGT: it composes the operators for these gradients"
        sums, z, m, r, a, t, im, ic, zg = ctx.saved_tensors
        B, H, W, _ = z.shape
        g = _dense(g_losses.to(torch.float32))
        gz = torch.empty_like(z)
        with _on_device(z.device):
            _check(_lib.fc3d_reproj_loss_bwd(_stream(), g.data_ptr(), sums.data_ptr(), z.data_ptr(), m.data_ptr(), r.data_ptr(),
                                             a.data_ptr(), t.data_ptr(), im.data_ptr(), ic.data_ptr(), zg.data_ptr(),
                                             B, H, W, im.shape[1], im.shape[2], gz.data_ptr()),
                   "fc3d_reproj_loss_bwd")
        return gz, None, None, None, None, None, None, None, None


# --------------------------------------------------------------------------- #
# capture simulator (forward only)
# --------------------------------------------------------------------------- #
def _plane(t: torch.Tensor, B: int, H: int, W: int, what: str) -> torch.Tensor:
    if t.numel() != B * H * W:
        raise ValueError(f"{what}: expected {B}x{H}x{W} values, got shape {tuple(t.shape)}")
    return _dense(t)


def code_pattern_shaded_supported(psf: torch.Tensor) -> bool:
    """Can code_pattern_shaded fold the shading into the blur with this PSF stack?"""
    return psf.dim() == 3 and psf.shape[1] == psf.shape[2] and int(psf.shape[1]) in (3, 5, 7)


def code_pattern_shaded(I: torch.Tensor, z: torch.Tensor, grid: LayerGrid, psf: torch.Tensor | None = None) -> torch.Tensor:
    """codePattern(withReflectance(I, z), z) in one pass plus the normalisation.  psf None:
    is_conv_psf=False, the pattern path of the reference's training loop (one streaming pass);
    psf [L,K,K] with K in (3, 5, 7): is_conv_psf=True, the shading folded into the rolling-row blur.
    I [B or 1,H,W,1] (channel 0 of the pattern), z [B,H,W,1] -> [B,H,W,1]; no gradient."""
    dev = _need_gpu(I, z, psf)
    if psf is not None and (not code_pattern_shaded_supported(psf) or psf.shape[0] != grid.n_layers):
        raise ValueError(f"no shaded blur for a PSF stack of shape {tuple(psf.shape)} with {grid.n_layers} layers")
    B, H, W = z.shape[0], z.shape[1], z.shape[2]
    if I.dim() != 4 or I.shape[3] != 1 or I.shape[0] not in (1, B) or tuple(I.shape[1:3]) != (H, W):
        raise ValueError(f"pattern of shape {tuple(I.shape)} does not match depth {tuple(z.shape)}")
    I = I.detach()
    zd = _plane(z.detach(), B, H, W, "z")
    sb = I.stride(0) if I.shape[0] == B and B > 1 else 0
    out = torch.empty((B, H, W, 1), dtype=torch.float32, device=dev)
    keys = torch.empty(_lib.fc3d_code_pattern_fwd_workspace(B, H, W) // 8, dtype=torch.int64, device=dev)
    mx = torch.empty(B, dtype=torch.float32, device=dev)
    am = torch.empty(B, dtype=torch.int32, device=dev)
    nkeys = ctypes.c_int(0)
    with _on_device(dev):
        s = _stream()
        if psf is None:
            _check(_lib.fc3d_code_pattern_shaded_fwd(s, I.data_ptr(), sb, I.stride(1), I.stride(2), zd.data_ptr(), B, H, W,
                                                     grid, out.data_ptr(), None, keys.data_ptr(), ctypes.byref(nkeys)),
                   "fc3d_code_pattern_shaded_fwd")
        else:
            pd = _dense(psf)
            _check(_lib.fc3d_code_pattern_shaded_conv_fwd(s, I.data_ptr(), sb, I.stride(1), I.stride(2), zd.data_ptr(),
                                                          pd.data_ptr(), B, H, W, int(pd.shape[1]), int(pd.shape[2]),
                                                          grid, out.data_ptr(), None, keys.data_ptr(),
                                                          ctypes.byref(nkeys)),
                   "fc3d_code_pattern_shaded_conv_fwd")
        _check(_lib.fc3d_code_pattern_norm(s, out.data_ptr(), keys.data_ptr(), nkeys.value, B, H, W, out.data_ptr(),
                                           mx.data_ptr(), am.data_ptr()), "fc3d_code_pattern_norm")
    return out


def reflectance(I: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
    """I [B or 1,H,W,1] * (0.8 cos(atan(|grad z| / 100um)) + 0.2), z [B,H,W,1]; no gradient."""
    dev = _need_gpu(I, z)
    B, H, W = z.shape[0], z.shape[1], z.shape[2]
    if I.shape[0] not in (1, B) or I.numel() != I.shape[0] * H * W:
        raise ValueError(f"pattern of shape {tuple(I.shape)} does not match depth {tuple(z.shape)}")
    I, z = _dense(I.detach()), _plane(z.detach(), B, H, W, "z")
    out = torch.empty((B, H, W, 1), dtype=torch.float32, device=dev)
    with _on_device(dev):
        _check(_lib.fc3d_reflectance_fwd(_stream(), I.data_ptr(), H * W if I.shape[0] == B and B > 1 else 0,
                                         z.data_ptr(), B, H, W, out.data_ptr()), "fc3d_reflectance_fwd")
    return out


def capture_warp(Ip, z_p, z_c, ray_x, ray_y, T_p2c, T_c2p, zero_mask=None, want_grids=False):
    """Ic_masked [B,H,W,1], xyz_cview [B,H,W,3], mask u8 [B,H,W] (and the two grids if asked)
    from the coded pattern and the two depth maps; no gradient."""
    dev = _need_gpu(Ip, z_p, z_c, ray_x, ray_y, T_p2c, T_c2p)
    B, H, W = z_p.shape[0], z_p.shape[1], z_p.shape[2]
    if ray_x.numel() != W or ray_y.numel() != H or T_p2c.numel() != 16 * B or T_c2p.numel() != 16 * B:
        raise ValueError("ray table or transforms do not match the depth maps")
    Ip, z_p, z_c = (_plane(t.detach(), B, H, W, n) for t, n in ((Ip, "Ip"), (z_p, "z_p"), (z_c, "z_c")))
    Ic = torch.empty((B, H, W, 1), dtype=torch.float32, device=dev)
    xyz = torch.empty((B, H, W, 3), dtype=torch.float32, device=dev)
    mask = zero_mask if zero_mask is not None else torch.empty((B, H, W), dtype=torch.uint8, device=dev)
    if mask.dtype != torch.uint8 or mask.numel() != B * H * W or not mask.is_contiguous():
        raise ValueError("zero_mask must be a dense uint8 [B,H,W] tensor")
    gp = torch.empty((B, H, W, 2), dtype=torch.float32, device=dev) if want_grids else None
    gc = torch.empty((B, H, W, 2), dtype=torch.float32, device=dev) if want_grids else None
    with _on_device(dev):
        _check(_lib.fc3d_capture_warp_fwd(_stream(), Ip.data_ptr(), z_p.data_ptr(), z_c.data_ptr(),
                                          _dense(ray_x).data_ptr(), _dense(ray_y).data_ptr(),
                                          _dense(T_p2c).data_ptr(), _dense(T_c2p).data_ptr(), B, H, W, Ic.data_ptr(),
                                          xyz.data_ptr(), mask.data_ptr(), int(zero_mask is not None),
                                          gp.data_ptr() if want_grids else None,
                                          gc.data_ptr() if want_grids else None), "fc3d_capture_warp_fwd")
    return (Ic, xyz, mask, gp, gc) if want_grids else (Ic, xyz, mask)


def capture_post(Ic, offset, gauss, scale, mask_u8):
    """(Ic_scaled, Ic_mask) [B,H,W,1] of the sensor model from the warped image, the three noise
    draws (offset, scale: B values; gauss: B*H*W values) and the visibility mask; no gradient."""
    dev = _need_gpu(Ic, offset, gauss, scale)
    B, H, W = Ic.shape[0], Ic.shape[1], Ic.shape[2]
    if offset.numel() != B or scale.numel() != B:
        raise ValueError("offset and scale hold one value per sample")
    if mask_u8.dtype != torch.uint8 or mask_u8.numel() != B * H * W or mask_u8.device != dev:
        raise ValueError("mask_u8 must be the warp's uint8 [B,H,W] visibility mask")
    Ic, gauss = _plane(Ic.detach(), B, H, W, "Ic"), _plane(gauss, B, H, W, "gauss")
    scaled = torch.empty((B, H, W, 1), dtype=torch.float32, device=dev)
    valid = torch.empty((B, H, W, 1), dtype=torch.float32, device=dev)
    ws = torch.empty(max(1, _lib.fc3d_capture_post_workspace(B, H, W) // 4), dtype=torch.float32, device=dev)
    with _on_device(dev):
        _check(_lib.fc3d_capture_post(_stream(), Ic.data_ptr(), _dense(offset).data_ptr(), gauss.data_ptr(),
                                      _dense(scale).data_ptr(), _dense(mask_u8).data_ptr(), B, H, W,
                                      scaled.data_ptr(), valid.data_ptr(), ws.data_ptr()), "fc3d_capture_post")
    return scaled, valid


def capture_fwd(Ip, z_p, z_c, ray_x, ray_y, pose, K, offset, gauss, scale, want_grids=False):
    """capture_warp + capture_post from the pose and the intrinsic matrix in three launches (forward warp
    that forms K*pose itself, inverse warp that also finds the sensor model's peak, sensor model):
    (Ic_masked, xyz_cview, Ic_scaled, Ic_mask[, grid_p2c, grid_c2p]), bit-identical to the two-step form;
    no gradient.  The visibility mask is a buffer that the sensor kernel hands back cleared."""
    dev = _need_gpu(Ip, z_p, z_c, ray_x, ray_y, pose, K, offset, gauss, scale)
    B, H, W = z_p.shape[0], z_p.shape[1], z_p.shape[2]
    if ray_x.numel() != W or ray_y.numel() != H or tuple(pose.shape) != (B, 4, 4):
        raise ValueError("ray table or pose do not match the depth maps")
    if offset.numel() != B or scale.numel() != B:
        raise ValueError("offset and scale hold one value per sample")
    Ip, z_p, z_c, gauss = (_plane(t.detach(), B, H, W, n) for t, n in ((Ip, "Ip"), (z_p, "z_p"), (z_c, "z_c"),
                                                                       (gauss, "gauss")))
    K, k_sb = _intrinsics_args(K, B)
    Ic = torch.empty((B, H, W, 1), dtype=torch.float32, device=dev)
    xyz = torch.empty((B, H, W, 3), dtype=torch.float32, device=dev)
    scaled = torch.empty((B, H, W, 1), dtype=torch.float32, device=dev)
    valid = torch.empty((B, H, W, 1), dtype=torch.float32, device=dev)
    gp = torch.empty((B, H, W, 2), dtype=torch.float32, device=dev) if want_grids else None
    gc = torch.empty((B, H, W, 2), dtype=torch.float32, device=dev) if want_grids else None
    ws = torch.empty(_lib.fc3d_capture_fwd_workspace(B, H, W) // 4, dtype=torch.float32, device=dev)
    with _on_device(dev):
        s = _stream()
        key, mask = _pooled_zero_mask(B, H, W, dev, s, torch.cuda.is_current_stream_capturing())
        pooled = mask is not None
        if not pooled:
            mask = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
        status = _lib.fc3d_capture_fwd(s, Ip.data_ptr(), z_p.data_ptr(), z_c.data_ptr(), _dense(ray_x).data_ptr(),
                                       _dense(ray_y).data_ptr(), _dense(pose.detach()).data_ptr(), K.data_ptr(), k_sb,
                                       _dense(offset).data_ptr(), gauss.data_ptr(), _dense(scale).data_ptr(), B, H, W,
                                       Ic.data_ptr(), xyz.data_ptr(), scaled.data_ptr(), valid.data_ptr(),
                                       mask.data_ptr(), int(pooled), int(pooled), ws.data_ptr(),
                                       gp.data_ptr() if want_grids else None, gc.data_ptr() if want_grids else None)
        if status != 0 and pooled:
            _zero_masks.pop(key, None)
        _check(status, "fc3d_capture_fwd")
    return (Ic, xyz, scaled, valid, gp, gc) if want_grids else (Ic, xyz, scaled, valid)


def box_filter(img: torch.Tensor, k: int, mode: str = "mean", c: float = 0.0) -> torch.Tensor:
    """k x k zero-padded box mean of a one-channel [B,H,W,1] image ("mean"), the image divided by
    (mean + c) ("local_norm"), or the erosion mean > 0.999 of a 0/1 mask ("erode"); no gradient."""
    dev = _need_gpu(img)
    if img.dim() != 4 or img.shape[3] != 1:
        raise ValueError(f"expected a [B,H,W,1] image, got {tuple(img.shape)}")
    m = {"mean": 0, "local_norm": 1, "erode": 2}[mode]
    B, H, W = img.shape[0], img.shape[1], img.shape[2]
    src = _dense(img.detach())
    out = torch.empty((B, H, W, 1), dtype=torch.float32, device=dev)
    with _on_device(dev):
        _check(_lib.fc3d_box_filter(_stream(), src.data_ptr(), B, H, W, int(k), m, float(c), out.data_ptr()),
               "fc3d_box_filter")
    return out
