"""Seeded synthetic capture scenes (host side, numpy float64 -> fp32).

The reference trains on a Blender-rendered dataset that is not shipped
(data_provider_pytorch.py:12-37 reads z_p, z_c [N,800,1200,1] and pose_p2c
[N,4,4] from .npz files).  This module produces tensors of the same meaning
for tests and benchmarks: an analytic scene (tilted background plane, a
foreground sphere and a fronto-parallel foreground disc) is ray-cast from the
projector frame and from the camera frame, so z_p, z_c and pose_p2c are
mutually consistent and the disc/sphere produce true occlusions.

Everything is computed in float64 from numpy.random.RandomState(seed) and cast
to fp32 at the end, so the tensors are identical on every machine.
"""
from __future__ import annotations

import numpy as np

FOCAL = 1666.67          # geometry_pytorch.py:25-26, 247
Z_MIN = 0.695714566      # optics_pytorch.py:16-17
Z_MAX = 0.941062289


def ray_table(width: int, height: int) -> np.ndarray:
    """float64 [H,W,3] pinhole rays ((x+.5-W/2)/f, (y+.5-H/2)/f, 1); the closed
    form of geometry_pytorch.py:15-41 evaluated on index grids."""
    x = np.arange(width, dtype=np.float64)
    y = np.arange(height, dtype=np.float64)
    rx = width / (2 * FOCAL) * ((2.0 * ((x + 0.5) / width)) - 1.0)
    ry = height / (2 * FOCAL) * ((2.0 * ((y + 0.5) / height)) - 1.0)
    t = np.empty((height, width, 3), dtype=np.float64)
    t[..., 0] = rx[None, :]
    t[..., 1] = ry[:, None]
    t[..., 2] = 1.0
    return t


def rodrigues(axis: np.ndarray, angle: float) -> np.ndarray:
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


def random_pose(rs: np.random.RandomState, theta_deg: float, t_max) -> np.ndarray:
    """4x4 projector->camera rigid transform: axis-angle with angle U(-theta,theta),
    translation U(-t_max, t_max) per axis."""
    axis = rs.randn(3)
    ang = np.deg2rad(rs.uniform(-theta_deg, theta_deg))
    T = np.eye(4)
    T[:3, :3] = rodrigues(axis, ang)
    T[:3, 3] = rs.uniform(-1.0, 1.0, 3) * np.asarray(t_max, dtype=np.float64)
    return T


def _cast_depth(origin, dirs, plane, sphere, disc):
    """Depth parameter s (point = origin + s*dir) of the nearest surface hit."""
    n, d0 = plane
    denom = dirs @ n
    s = (d0 - origin @ n) / denom
    s = np.where(s > 0, s, np.inf)
    if sphere is not None:
        c, r = sphere
        oc = origin - c
        a = np.sum(dirs * dirs, axis=-1)
        b = dirs @ oc
        cc = oc @ oc - r * r
        disc_ = b * b - a * cc
        hit = disc_ > 0
        root = (-b - np.sqrt(np.where(hit, disc_, 0.0))) / a
        s = np.where(hit & (root > 0) & (root < s), root, s)
    if disc is not None:
        c, r = disc           # fronto-parallel (normal = +z of the projector frame)
        sd = (c[2] - origin[2]) / dirs[..., 2]
        px = origin[0] + sd * dirs[..., 0] - c[0]
        py = origin[1] + sd * dirs[..., 1] - c[1]
        hit = (px * px + py * py <= r * r) & (sd > 0) & (sd < s)
        s = np.where(hit, sd, s)
    return s


def make_scene(B: int, H: int, W: int, seed: int, theta_deg: float = 3.0,
               t_max=(0.03, 0.03, 0.01), pose_scale: float | None = None):
    """Returns dict(z_p [B,H,W,1] f32, z_c [B,H,W,1] f32, pose_p2c [B,4,4] f32).

    Scene sizes scale with the field of view so small test images still contain
    the occluders.  pose_scale (default W/512, capped at 1) shrinks the motion
    for small images so that the two views keep overlapping."""
    rs = np.random.RandomState(seed)
    rays = ray_table(W, H)
    ext = 0.8 * 0.5 * min(H, W) / FOCAL           # half extent of the view at z = 0.8
    if pose_scale is None:
        pose_scale = min(1.0, W / 512.0)
    z_p = np.empty((B, H, W, 1), np.float32)
    z_c = np.empty((B, H, W, 1), np.float32)
    pose = np.empty((B, 4, 4), np.float32)
    lo, hi = Z_MIN + 1e-3, Z_MAX - 1e-3
    for b in range(B):
        zc0 = rs.uniform(0.84, 0.88)
        relief = rs.uniform(0.03, 0.055)
        sx, sy = rs.uniform(-1, 1, 2) * relief / ext
        plane = (np.array([-sx, -sy, 1.0]), zc0)
        sphere = (np.array([rs.uniform(-0.5, 0.5) * ext, rs.uniform(-0.5, 0.5) * ext, rs.uniform(0.77, 0.80)]),
                  rs.uniform(0.25, 0.4) * ext)
        disc = (np.array([rs.uniform(-0.6, 0.6) * ext, rs.uniform(-0.6, 0.6) * ext, rs.uniform(0.72, 0.75)]),
                rs.uniform(0.15, 0.3) * ext)
        T = random_pose(rs, theta_deg * pose_scale, np.asarray(t_max) * pose_scale)
        R, t = T[:3, :3], T[:3, 3]
        zp = _cast_depth(np.zeros(3), rays, plane, sphere, disc)
        origin_c = -R.T @ t                       # camera centre in the projector frame
        dirs_c = rays @ R                         # R^T applied to each camera ray
        zc = _cast_depth(origin_c, dirs_c, plane, sphere, disc)
        z_p[b, ..., 0] = np.clip(zp, lo, hi)
        z_c[b, ..., 0] = np.clip(zc, lo, hi)
        pose[b] = T
    return {"z_p": z_p, "z_c": z_c, "pose_p2c": pose}


def dot_pattern(H: int, W: int, n_rows: int | None = None, n_cols: int | None = None) -> np.ndarray:
    """Binary dot-array image [H,W] f32.  Same drawing rule as the 'dotArray'
    mode of utils_pytorch.py:85-119 (dot centres on an even grid, filled discs
    of diameter ~ spacing/3), with the grid scaled to the image (default one dot
    per 8x8 pixels, which is the reference's 100x150 grid on 800x1200)."""
    n_rows = max(1, H // 8) if n_rows is None else n_rows
    n_cols = max(1, W // 8) if n_cols is None else n_cols
    sp_h, sp_w = H / n_rows, W / n_cols
    dot = max(2, int(min(sp_h, sp_w) / 3))
    lo, hi = -dot // 2, dot // 2                    # python floor division, as the reference
    ci = ((np.arange(n_rows) + 0.5) * sp_h).astype(np.int64)
    cj = ((np.arange(n_cols) + 0.5) * sp_w).astype(np.int64)
    img = np.zeros((H, W), np.float32)
    for di in range(lo, hi + 1):
        for dj in range(lo, hi + 1):
            if di * di + dj * dj <= (dot // 2) ** 2:
                ii = ci + di
                jj = cj + dj
                ii = ii[(ii >= 0) & (ii < H)]
                jj = jj[(jj >= 0) & (jj < W)]
                img[np.ix_(ii, jj)] = 1.0
    return img


def reflectance(z: np.ndarray, pixelsize: float = 100e-6) -> np.ndarray:
    """0.8*cos(atan(|grad z|/pixelsize)) + 0.2 with forward differences padded by
    zero at the last row/column; utils_pytorch.py:206-222, 251-280.  z [B,H,W,1]."""
    z = z.astype(np.float32)
    dy = np.zeros_like(z)
    dx = np.zeros_like(z)
    dy[:, :-1] = z[:, 1:] - z[:, :-1]
    dx[:, :, :-1] = z[:, :, 1:] - z[:, :, :-1]
    mag = np.sqrt(dy * dy + dx * dx)
    return (np.cos(np.arctan(mag / np.float32(pixelsize))) * np.float32(0.8) + np.float32(0.2)).astype(np.float32)


def make_pattern(z_p: np.ndarray, seed: int, kind: str = "dots") -> np.ndarray:
    """Projected pattern plane [B,H,W,1] f32: dot array x reflectance ('dots'),
    or U(0,1) noise ('uniform', no ties in the per-image maximum)."""
    B, H, W, _ = z_p.shape
    if kind == "dots":
        return (dot_pattern(H, W)[None, :, :, None] * reflectance(z_p)).astype(np.float32)
    if kind == "uniform":
        return np.random.RandomState(seed).uniform(0.0, 1.0, (B, H, W, 1)).astype(np.float32)
    raise ValueError(f"unknown pattern kind {kind!r}")


def make_psfs(L: int, K: int, kind: str = "disc") -> np.ndarray:
    """PSF stack [L,K,K,1] f32 in the layout of optics_pytorch.py:22-27.
    'ones' is the reference's fallback stack; 'disc'/'gauss' grow with the
    distance of the layer from the in-focus layer (L-1)/2 and sum to one;
    'rand' is RandomState(0).rand for asymmetric-kernel tests."""
    if kind == "ones":
        return np.ones((L, K, K, 1), np.float32)
    if kind == "rand":
        return np.random.RandomState(0).rand(L, K, K, 1).astype(np.float32)
    c = (K - 1) / 2.0
    yy, xx = np.meshgrid(np.arange(K) - c, np.arange(K) - c, indexing="ij")
    rr = np.sqrt(xx * xx + yy * yy)
    out = np.empty((L, K, K, 1), np.float64)
    mid = (L - 1) / 2.0
    for l in range(L):
        rad = 0.6 + (K / 2.0 - 0.6) * abs(l - mid) / max(mid, 1.0)
        if kind == "disc":
            k = np.clip(rad + 0.5 - rr, 0.0, 1.0)
        elif kind == "gauss":
            k = np.exp(-0.5 * (rr / (0.5 * rad + 0.3)) ** 2)
        else:
            raise ValueError(f"unknown psf kind {kind!r}")
        out[l, :, :, 0] = k / k.sum()
    return out.astype(np.float32)


# Named workloads used by the benchmark and the full-size property tests.
CONFIGS = {
    "C1": dict(B=1, H=256, W=256, L=21, K=5, theta_deg=3.0, t_max=(0.03, 0.03, 0.01), seed=1001, pose_scale=1.0),
    "C2": dict(B=32, H=512, W=512, L=21, K=5, theta_deg=3.0, t_max=(0.03, 0.03, 0.01), seed=1002, pose_scale=1.0),
    "C4": dict(B=4, H=1024, W=1024, L=41, K=23, theta_deg=3.0, t_max=(0.03, 0.03, 0.01), seed=1004, pose_scale=1.0),
    "C5": dict(B=32, H=512, W=512, L=21, K=5, theta_deg=20.0, t_max=(0.15, 0.15, 0.05), seed=1005, pose_scale=1.0),
}


def make_workload(name: str, B: int | None = None, K: int | None = None, psf_kind: str = "disc",
                  pattern_kind: str = "dots", seed_offset: int = 0):
    """All host tensors for one named configuration (numpy fp32)."""
    cfg = dict(CONFIGS[name])
    if B is not None:
        cfg["B"] = B
    if K is not None:
        cfg["K"] = K
    seed = cfg["seed"] + seed_offset
    sc = make_scene(cfg["B"], cfg["H"], cfg["W"], seed, cfg["theta_deg"], cfg["t_max"], cfg["pose_scale"])
    pat = make_pattern(sc["z_p"], seed, pattern_kind)
    psfs = make_psfs(cfg["L"], cfg["K"], psf_kind)
    g = np.random.RandomState(seed + 7).uniform(0.0, 1.0, sc["z_p"].shape).astype(np.float32)
    return dict(cfg=cfg, pattern=pat, psfs=psfs, upstream_grad=g, **sc)
