#!/usr/bin/env python3
"""Re-projection chain of reProj_IpGT (train_pytorch.py:194-202), forward + backward to zp_cView:
the fused kernels against the same chain composed from the drop-in operators.

    python bench_reproj.py [--steps K] [--warmup W] [--batch B] [--size N]

Secondary measurement (the headline one is bench.py).  Algorithmic bytes per pixel: forward
zp 4 + xy 8 + IpGT 4 -> zc 4 + Ic 4; backward dL/dzc 4 + dL/dIc 4 + zc 4 + xy 8 + IpGT 4 -> dL/dzp 4;
52 B/px in total.  Prints one json line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "depth-estimation_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

ALG_BYTES = 52
HBM_PEAK = 8.0e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=512)
    a = ap.parse_args()
    if not torch.cuda.is_available():
        sys.exit("bench_reproj.py needs a GPU (the HIP path has no CPU fallback)")
    import geometry_pytorch as G
    from fc3d import capture, profiler, synth
    dev = torch.device("cuda", 0)
    B, H, W = a.batch, a.size, a.size
    sc = synth.make_scene(B, H, W, 4242)
    K = np.diag([1666.67, 1666.67, 1, 1])
    K[0][2], K[1][2] = W / 2, H / 2
    Kt = torch.from_numpy(K).float().to(dev).unsqueeze(0).repeat(B, 1, 1)
    G.get_intrinsic = lambda n, device="cuda": Kt
    d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
    rays = G.pixel_to_ray_array(W, H)
    xy = d(rays[:, :, 0:2].astype(np.float32)).unsqueeze(0).repeat(B, 1, 1, 1)
    rs = np.random.RandomState(7)
    # projector depth seen from the camera: the camera depth of the scene stands in for it (same range)
    zp = d(sc["z_c"]).requires_grad_(True)
    pose = d(sc["pose_p2c"])
    img = d(rs.uniform(0, 1, (B, H, W, 1)).astype(np.float32))
    g_zc = d(rs.uniform(-1, 1, (B, H, W, 1)).astype(np.float32))
    g_ic = d(rs.uniform(-1, 1, (B, H, W, 1)).astype(np.float32))

    def step(fused):
        zp.grad = None
        zc, Ic = capture.reproject(zp, pose, xy, img, fused=fused)
        torch.autograd.backward((zc, Ic), (g_zc, g_ic))

    out = {}
    for name, fused in (("fused", True), ("composed", False)):
        for _ in range(a.warmup):
            step(fused)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step(fused)
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) / a.steps
    # the whole loss (erosion, chain, masked sums, backward to the predicted depth): loss kernels
    # against the chain kernels + tensor reductions, and against the fully composed form
    Ic_cap = d(rs.uniform(0, 1, (B, H, W, 1)).astype(np.float32))
    zc_gt = d(sc["z_c"])
    mask = d((rs.uniform(0, 1, (B, H, W, 1)) > 0.02).astype(np.float32))

    def loss_step(mode):
        zp.grad = None
        if mode == "kernels":
            vals = capture.reProj_IpGT(zp, pose, xy, Ic_cap, zc_gt, mask, img, fused=True)
        else:
            me = capture.erode(mask, 5)
            zc, Ir = capture.reproject(zp * me, pose, xy, img, fused=(mode == "chain"))
            vals = (capture.cost_L1_mask(Ic_cap, Ir, me) + capture.cost_rms_mask(zc_gt, zc, me),)
        vals[0].backward()

    loss_t = {}
    for mode in ("kernels", "chain", "composed"):
        for _ in range(a.warmup):
            loss_step(mode)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            loss_step(mode)
        torch.cuda.synchronize()
        loss_t[mode] = (time.perf_counter() - t0) / a.steps
    profiler.enable(True)
    profiler.report()
    for _ in range(a.steps):
        step(True)
    torch.cuda.synchronize()
    prof = profiler.report()
    profiler.enable(False)
    px = B * H * W
    kern = {k: round(1e3 * ms / n, 2) for k, (n, ms) in prof.items()}
    kus = sum(1e3 * ms / n for k, (n, ms) in prof.items() if k.startswith("reproject"))
    print(json.dumps({
        "metric": "re-projections/sec fwd+bwd (secondary; reProj_IpGT chain)", "value": B / out["fused"],
        "unit": "images/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * out["fused"],
        "higher_is_better": True, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{B}x{H}x{W}: zp_cView -> zc -> scene points -> projector pixel -> sampled pattern, "
                               "fwd + bwd to zp_cView"},
        "composed_ms_per_step": 1e3 * out["composed"], "speedup_vs_composed": out["composed"] / out["fused"],
        "roofline": {"bound": "hbm", "kernel": "reproject_fwd + reproject_bwd", "achieved": ALG_BYTES * px / (kus * 1e-6) / 1e9,
                     "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": ALG_BYTES * px / (kus * 1e-6) / HBM_PEAK,
                     "traffic": None, "kernel_us": kus},
        "kernels_us": kern,
        "loss_ms_per_step": {"loss_kernels": 1e3 * loss_t["kernels"], "chain_kernels_plus_tensor_reductions": 1e3 * loss_t["chain"],
                             "composed": 1e3 * loss_t["composed"]}}), flush=True)


if __name__ == "__main__":
    main()
