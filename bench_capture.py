#!/usr/bin/env python3
"""The capture simulator, forward_model (train_pytorch.py:79-162), forward only as the reference
runs it: the capture kernels against the same function composed from the drop-in operators and
tensor arithmetic.

    python bench_capture.py [--steps K] [--warmup W] [--batch B] [--size N]

Secondary measurement (the headline one is bench.py).  Algorithmic bytes per pixel: depth maps
4 + 4 and the noise plane 4 in; Ip_coded 4, Ic 4, Ic_scaled 4, Ic_mask 4 and xyz_cView 12 out;
40 B/px in total (the dot pattern is shared by the batch).  Prints one json line.
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

ALG_BYTES = 40
HBM_PEAK = 8.0e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--no-conv", action="store_true",
                    help="is_conv_psf=False, the mode the reference's training loop runs (train_pytorch.py:51-52)")
    a = ap.parse_args()
    conv = not a.no_conv
    if not torch.cuda.is_available():
        sys.exit("bench_capture.py needs a GPU (the HIP path has no CPU fallback)")
    import geometry_pytorch as G
    import optics_pytorch as O
    from fc3d import capture, profiler, synth
    dev = torch.device("cuda", 0)
    B, H, W, L = a.batch, a.size, a.size, 21
    sc = synth.make_scene(B, H, W, 4242)
    K = np.diag([1666.67, 1666.67, 1, 1])
    K[0][2], K[1][2] = W / 2, H / 2
    Kt = torch.from_numpy(K).float().to(dev).unsqueeze(0).repeat(B, 1, 1)
    G.get_intrinsic = lambda n, device="cuda": Kt
    O.PSFs_np = synth.make_psfs(L, 5, "disc")
    d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
    z_p, z_c, pose = d(sc["z_p"]), d(sc["z_c"]), d(sc["pose_p2c"])
    noise = capture._draw_noise(B, H, W, dev)

    def step(fused):
        with torch.no_grad():
            return capture.forward_model(z_p, z_c, pose, L, conv, None, noise, fused=fused)

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
    from fc3d.graphs import GraphedCapture
    gcap = GraphedCapture(z_p, z_c, pose, noise, L, conv, None)
    for _ in range(a.warmup):
        gcap.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        gcap.replay()
    torch.cuda.synchronize()
    out["graph"] = (time.perf_counter() - t0) / a.steps
    profiler.enable(True)
    profiler.report()
    for _ in range(a.steps):
        step(True)
    torch.cuda.synchronize()
    prof = profiler.report()
    profiler.enable(False)
    # the box filters that follow the capture in the training step, against avg_pool2d
    import torch.nn.functional as F
    img, msk = step(True)[0], step(True)[1]

    def timed(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return 1e6 * (time.perf_counter() - t0) / n

    pool = lambda t, k: F.avg_pool2d(t.permute(0, 3, 1, 2), k, stride=1, padding=k // 2).permute(0, 2, 3, 1)  # noqa: E731
    box = {"local_norm17_us": timed(lambda: capture.local_norm(img, 17, 0.01)),
           "local_norm17_torch_us": timed(lambda: img / (pool(img, 17) + 0.01)),
           "erode5_us": timed(lambda: capture.erode(msk, 5)),
           "erode5_torch_us": timed(lambda: (pool(msk, 5) > 0.999).float())}
    px = B * H * W
    kern = {k: round(1e3 * ms / n, 2) for k, (n, ms) in prof.items()}
    kus = sum(1e3 * ms / n for k, (n, ms) in prof.items())
    print(json.dumps({
        "metric": "simulated captures/sec, forward only (secondary; whole forward_model)", "value": B / out["fused"],
        "unit": "captures/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * out["fused"],
        "higher_is_better": True, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{B}x{H}x{W}, L={L}, " + ("5x5 PSFs" if conv else "no blur (is_conv_psf=False)") +
                               ": depth maps -> shaded pattern -> codePattern -> warp -> sensor model, forward only"},
        "graph_replay_ms_per_step": 1e3 * out["graph"], "composed_ms_per_step": 1e3 * out["composed"], "speedup_vs_composed": out["composed"] / out["fused"],
        "roofline": {"bound": "hbm", "kernel": "all kernels of one capture", "achieved": ALG_BYTES * px / (kus * 1e-6) / 1e9,
                     "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": ALG_BYTES * px / (kus * 1e-6) / HBM_PEAK,
                     "traffic": None, "kernel_us": kus},
        "kernels_us": kern, "box_filters": {k: round(v, 1) for k, v in box.items()}}), flush=True)


if __name__ == "__main__":
    main()
