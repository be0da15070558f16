#!/usr/bin/env python3
"""Simulated captures/s, forward + backward, of the FreeCam3D image-formation path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C2] [--psf-k 5]

One step = one batch of synthetic scenes through
    codePattern(pattern, z_p)  ->  warp_p2c(Ip_coded, coord_p, coord_c, pose_p2c)
forward, then backward from a fixed dL/dIc to dL/dpattern[...,0], called through the
drop-in modules (optics_pytorch / geometry_pytorch) exactly as the reference's
forward_model calls them (train_pytorch.py:94-97).  All inputs are resident in HBM before
the timed region.  The batch is sharded over ranks (weak scaling: --gpus N runs N times the
per-GPU batch, no collective on the data path), launched one process per GPU by
torch.distributed.run.

Prints ONE json line on rank 0 with the whole-job rate, the roofline fraction of the
dominant kernel (timed with HIP events on the launching stream) and, at N=1, the CPU
oracle's rate on the host cores for comparison.
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

HBM_PEAK = 8.0e12   # B/s, MI355X HBM3E nominal

# Algorithmic bytes per pixel charged to each kernel of the path: every consumed input element
# read once, every returned tensor / gradient written once (fp32).  They add up to the 100 B/px
# of one capture, forward + backward.
ALG_BYTES = {
    "code_pattern_fwd": 12,           # pattern[...,0] 4 + zMap 4 -> Ip_coded 4
    "code_pattern_norm": 0,           # second pass over Ip_coded: pure overhead
    "pose_transforms": 0,
    "warp_project_splat": 24,         # coord_p 16 -> grid_p2c 8 (mask is internal)
    "warp_sample": 32,                # Ip 4 + coord_c 16 -> Ic 4 + grid_c2p 8
    "bilinear_bwd_img_gather": 16,    # dL/dIc 4 + grid_c2p 8 -> dL/dIp 4 (--bwd ordered: source tiles gather their samples)
    "bilinear_bwd_img_scatter": 16,   # the same bytes with --bwd atomic (window sums added with float atomics) ...
    "zero_fill": 0,                   # ... into a gradient image that has to be cleared first
    "bilinear_tap_index": 0,          # only when the sampler runs outside warp_p2c (which writes the index itself)
    "code_pattern_bwd_reduce": 0,     # only for PSF sizes without a rolling-row kernel
    "code_pattern_bwd": 16,           # dL/dIp 4 + layer(z) 4 + Ip_coded 4 -> dL/dpattern0 4
    "code_pattern_bwd_max": 0,        # footprint of the per-image maximum: K*K pixels per image
}
assert sum(v for k, v in ALG_BYTES.items() if k != "bilinear_bwd_img_scatter") == 100
FP32_VECTOR_PEAK = 157.3e12          # flop/s, MI355X fp32 vector (packed FMA) peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="C2", help="C1, C2 (headline), C4 or C5")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch override")
    ap.add_argument("--psf-k", type=int, default=None)
    ap.add_argument("--psf-kind", default="disc")
    ap.add_argument("--launch", choices=("graph", "eager"), default="graph",
                    help="graph: the step is captured once into a HIP graph and replayed (same kernels, order and "
                         "buffers); eager: every launch issued from Python (about 125 us of host time per step; "
                         "at the headline size that hides behind the kernels on a quiet host and is ~2%% faster "
                         "than replay, but one run in three measured 15%% slower, so replay is the default)")
    ap.add_argument("--bwd", choices=("ordered", "atomic"), default="ordered",
                    help="gradient of the warp w.r.t. the pattern: ordered = gather form without atomics (fixed "
                         "summation order), atomic = scatter form with global float atomics")
    ap.add_argument("--overlap", action="store_true",
                    help="graph launch only: put warp_p2c's projection + splat launch beside codePattern on a second "
                         "branch of the graph instead of behind it (measured 2-4 us per step slower: the fork and "
                         "join cost more than the overlap of the two kernels gives)")
    ap.add_argument("--unroll", type=int, default=0,
                    help="graph launch: consecutive steps held by the graph that is replayed in the timed region "
                         "(0 = automatic, see main)")
    ap.add_argument("--settle", type=int, default=100,
                    help="untimed steps issued right before the W warm-up steps, through the same launch path as the timed "
                         "ones and without a pause in between: the per-kernel event pass and the copy yardstick that run "
                         "before them leave gaps of milliseconds in which the GPU's clocks fall back, and W = 5 steps "
                         "(1 ms) do not bring them up again")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--n1-value", type=float, default=None,
                    help="captures/s of a previous --gpus 1 run: the line then carries scaling_efficiency_vs_n1")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="run all N ranks on GPU 0, with gloo for the barrier and the timing reductions (RCCL refuses two "
                         "ranks on one device): the whole N-rank path of this benchmark -- child ranks, rendezvous, real "
                         "kernels, per-rank rates, one line from rank 0 -- on a box with a single GPU.  The value is not "
                         "a scaling measurement (the ranks share the GPU) and the line says so")
    ap.add_argument("--launch-selftest", action="store_true",
                    help="exercise the launch / rendezvous / max-over-ranks / reporting path with gloo ranks and a host "
                         "stand-in for the step: touches no GPU (for machines without one and for CPU tests)")
    return ap.parse_args()


def build_inputs(args, rank, device):
    from fc3d import synth
    import geometry_pytorch as G
    import optics_pytorch as O

    wl = synth.make_workload(args.workload, B=args.batch, K=args.psf_k, psf_kind=args.psf_kind,
                             seed_offset=100 * rank)
    cfg = wl["cfg"]
    H, W = cfg["H"], cfg["W"]
    d = lambda a: torch.from_numpy(a).to(device)  # noqa: E731
    z_p, z_c, pose = d(wl["z_p"]), d(wl["z_c"]), d(wl["pose_p2c"])
    rays = G.pixel_to_ray_array(W, H)
    coord_p = G.points_in_camera_coords(z_p, rays).permute(0, 3, 1, 2)
    coord_c = G.points_in_camera_coords(z_c, rays).permute(0, 3, 1, 2)
    O.PSFs_np = wl["psfs"]
    # principal point at the image centre (the reference hard-codes that of its 1200x800 renders)
    K = np.diag([1666.67, 1666.67, 1, 1])
    K[0][2], K[1][2] = W / 2, H / 2
    Kt = torch.from_numpy(K).float().to(device)
    Kb = {}

    def centred_intrinsic(B, device="cuda"):
        if B not in Kb:
            Kb[B] = Kt.unsqueeze(0).repeat(B, 1, 1)
        return Kb[B]
    G.get_intrinsic = centred_intrinsic
    pat = d(wl["pattern"]).requires_grad_(True)
    g = d(wl["upstream_grad"])
    return wl, dict(pat=pat, z_p=z_p, coord_p=coord_p, coord_c=coord_c, pose=pose, g=g, L=cfg["L"])


def make_step(t):
    import geometry_pytorch as G
    import optics_pytorch as O

    def step():
        t["pat"].grad = None
        Ip = O.codePattern(t["pat"], t["z_p"], n_layers=t["L"])
        Ic, _gp2c, _gc2p = G.warp_p2c(Ip, t["coord_p"], t["coord_c"], t["pose"])
        Ic.backward(t["g"])
        return Ic
    return step


def cpu_baseline(wl, seconds):
    """The CPU oracle (a torch-CPU restatement of the reference's ops, checked against
    fixtures captured from the reference) on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import fc3d_oracle as orc
    cfg = wl["cfg"]
    Bs = min(4, cfg["B"])
    H, W, L = cfg["H"], cfg["W"], cfg["L"]
    t = torch.from_numpy
    z_p, z_c, pose = t(wl["z_p"][:Bs]), t(wl["z_c"][:Bs]), t(wl["pose_p2c"][:Bs])
    cp, cc = orc.z2pointcloud(z_p), orc.z2pointcloud(z_c)
    g = t(wl["upstream_grad"][:Bs])
    p0 = t(wl["pattern"][:Bs]).requires_grad_(True)

    def one():
        p0.grad = None
        pat = p0.expand(-1, -1, -1, L)           # the reference consumes an L-channel pattern
        _, Ic, _, _ = orc.image_formation(pat, z_p, cp, cc, pose, wl["psfs"], W / 2, H / 2)
        Ic.backward(g)

    def best_of(limit_s, limit_n):
        one()
        n, t0, best = 0, time.perf_counter(), float("inf")
        while True:
            a = time.perf_counter()
            one()
            b = time.perf_counter()
            best = min(best, b - a)
            n += 1
            if b - t0 > limit_s or n >= limit_n:
                return best, n

    threads = torch.get_num_threads()
    best, n = best_of(0.6 * seconds, 10 ** 6)          # bounded by time: about `seconds` of CPU work in all
    torch.set_num_threads(1)
    try:
        best1, n1 = best_of(0.4 * seconds, 10 ** 6)
    finally:
        torch.set_num_threads(threads)
    return dict(value=Bs / best, unit="captures/s", cores=threads, kind="port",
                value_1core=Bs / best1,
                sample=f"{n} repeats of batch {Bs} at {H}x{W}, L={L}, K={cfg['K']} (best of {n}) on {threads} threads, "
                       f"{n1} repeats on 1 thread; host cpu_count={os.cpu_count()}")


def copy_bandwidth(device, mib=512, iters=10):
    """Device-to-device copy rate (read + write bytes per second): what a plain streaming kernel
    reaches on this box, as a yardstick next to the nominal peak."""
    n = mib * 1024 * 1024 // 4
    src = torch.empty(n, dtype=torch.float32, device=device).fill_(1.0)
    dst = torch.empty_like(src)
    for _ in range(2):
        dst.copy_(src)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        dst.copy_(src)
    b.record()
    torch.cuda.synchronize()
    return 2.0 * n * 4 * iters / (a.elapsed_time(b) * 1e-3)


def rank_rates(value_rank, world, dist, device):
    """Every rank's own rate (its captures / its own elapsed time), gathered on all ranks."""
    if dist is None:
        return [value_rank]
    t = torch.zeros(world, dtype=torch.float64, device=device)
    t[dist.get_rank()] = value_rank
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t.tolist()]


def scaling_fields(args, value, world, dist_world, per_rank):
    out = {"world_size": dist_world,
           "per_rank_captures_per_s": {"min": min(per_rank), "max": max(per_rank)}}
    if args.n1_value:
        out["scaling_efficiency_vs_n1"] = value / (world * args.n1_value)
    return out


def selftest(args, world, rank):
    """The N-rank plumbing of this benchmark without a GPU: rendezvous (gloo), barrier, K timed stand-in
    steps, MAX over ranks, one JSON line from rank 0."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
    B = args.batch or 2
    x = np.random.RandomState(100 * rank).rand(B, 64, 64).astype(np.float32)

    def step():
        return float(np.sqrt(x * x + 1.0).sum())
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    own = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    per_rank = rank_rates(B * args.steps / own, world, dist if world > 1 else None, "cpu")
    if rank == 0:
        value = world * B * args.steps / elapsed
        out = {"metric": "launch self-test (host stand-in step, no GPU)", "selftest": True, "value": value,
               "unit": "stand-in steps*batch/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": 1e3 * elapsed / args.steps, "scaling": "weak",
               "config": {"global_batch": world * B, "parallelism": f"batch-sharded x{world}, no collective"}}
        out.update(scaling_fields(args, value, world, dist.get_world_size() if world > 1 else 1, per_rank))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    from fc3d import launch as launcher
    if args.gpus > 1 and not launcher.launched_by_torchrun():
        # invoked plainly: start the N ranks as child processes (one per GPU) before this process has made
        # any GPU call, relay their output and leave with their status
        sys.exit(launcher.spawn_ranks(__file__, sys.argv[1:], args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.launch_selftest:
        return selftest(args, world, rank)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if args.rehearse_on_one_gpu:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    coll_device = device                 # where the tensors of the timing collectives live
    if world > 1:
        import torch.distributed as dist
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
            coll_device = torch.device("cpu")
        else:
            dist.init_process_group("nccl", device_id=device)

    from fc3d import ops as fc3d_ops, profiler
    fc3d_ops.set_image_gradient_mode(args.bwd)
    # The headline reads the scene points as the tensors they are handed over as (16 B per pixel and warp: the
    # 100 B/px the roofline is quoted on).  The library's default for points that come from
    # points_in_camera_coords -- read the 4-byte depth behind them and form the point in the kernel -- is
    # measured after the timed region and reported beside it.
    fc3d_ops.set_points_from_depth(False)
    wl, tens = build_inputs(args, rank, device)
    cfg = wl["cfg"]
    B, H, W = cfg["B"], cfg["H"], cfg["W"]
    step = make_step(tens)          # eager step: used for the per-kernel event pass in either mode
    launch = "eager"
    gd, gd_form = None, None

    def run_steps(k):
        for _ in range(k):
            step()
    if args.launch == "graph":
        from fc3d.graphs import GraphedImageFormation
        # the K timed steps go out as K / unroll launches of a graph that holds `unroll` consecutive
        # steps (the largest common divisor of K and W up to 10): same kernels, order and buffers as K single-step
        # replays, without the pause between two graph launches after every step
        # ... that also divides W, so that the warm-up steps run the very graph (and touch the very buffers)
        # that the timed steps use
        unroll = max(u for u in range(1, 11) if args.steps % u == 0 and (args.warmup <= 0 or args.warmup % u == 0))
        if args.unroll > 0:
            if args.steps % args.unroll:
                sys.exit("--unroll must divide --steps")
            unroll = args.unroll
        try:
            gf = GraphedImageFormation(tens["pat"], tens["z_p"], tens["coord_p"], tens["coord_c"], tens["pose"],
                                       tens["g"], tens["L"], unroll=unroll, overlap=args.overlap)
            run_steps, launch = gf.replay_steps, f"hip_graph x{unroll}" + (", projection beside codePattern" if args.overlap else "")
            if rank == 0:
                # a second graph of the same step with the depth-plane form of the coordinate reads (captured here,
                # before any eager step has left autograd nodes of the default stream behind; timed after the headline)
                fc3d_ops.set_points_from_depth(True)
                try:
                    gd = GraphedImageFormation(tens["pat"], tens["z_p"], tens["coord_p"], tens["coord_c"], tens["pose"],
                                               tens["g"], tens["L"], unroll=unroll, overlap=args.overlap,
                                               points_from_depth=True)
                    gd_form = fc3d_ops.WarpP2CPoseFn.last_form
                except Exception as e:
                    gd, gd_form = None, f"{type(e).__name__}: {e}"
                finally:
                    fc3d_ops.set_points_from_depth(False)
        except Exception as e:   # capture refused (e.g. by another library's activity): measure eager instead
            print(f"[bench] graph capture failed ({type(e).__name__}: {e}); using eager launches", file=sys.stderr)
            torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Everything that is not the timed region runs before it, so that the timed steps meet a GPU that
    # is already at its working clocks (a fresh process needs a few milliseconds of load for that,
    # more than W short steps provide): the per-kernel event pass (max(K, 100) eager steps with an event pair
    # around every launch) and the copy-rate yardstick.
    n_prof = max(args.steps, 100)          # at least 100 steps: the clocks take some 20 ms of load to settle
    profiler.enable(True)
    profiler.report()
    for _ in range(n_prof):
        step()
    torch.cuda.synchronize()
    prof = profiler.report()
    profiler.enable(False)
    copy_rate = copy_bandwidth(device)

    settle = 0
    if args.settle > 0:
        per_launch = int(launch.split("x")[1].split(",")[0]) if launch.startswith("hip_graph") else 1
        settle = -(-args.settle // per_launch) * per_launch          # whole launches of the timed graph
        run_steps(settle)
    if args.warmup > 0:
        run_steps(args.warmup)
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize()
    own = time.perf_counter() - t0          # this rank's own K steps (per-rank rates below)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], device=coll_device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    per_rank = rank_rates(cfg["B"] * args.steps / own, world, dist, coll_device)

    # the same K steps issued the two other ways, after the timed region (reported beside the headline): one
    # graph launch per step -- what a caller that runs one step per iteration sees -- and launch by launch
    def timed(fn, k):
        fn(min(k, 5))
        torch.cuda.synchronize()
        a = time.perf_counter()
        fn(k)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - a) / k

    def eager_steps(k):
        for _ in range(k):
            step()
    ms_graph_x1 = ms_eager = None
    if rank == 0:
        if launch.startswith("hip_graph"):
            def single_replays(k):
                for _ in range(k):
                    gf.replay()
            ms_graph_x1 = timed(single_replays, args.steps)
        ms_eager = timed(eager_steps, args.steps)

    # the replayed step must give what eager launches give (guards the timed number against a
    # capture that silently computes something else)
    replay_ok = None
    if launch.startswith("hip_graph"):
        Ic_g, gr_g = (x.clone() for x in gf.replay_steps(args.steps))
        Ic_e = step()
        gr_e = tens["pat"].grad
        replay_ok = bool(torch.equal(Ic_g, Ic_e)) and \
            float((gr_g - gr_e).abs().max()) <= 1e-5 * float(gr_e.abs().max())
        if not replay_ok:
            print("[bench] WARNING: graph replay and eager results differ", file=sys.stderr)

    # the same step with the depth-plane form of the coordinate reads (the library's default), through the same
    # kind of graph; its results must equal the stored-points form bit for bit
    from_depth = None
    if rank == 0 and launch.startswith("hip_graph"):
        if gd is None:
            from_depth = {"error": gd_form}
        else:
            Ic_d, gr_d = (x.clone() for x in gd.replay_steps(unroll))
            Ic_p = step()                      # eager, stored-points form (the switch is off)
            same = bool(torch.equal(Ic_d, Ic_p)) and bool(torch.equal(gr_d, tens["pat"].grad))
            gd.replay_steps(max(unroll, (args.settle // unroll) * unroll))
            ms_d = timed(gd.replay_steps, args.steps)
            from_depth = {"form_used": gd_form, "ms_per_step": ms_d, "captures_per_s": B * 1e3 / ms_d,
                          "alg_bytes_per_capture": 76 * H * W,
                          "path_frac_per_gpu_on_76_B_per_px": (B * 1e3 / ms_d) * 76 * H * W / HBM_PEAK,
                          "equals_stored_points_form": same,
                          "note": "coord_p / coord_c carry the note of points_in_camera_coords: the two warp kernels read "
                                  "z (4 B/px) and form the point as stored instead of reading it (16 B/px); not the headline"}

    if rank == 0:
        px = B * H * W
        kern = {k: dict(launches=n, avg_us=1e3 * ms / n, alg_bytes_per_launch=ALG_BYTES.get(k, 0) * px)
                for k, (n, ms) in prof.items()}
        dom = max(kern, key=lambda k: kern[k]["avg_us"] * kern[k]["launches"])
        dk = kern[dom]
        achieved = dk["alg_bytes_per_launch"] / (dk["avg_us"] * 1e-6)
        gpu_sum_us = sum(v["avg_us"] * v["launches"] for v in kern.values()) / n_prof
        value = world * B * args.steps / elapsed
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    traffic = json.load(f).get(f"{args.workload}:K{cfg['K']}:B{B}", {}).get(dom)
            except Exception:
                traffic = None
        out = {
            "metric": "simulated captures/sec fwd+bwd @512x512 at 1/2/4/8 GPUs; % HBM roofline",
            "value": value, "unit": "captures/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "ms_per_step_graph_x1": ms_graph_x1, "ms_per_step_eager": ms_eager, "settle_steps": settle,
            "points_from_depth": from_depth,
            **scaling_fields(args, value, world, dist.get_world_size() if dist is not None else 1, per_rank),
            "config": {"workload": f"{args.workload}: {B}x{H}x{W} per GPU, L={cfg['L']} depth layers, "
                                   f"{cfg['K']}x{cfg['K']} '{args.psf_kind}' PSFs, pose theta<={cfg['theta_deg']}deg; "
                                   "codePattern -> warp_p2c fwd + bwd to pattern[...,0]",
                       "global_batch": world * B, "parallelism": f"batch-sharded x{world}, no collective",
                       "launch": launch, "replay_matches_eager": replay_ok, "bwd": args.bwd},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK, "traffic": traffic,
                         "kernel_avg_us": dk["avg_us"], "kernel_clock": "HIP events on the launching stream (2-4 % above rocprofv3's kernel durations)",
                         "alg_bytes_per_launch": dk["alg_bytes_per_launch"],
                         "measured_copy_GBps": copy_rate / 1e9,
                         "path_alg_bytes_per_capture": 100 * H * W,
                         "path_frac_per_gpu": (value / world) * 100 * H * W / HBM_PEAK,
                         "gpu_kernel_us_per_step": gpu_sum_us,
                         # the PSF stage's arithmetic (4 K^2 + 120 flop per pixel, forward + backward) against the
                         # fp32 vector peak: what bounds the step once the PSFs are large
                         "valu": {"alg_flops_per_capture": (4 * cfg["K"] ** 2 + 120) * H * W,
                                  "achieved_TFLOPs": (value / world) * (4 * cfg["K"] ** 2 + 120) * H * W / 1e12,
                                  "peak_TFLOPs": FP32_VECTOR_PEAK / 1e12,
                                  "frac": (value / world) * (4 * cfg["K"] ** 2 + 120) * H * W / FP32_VECTOR_PEAK}},
            "kernels": {k: {"avg_us": round(v["avg_us"], 2), "launches_per_step": v["launches"] / n_prof,
                            "alg_GBps": round(v["alg_bytes_per_launch"] / (v["avg_us"] * 1e-6) / 1e9, 1)}
                        for k, v in kern.items()},
        }
        if args.rehearse_on_one_gpu:
            out["rehearsal"] = f"all {world} ranks on GPU 0, gloo for the timing collectives: exercises the N-rank path, measures no scaling"
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(wl, args.cpu_seconds)
            out["cpu_baseline"] = cb
            out["gpu_over_cpu"] = value / cb["value"]
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
