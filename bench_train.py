#!/usr/bin/env python3
"""Training steps/s of the full loop (capture simulation -> U-Nets -> losses incl. the
re-projection loss -> backward -> RCCL gradient all-reduce -> Adam), data parallel.

    python bench_train.py [--gpus N] [--steps K] [--warmup W] [--batch 32] [--size 512]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench_train.py --gpus N ...

Batch is per GPU (weak scaling; the reference's configuration 3 is 32 per GPU on 8 GPUs = 256).
Prints one json line on rank 0.  bench.py (image formation alone) is the headline benchmark;
this one shows the same kernels inside a training step and the cost of the collective."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "depth-estimation_amd"))

# A fresh machine has no tuned-convolution database: the default search compiles and times every
# candidate kernel of every U-Net layer shape, minutes before the first step returns.  FAST takes
# the library's heuristic pick instead (MIOPEN_FIND_MODE=NORMAL in the environment restores the search).
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--loss", choices=("rank_mean", "global"), default="rank_mean",
                    help="global: masked sums and counts added over ranks before the losses are formed (the reference's "
                         "batch semantics); rank_mean: per-rank losses, averaged gradients")
    ap.add_argument("--buckets", type=int, default=3, help="ranges the 69 MB gradient buffer is all-reduced in")
    ap.add_argument("--no-overlap", action="store_true", help="start the all-reduce after backward instead of from hooks")
    ap.add_argument("--n1-value", type=float, default=None,
                    help="steps/s of a previous --gpus 1 run: the line then carries scaling_efficiency_vs_n1")
    args = ap.parse_args()
    from fc3d import launch as launcher
    if args.gpus > 1 and not launcher.launched_by_torchrun():
        # invoked plainly: the N ranks are started as child processes before this process makes any GPU call
        sys.exit(launcher.spawn_ranks(__file__, sys.argv[1:], args.gpus))
    from fc3d import capture, dist as fdist, synth
    import optics_pytorch as O
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    rank, world, _ = fdist.init_from_env("nccl", dev)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    H = W = args.size
    sc = synth.make_scene(args.batch, H, W, fdist.shard_seed(3000, rank), pose_scale=min(1.0, W / 512))
    O.PSFs_np = synth.make_psfs(21, 5, "disc")
    d = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    z_p, z_c, pose = d(sc["z_p"]), d(sc["z_c"]), d(sc["pose_p2c"])
    torch.manual_seed(1234)          # identical initial weights on every rank
    step = capture.TrainStep(dev, intrinsics=(W / 2, H / 2), loss=args.loss, n_buckets=args.buckets,
                             overlap=not args.no_overlap)
    # The first steps can sit in the convolution library's kernel search for minutes without a
    # word; a heartbeat on stderr tells a watching harness that the run is alive.
    import threading
    alive = threading.Event()

    def heartbeat():
        t_start = time.time()
        while not alive.wait(45.0):
            print(f"[bench_train] rank {rank}: warming up ({time.time() - t_start:.0f} s)", file=sys.stderr, flush=True)
    hb = threading.Thread(target=heartbeat, daemon=True)
    hb.start()
    for i in range(args.warmup):
        step(z_p, z_c, pose)
        torch.cuda.synchronize()
        print(f"[bench_train] rank {rank}: warm-up step {i + 1}/{args.warmup} done", file=sys.stderr, flush=True)
    alive.set()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step(z_p, z_c, pose)
    torch.cuda.synchronize()
    own = time.perf_counter() - t0
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    el = fdist.max_over_ranks(time.perf_counter() - t0, dev)
    slowest_own = fdist.max_over_ranks(own, dev)
    fastest_own = -fdist.max_over_ranks(-own, dev)
    # where the step goes: the capture simulation alone, timed the same way
    with torch.no_grad():
        capture.forward_model(z_p, z_c, pose, 21, True, None, intrinsics=(W / 2, H / 2))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            capture.forward_model(z_p, z_c, pose, 21, True, None, intrinsics=(W / 2, H / 2))
        torch.cuda.synchronize()
        sim_ms = 1e3 * (time.perf_counter() - t1) / args.steps
    if rank == 0:
        print(json.dumps({"metric": "training steps/sec (capture simulation + 2 U-Nets + reproj loss + Adam), data parallel",
                          "value": args.steps / el, "unit": "steps/s", "samples_per_s": world * args.batch * args.steps / el,
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps,
                          "world_size": torch.distributed.get_world_size() if world > 1 else 1,
                          "per_rank_steps_per_s": {"min": args.steps / slowest_own, "max": args.steps / fastest_own},
                          **({"scaling_efficiency_vs_n1": (args.steps / el) / args.n1_value} if args.n1_value else {}),
                          "scaling": "weak", "dtype": "f32", "data": "synthetic",
                          "config": {"workload": f"{args.batch}x{H}x{W} per GPU, global batch {world * args.batch}",
                                     "grad_allreduce_bytes": step.bucket.nbytes, "grad_ranges": len(step.bucket.ranges),
                                     "overlap": step.bucket.overlap, "loss_mode": args.loss,
                                     "conv_find_mode": os.environ["MIOPEN_FIND_MODE"]},
                          "capture_simulation_ms": sim_ms, "capture_share": sim_ms / (1e3 * el / args.steps),
                          "loss": float(loss)}), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
