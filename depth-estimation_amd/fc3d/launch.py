"""One process per GPU, started from a plain `python script.py --gpus N`.

A benchmark invoked without torch.distributed.run's environment (no WORLD_SIZE) and asked for
N > 1 GPUs starts the N ranks itself: as CHILD processes of a `python -m torch.distributed.run`
child, before the parent has made any GPU call (a process that has initialised the GPU must not
be replaced or forked into ranks), and relays what the ranks print.  The parent never touches
the GPU; it only waits and exits with the child's status.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys


def launched_by_torchrun() -> bool:
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def rank_command(script: str, argv, n_ranks: int, port: int | None = None):
    """The command line that runs `script argv...` as n_ranks ranks of one node."""
    port = free_port() if port is None else int(port)
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_ranks)}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(script), *argv]


def spawn_ranks(script: str, argv, n_ranks: int, timeout: float | None = None, extra_env: dict | None = None) -> int:
    """Runs the ranks to completion; their stdout/stderr are inherited (rank 0's JSON line arrives on
    this process's stdout unchanged).  Returns the exit status to leave with."""
    if launched_by_torchrun():
        raise RuntimeError("spawn_ranks called from inside a rank")
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # RCCL needs dmabuf IPC on these hosts
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, n_ranks))))
    if extra_env:
        env.update(extra_env)
    cmd = rank_command(script, list(argv), n_ranks)
    print("[launch] " + " ".join(cmd), file=sys.stderr, flush=True)
    try:
        return subprocess.run(cmd, env=env, timeout=timeout).returncode
    except subprocess.TimeoutExpired:
        print(f"[launch] ranks did not finish within {timeout} s", file=sys.stderr, flush=True)
        return 124
