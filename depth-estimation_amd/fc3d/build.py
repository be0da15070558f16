"""Builds libfc3d_hip.so (gfx950 code objects + the C ABI) in-tree with hipcc.

    python depth-estimation_amd/fc3d/build.py [--force]

hipcc cross-compiles without a GPU.  The library lands next to this file so
that it travels with the source tree; nothing is installed.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libfc3d_hip.so")
SOURCES = ["optics.hip", "geometry.hip", "capture.hip", "profile.hip"]

# -ffp-contract=off                         index/mask decisions need the reference's separately rounded ops
# -fhip-fp32-correctly-rounded-divide-sqrt  IEEE fp32 division (torch's CPU kernels divide exactly)
# -munsafe-fp-atomics                       hardware float atomic add instead of a compare-and-swap loop
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-munsafe-fp-atomics", "-fno-gpu-rdc"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; cannot build libfc3d_hip.so")
    return exe


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "fc3d.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, extra_flags=(), out: str = LIB) -> str:
    """extra_flags / out: experiment builds (e.g. -D switches) into another file."""
    if not force and out == LIB and not needs_build():
        return LIB
    cmd = [_hipcc(), *FLAGS, *extra_flags, "-I", os.path.join(ROOT, "include"), "-I", CSRC,
           *[os.path.join(CSRC, s) for s in SOURCES], "-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
