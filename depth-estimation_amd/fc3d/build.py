"""Builds libfc3d_hip.so (gfx950 code objects + the C ABI) in-tree with hipcc.

    python depth-estimation_amd/fc3d/build.py [--force]

hipcc cross-compiles without a GPU.  The library lands next to this file so
that it travels with the source tree; nothing is installed.
"""
from __future__ import annotations

import hashlib
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


STAMP = LIB + ".stamp"      # sha256 of what the library was built from (sources, header, flags, compiler version)


def source_hash(extra_flags=()) -> str:
    """sha256 over every file of csrc/, include/fc3d.h, the compiler flags and the compiler's version string."""
    h = hashlib.sha256()
    for path in sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)) + [os.path.join(ROOT, "include", "fc3d.h")]:
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    h.update(" ".join([*FLAGS, *extra_flags, *SOURCES]).encode())
    try:
        h.update(subprocess.run([_hipcc(), "--version"], capture_output=True, timeout=60).stdout)
    except Exception:
        pass
    return h.hexdigest()


def needs_build() -> bool:
    """True unless the library exists AND was produced from exactly the current sources, header and flags
    (a hash stored beside it; file times say nothing once a tree has been copied or checked out)."""
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return True
    try:
        with open(STAMP) as f:
            return f.read().strip() != source_hash()
    except OSError:
        return True


def build(force: bool = False, verbose: bool = True, extra_flags=(), out: str = LIB) -> str:
    """extra_flags / out: experiment builds (e.g. -D switches) into another file.
    The sources are compiled side by side (one hipcc per file), then linked."""
    if not force and out == LIB and not needs_build():
        return LIB
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    cflags = [f for f in FLAGS if f != "-shared"]
    with tempfile.TemporaryDirectory(prefix="fc3d_build_") as tmp:
        def compile_one(src):
            obj = os.path.join(tmp, src + ".o")
            cmd = [_hipcc(), *cflags, *extra_flags, "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-c",
                   os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            return obj
        with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1, 8)) as pool:
            objs = list(pool.map(compile_one, SOURCES))
        link = [_hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", "-fno-gpu-rdc", *objs, "-o", out + ".tmp"]
        if verbose:
            print(" ".join(link), flush=True)
        subprocess.check_call(link)
    os.replace(out + ".tmp", out)
    if out == LIB:
        with open(STAMP, "w") as f:
            f.write(source_hash(extra_flags) + "\n")
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
