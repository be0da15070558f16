"""Per-kernel timing through the library's HIP-event hooks."""
from __future__ import annotations

import ctypes

from . import _native


def enable(on: bool = True) -> None:
    _native.check(_native.load().fc3d_profile_enable(int(on)), "fc3d_profile_enable")


def report() -> dict:
    """{kernel name: (launches, total milliseconds)} since the last report; waits for the
    recorded events, so call it after the work of interest has been enqueued."""
    buf = ctypes.create_string_buffer(1 << 16)
    _native.check(_native.load().fc3d_profile_report(buf, len(buf)), "fc3d_profile_report")
    out = {}
    for line in buf.value.decode().splitlines():
        name, n, ms = line.split()
        out[name] = (int(n), float(ms))
    return out
