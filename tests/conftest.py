import os
import sys

# The convolution library reads its search mode once, when it is first used: without this every new
# layer shape of the U-Nets is searched exhaustively (minutes at 32x512x512 on a machine without a
# tuning database).  FAST takes the library's heuristic pick.  Must be set before torch is imported.
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "depth-estimation_amd")
for p in (PKG, os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as f:
        return {k: f[k] for k in f.files}


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get


def rel_err(a, b):
    """max |a-b| / max(|b|) -- the scale-relative error used for the fp32 tolerance."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-30))


def rel_err_elem(a, b, floor=1e-3):
    """max over the elements with |b| > floor * max|b| of |a-b| / |b|: a per-element relative error
    that ignores values too small to carry one (next to rel_err, which is relative to the tensor's scale)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    big = np.abs(b) > floor * max(float(np.max(np.abs(b))), 1e-30)
    if not big.any():
        return 0.0
    return float(np.max(np.abs(a - b)[big] / np.abs(b)[big]))
