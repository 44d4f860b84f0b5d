import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Small batches are answered by the exact scan (option "scan_max"); the parity tests are mostly small,
# so by default they pin the tree-walking kernels (strict, packet) and tests/test_gpu_scan.py covers
# the scan with the library's own default.
os.environ.setdefault("SDFHIP_SCAN_MAX", "0")
SCAN_MAX_DEFAULT = -1  # automatic

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
GOLDEN_NAMES = ["teapot", "torus", "boxframe", "cubegrid", "syntorus83", "icosphere3"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN_DIR, f"golden_{name}.npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session", params=GOLDEN_NAMES)
def golden(request):
    g = load_golden(request.param)
    g["name"] = request.param
    return g


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    from oracle import pyoracle
    pyoracle.build()
    yield


def same_float(a, b):
    """bit-equality that treats any NaN as equal to any NaN (x86 sqrt(-x) gives -NaN, the GPU
    gives +NaN; the reference defines no payload)."""
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
