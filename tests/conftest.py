import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names(kind=None):
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    if kind is not None:
        names = [n for n in names if n.startswith(kind)]
    return names


def load_golden(name):
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def assert_close(a, b, rtol, what=""):
    """|a-b| <= rtol * max(1, |b|) elementwise, NaN patterns equal."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.array_equal(np.isnan(a), np.isnan(b)), f"{what}: NaN pattern differs"
    a, b = np.nan_to_num(a), np.nan_to_num(b)
    scaled = np.abs(a - b) / np.maximum(1.0, np.abs(b))
    worst = scaled.max() if scaled.size else 0.0
    assert worst <= rtol, f"{what}: max scaled error {worst:.3e} > {rtol:.1e}"


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN_DIR
