import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


def load_golden(entry, key):
    z = np.load(os.path.join(GOLDEN, entry[key]), allow_pickle=False)
    return z["canvas" if key == "canvas" else "rgb"]


def quantise(canvas):
    """color_convert (RayTracer.c:1322) on a double canvas."""
    v = canvas * 255.0
    ok = (v > -2147483649.0) & (v < 2147483648.0)  # anything else casts to INT_MIN on x86-64, i.e. 0 after the clamp
    return np.clip(np.trunc(np.where(ok, v, 0.0)), 0, 255).astype(np.uint8)


@pytest.fixture(scope="session")
def hip():
    """The HIP renderer on cuda:0; fails loudly if the library or the device is missing."""
    import raytracer_amd as R

    r = R.HipRenderer(0)
    yield r
    r.shutdown()
