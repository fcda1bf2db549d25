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

    class Checked(R.HipRenderer):
        """Tests ask for ray counts, which selects the counting build of a kernel family.  What ships is the build
        without counters, specialised further (primitive kinds, LDS staging, no reflection stack): render it as
        well, every time, and require the same bytes, so that every parity test holds for both."""

        def render(self, params=None, stats=True, **kw):
            out, st = super().render(params, stats, **kw)
            counted = params.count_rays if params is not None else kw.get("count_rays", False)
            if counted:
                if params is not None:
                    q = type(params).from_buffer_copy(params)
                    q.count_rays = 0
                    plain, _ = super().render(q, stats)
                else:
                    plain, _ = super().render(None, stats, **dict(kw, count_rays=False))
                assert np.array_equal(plain, out), "production kernel and counting kernel disagree"
            return out, st

        def render_f64(self, params=None, **kw):
            out, st = super().render_f64(params, **kw)
            if params is None and kw.get("count_rays", False):
                plain, _ = super().render_f64(None, **dict(kw, count_rays=False))
                assert np.array_equal(plain, out, equal_nan=True), "production kernel and counting kernel disagree"
            return out, st

    r = Checked(0)
    yield r
    r.shutdown()
