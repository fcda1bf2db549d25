"""Child process of test_hip_split_tiles.py: the split-tile knobs are read when the library initialises, so every
setting gets a process of its own.  Prints one line of JSON."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import oracle_lib as O  # noqa: E402
import raytracer_amd as R  # noqa: E402

hip = R.HipRenderer(0)
out = {}
for name, w, h, refl, ranks in (("plain", 320, 200, 0.0, 1), ("reflective", 203, 131, 0.3, 1), ("bands", 160, 120, 0.3, 3)):
    s = R.stock_scene("lighthouse", w, h, tri_reflective=refl)
    hip.upload(s)
    want, ost = O.render_rgb(s)
    if ranks == 1:
        tree, st = hip.render(accel=R._ffi.RT_ACCEL_BVH, count_rays=True)
        brute, sb = hip.render(accel=R._ffi.RT_ACCEL_BRUTE, count_rays=True)
        rays, brute_rays = st.rays, sb.rays
        f64, _ = hip.render_f64(accel=R._ffi.RT_ACCEL_BVH)
        wf, _ = O.render_f64(s)
        out[name + "_f64_equal"] = bool(np.array_equal(f64, wf))
    else:
        # every rank's rows, put together on the host
        tree = np.zeros_like(want)
        brute = np.zeros_like(want)
        rays = brute_rays = 0
        for rank in range(ranks):
            ys = [y for y in range(h) if (y // 16) % ranks == rank]
            a, st = hip.render(accel=R._ffi.RT_ACCEL_BVH, count_rays=True, band_rows=16, n_ranks=ranks, rank=rank)
            b, sb = hip.render(accel=R._ffi.RT_ACCEL_BRUTE, count_rays=True, band_rows=16, n_ranks=ranks, rank=rank)
            tree[ys] = a[: len(ys)]
            brute[ys] = b[: len(ys)]
            rays += st.rays
            brute_rays += sb.rays
    out[name] = {"tree_equals_brute": bool(np.array_equal(tree, brute)), "tree_equals_oracle": bool(np.array_equal(tree, want)),
                 "rays": int(rays), "brute_rays": int(brute_rays), "oracle_rays": int(ost.rays)}
# four samples per pixel, an image that does not fill its last tiles, the counting and the plain kernel
s = R.stock_scene("lighthouse", 171, 99, tri_reflective=0.2)
hip.upload(s)
want, ost = O.render_rgb(s, spp_grid=2)
tree, st = hip.render(accel=R._ffi.RT_ACCEL_BVH, count_rays=True, spp_grid=2)
plain, _ = hip.render(accel=R._ffi.RT_ACCEL_BVH, spp_grid=2)
out["supersampled"] = {"tree_equals_brute": bool(np.array_equal(tree, plain)), "tree_equals_oracle": bool(np.array_equal(tree, want)),
                       "rays": int(st.rays), "brute_rays": int(ost.rays), "oracle_rays": int(ost.rays)}
hip.shutdown()
print(json.dumps(out))
