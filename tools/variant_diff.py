"""Pixels of the golden images that a development build of the library renders differently, and its frame times.

    RT_HIP_LIB=<tag> python tools/variant_diff.py

For A/B builds that change arithmetic (e.g. -DRT_SHADE_FMA=1): every golden image of tests/golden/manifest.json is
rendered and compared byte for byte with the reference's; then the 1080p spheres frame and (optionally) the 8K x 16
samples frame are timed."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import raytracer_amd as R

man = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))
hip = R.HipRenderer(0)
total = differing = 0
worst = 0
for name, e in sorted(man.items()):
    if "rgb" not in e or e.get("scene") not in ("lighthouse", "cube", "spheres", "cube_spheres"):
        continue
    want = np.load(os.path.join(ROOT, "tests", "golden", e["rgb"]))["rgb"]
    hip.upload(R.stock_scene(e["scene"], e["width"], e["height"], e["tri_reflective"], e["sph_reflective"]))
    got, _ = hip.render(spp_grid=e.get("spp_grid", 1), stats=False)
    d = np.abs(got.astype(np.int16) - want.astype(np.int16)).max(axis=2)
    n = int((d != 0).sum())
    total += d.size
    differing += n
    worst = max(worst, int(d.max()))
    if n:
        print(f"  {name}: {n} of {d.size} pixels differ (largest step {int(d.max())})")
print(f"goldens: {differing} of {total} pixels differ, largest step {worst}")
for label, scene, w, h, tr, sr, spp, n in (("spheres 1080p", "spheres", 1920, 1080, 0.0, 0.0, 1, 200),
                                            ("C5 8K x 16 spp", "cube_spheres", 7680, 4320, 0.3, 0.3, 4, 3)):
    hip.upload(R.stock_scene(scene, w, h, tr, sr))
    buf = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda:0")
    p = R.make_params(spp_grid=spp)
    for _ in range(3):
        hip.render_device(buf.data_ptr(), p)
    best = 1e9
    for _ in range(3):
        hip.timer_start()
        for _ in range(n):
            hip.render_device(buf.data_ptr(), p)
        hip.timer_stop()
        best = min(best, hip.timer_ms() / n)
    print(f"{label}: {best * 1e3:.1f} us per frame")
    del buf
hip.shutdown()
