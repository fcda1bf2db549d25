"""Where does the frame time go?  Back-to-back launches (the clock stays up), HIP events around the batch.
Bits 8-15 of the `flags` field of rt_render_params select a timing-only cut of the kernel (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import raytracer_amd as R

K = 300
r = R.HipRenderer(0)
buf = torch.zeros((1080, 1920, 3), dtype=torch.uint8, device="cuda:0")
scenes = sys.argv[1:] or ["cube", "spheres", "lighthouse"]
for scene in scenes:
    r.upload(R.stock_scene(scene, 1920, 1080))
    if os.environ.get("RT_ABLATE_MODES"):
        want = {int(x) for x in os.environ["RT_ABLATE_MODES"].split(",")}
    else:
        want = None
    MODES = ((1, "store only"), (6, "store black"), (2, "ray generation"), (3, "primary hit"), (4, "no shadow ray"), (5, "no lighting"), (0, "full")) if os.environ.get("RT_ABLATE") else ((0, "full"),)
    for mode, name in MODES:
        if want is not None and mode not in want:
            continue
        for accel, an in ((R._ffi.RT_ACCEL_BRUTE, "brute"), (R._ffi.RT_ACCEL_BVH, "bvh")):
            if mode != 0 and an == "bvh" and not os.environ.get("RT_ABLATE_BVH"):
                continue
            if an == "bvh" and scene == "spheres":
                continue
            p = R.make_params(flags=mode << 8, accel=accel)
            res = []
            for rep in range(3):
                for _ in range(20):
                    r.render_device(buf.data_ptr(), p)
                r.timer_start()
                for _ in range(K):
                    r.render_device(buf.data_ptr(), p)
                r.timer_stop()
                res.append(r.timer_ms() / K * 1e3)
            print(f"{scene:10s} {name:15s} {an:6s} {min(res):8.2f} us/frame (runs: {' '.join(f'{x:.1f}' for x in res)})", flush=True)
r.shutdown()
