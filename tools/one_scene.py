"""Render one stock scene N times back to back (profiling target).

    python tools/one_scene.py <scene> [accel] [n] [width height [spp_grid]]

scene: a stock scene name, with the suffix _refl for reflective 0.3 on its triangles (and on its spheres where the
name holds "spheres").  The tile plan is made and tuned first (rt_hip_tune); the N launches that follow are the steady
state, and tools/pmc_summary.py averages the last N dispatches only, so the tuning's probe launches stay out.
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as R
scene = sys.argv[1] if len(sys.argv) > 1 else "cube"
accel = {"auto": R._ffi.RT_ACCEL_AUTO, "brute": R._ffi.RT_ACCEL_BRUTE, "bvh": R._ffi.RT_ACCEL_BVH}[sys.argv[2] if len(sys.argv) > 2 else "auto"]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
W, H = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (1920, 1080)
spp = int(sys.argv[6]) if len(sys.argv) > 6 else 1
tr = 0.3 if scene.endswith("_refl") else 0.0
r = R.HipRenderer(0)
r.upload(R.stock_scene(scene.replace("_refl", ""), W, H, tr, tr if "spheres" in scene else 0.0))
buf = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda:0")
p = R.make_params(accel=accel, spp_grid=spp)
r.tune(p)
torch.cuda.synchronize()
for _ in range(n):
    r.render_device(buf.data_ptr(), p)
torch.cuda.synchronize()
r.shutdown()
