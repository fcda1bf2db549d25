"""Render one stock scene at 1920x1080 N times back to back (profiling target)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as R
scene = sys.argv[1] if len(sys.argv) > 1 else "cube"
accel = {"auto": R._ffi.RT_ACCEL_AUTO, "brute": R._ffi.RT_ACCEL_BRUTE, "bvh": R._ffi.RT_ACCEL_BVH}[sys.argv[2] if len(sys.argv) > 2 else "auto"]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
tr = 0.3 if scene.endswith("_refl") else 0.0
r = R.HipRenderer(0)
r.upload(R.stock_scene(scene.replace("_refl", ""), 1920, 1080, tr, tr if "spheres" in scene else 0.0))
buf = torch.zeros((1080, 1920, 3), dtype=torch.uint8, device="cuda:0")
p = R.make_params(accel=accel)
for _ in range(n):
    r.render_device(buf.data_ptr(), p)
torch.cuda.synchronize()
r.shutdown()
