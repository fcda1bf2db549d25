"""Development aid for a build with per-wave maxima in the counters (not part of the product)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as R
r = R.HipRenderer(0)
buf = torch.zeros((1080, 1920, 3), dtype=torch.uint8, device="cuda:0")
for scene in sys.argv[1:] or ["lighthouse"]:
    r.upload(R.stock_scene(scene, 1920, 1080))
    for rep in range(300):
        st = R._ffi.Stats()
        r.render_device(buf.data_ptr(), R.make_params(accel=R._ffi.RT_ACCEL_BVH, count_rays=True), stats=st)
    print(scene, "kernel_ms", round(st.kernel_ms, 4), "max per wave: block cycles", st.rays_primary, "| primary walk: iters", st.rays_shadow,
          "cycles", st.prim_tests, "| shadow walk: iters", st.rays_secondary, "cycles", st.node_tests, "| leaf iters", st.sphere_tests, flush=True)
r.shutdown()
