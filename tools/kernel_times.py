"""Steady-state kernel time per 1080p frame of the production kernels (back-to-back launches on one stream), with the
counting build's test counts beside it.  Development aid: python tools/kernel_times.py [scene ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as R

W, H = 1920, 1080
CASES = {
    "cube": ("cube", 0.0, 0.0, R._ffi.RT_ACCEL_AUTO),
    "spheres": ("spheres", 0.0, 0.0, R._ffi.RT_ACCEL_AUTO),
    "lighthouse_bvh": ("lighthouse", 0.0, 0.0, R._ffi.RT_ACCEL_BVH),
    "lighthouse_brute": ("lighthouse", 0.0, 0.0, R._ffi.RT_ACCEL_BRUTE),
    "lighthouse_refl_bvh": ("lighthouse", 0.3, 0.0, R._ffi.RT_ACCEL_BVH),
    "cube_spheres_refl": ("cube_spheres", 0.3, 0.3, R._ffi.RT_ACCEL_AUTO),
}
names = sys.argv[1:] or list(CASES)
hip = R.HipRenderer(0)
buf = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda:0")
for name in names:
    scene, tr, sr, accel = CASES[name]
    hip.upload(R.stock_scene(scene, W, H, tr, sr))
    st = R._ffi.Stats()
    hip.render_device(buf.data_ptr(), R.make_params(accel=accel, count_rays=True), stats=st)
    p = R.make_params(accel=accel)
    n = 20 if name.endswith("brute") else 300
    for _ in range(max(3, n // 5)):
        hip.render_device(buf.data_ptr(), p)
    best = 1e9
    for rep in range(3):
        hip.timer_start()
        for _ in range(n):
            hip.render_device(buf.data_ptr(), p)
        hip.timer_stop()
        best = min(best, hip.timer_ms() / n)
    print(json.dumps({"scene": name, "us": round(best * 1e3, 2), "rays": st.rays, "tri_tests": st.prim_tests - st.sphere_tests,
                      "tri_exact": st.tri_exact, "sphere_tests": st.sphere_tests, "sphere_exact": st.sphere_exact,
                      "box_tests": st.node_tests}), flush=True)
hip.shutdown()
