import sys, os
sys.path.insert(0, "/root/repo")
import raytracer_amd as R
r = R.HipRenderer(0)
s = R.stock_scene("lighthouse", 1920, 1080)
r.upload(s)
for accel, an in ((R._ffi.RT_ACCEL_BRUTE, "brute"), (R._ffi.RT_ACCEL_BVH, "bvh")):
    for mode in (0, 4):
        _, st = r.render(accel=accel, count_rays=True, flags=mode << 8)
        print(an, "ablate", mode, "prim_tests", st.prim_tests, "tri_exact", st.tri_exact, "rays", st.rays, flush=True)
r.shutdown()
