"""Kernel-time survey over scenes and traversal modes (development aid, not the bench contract)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as R

def main():
    W, H = 1920, 1080
    r = R.HipRenderer(0)
    cases = [("cube", 0, 0), ("spheres", 0, 0), ("lighthouse", 0, 0), ("lighthouse", 0.3, 0), ("cube_spheres", 0.3, 0.3)]
    for name, tr, sr in cases:
        s = R.stock_scene(name, W, H, tr, sr)
        r.upload(s)
        for accel, an in ((R._ffi.RT_ACCEL_BRUTE, "brute"), (R._ffi.RT_ACCEL_BVH, "bvh")):
            if accel == R._ffi.RT_ACCEL_BVH and s.n_tris == 0:
                continue
            _, st = r.render(accel=accel, count_rays=True)
            rays = st.rays
            ms = []
            for _ in range(5):
                _, st2 = r.render(accel=accel)
                ms.append(st2.kernel_ms)
            m = float(np.median(ms))
            print(json.dumps(dict(scene=name, refl=(tr, sr), accel=an, rays=rays, prim_tests=st.prim_tests,
                                  node_tests=st.node_tests, tri_exact=st.tri_exact, kernel_ms=round(m, 4), mrays_s=round(rays / m / 1e3, 1))), flush=True)
    r.shutdown()

main()
