"""Frame time of the lighthouse at 1080p against the BVH leaf size (host build parameter)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as R
r = R.HipRenderer(0)
buf = torch.zeros((1080, 1920, 3), dtype=torch.uint8, device="cuda:0")
K = 200
for refl in (0.0, 0.3):
    for leaf in (1, 2, 3, 4, 6, 7):
        s = R.stock_scene("lighthouse", 1920, 1080, refl, 0.0, bvh_leaf_size=leaf)
        r.upload(s)
        p = R.make_params(accel=R._ffi.RT_ACCEL_BVH)
        st = R._ffi.Stats()
        r.render_device(buf.data_ptr(), R.make_params(accel=R._ffi.RT_ACCEL_BVH, count_rays=True), stats=st)
        best = 1e9
        for rep in range(3):
            for _ in range(20):
                r.render_device(buf.data_ptr(), p)
            r.timer_start()
            for _ in range(K):
                r.render_device(buf.data_ptr(), p)
            r.timer_stop()
            best = min(best, r.timer_ms() / K * 1e3)
        print(f"reflective {refl}: leaf<={leaf}: nodes {s.n_bvh_nodes:5d}  {best:7.2f} us/frame  box tests {st.node_tests}  tri tests {st.prim_tests}", flush=True)
r.shutdown()
