"""Benchmark configurations 4 and 5 at full size: warmed, non-counting kernels, HIP events over back-to-back frames."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as R

r = R.HipRenderer(0)
for name, scene, w, h, tr, sr, spp, accel, K in (("C4 lighthouse 4K reflective 0.3, BVH", "lighthouse", 3840, 2160, 0.3, 0.0, 1, R._ffi.RT_ACCEL_BVH, 20),
                                                  ("C5 cube+spheres 8K, 16 spp, reflective", "cube_spheres", 7680, 4320, 0.3, 0.3, 4, R._ffi.RT_ACCEL_AUTO, 3)):
    r.upload(R.stock_scene(scene, w, h, tr, sr))
    buf = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda:0")
    st = R._ffi.Stats()
    r.render_device(buf.data_ptr(), R.make_params(spp_grid=spp, accel=accel, count_rays=True), stats=st)
    p = R.make_params(spp_grid=spp, accel=accel)
    r.render_device(buf.data_ptr(), p)
    r.timer_start()
    for _ in range(K):
        r.render_device(buf.data_ptr(), p)
    r.timer_stop()
    ms = r.timer_ms() / K
    print(f"{name}: {st.rays} rays, {ms:.3f} ms/frame, {st.rays / ms / 1e3:.0f} Mrays/s", flush=True)
r.shutdown()
