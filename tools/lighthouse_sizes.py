import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import torch
import raytracer_amd as R
r = R.HipRenderer(0)
for name, w, h, tr, K in (("lighthouse 1080p", 1920, 1080, 0.0, 300), ("lighthouse 1080p reflective 0.3", 1920, 1080, 0.3, 200), ("lighthouse 4K", 3840, 2160, 0.0, 100), ("lighthouse 4K reflective 0.3", 3840, 2160, 0.3, 50), ("lighthouse 720p", 1280, 720, 0.0, 300)):
    r.upload(R.stock_scene("lighthouse", w, h, tr, 0.0))
    buf = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda:0")
    p = R.make_params(accel=R._ffi.RT_ACCEL_BVH)
    res = []
    t0 = time.perf_counter()  # the first frames plan the tiles and refine the split
    for _ in range(3):
        r.render_device(buf.data_ptr(), p)
    torch.cuda.synchronize()
    first_ms = (time.perf_counter() - t0) * 1e3
    for rep in range(3):
        for _ in range(20):
            r.render_device(buf.data_ptr(), p)
        r.timer_start()
        for _ in range(K):
            r.render_device(buf.data_ptr(), p)
        r.timer_stop()
        res.append(r.timer_ms() / K * 1e3)
    print(f"{name}: {min(res):.1f} us (first three frames {first_ms:.1f} ms)", end="; ")
print()
r.shutdown()
