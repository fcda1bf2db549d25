"""Kernel time of a few frames with the library RT_HIP_LIB selects (A/B runs: one process per variant, same box).

    RT_HIP_LIB=<tag> python tools/ab_frames.py scene[:refl[:W:H[:grid[:accel]]]] ...      (accel: auto, brute, bvh)

Each frame is tuned first, then timed over back-to-back launches with HIP events (best of three batches).
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as R

r = R.HipRenderer(0)
for spec in sys.argv[1:]:
    f = spec.split(":")
    scene = f[0]
    refl = float(f[1]) if len(f) > 1 else 0.0
    W, H = (int(f[2]), int(f[3])) if len(f) > 3 else (1920, 1080)
    grid = int(f[4]) if len(f) > 4 else 1
    accel = {"auto": R._ffi.RT_ACCEL_AUTO, "brute": R._ffi.RT_ACCEL_BRUTE, "bvh": R._ffi.RT_ACCEL_BVH}[f[5] if len(f) > 5 else "auto"]
    r.upload(R.stock_scene(scene, W, H, refl, refl if "spheres" in scene else 0.0))
    buf = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda:0")
    p = R.make_params(spp_grid=grid, accel=accel)
    r.tune(p)
    r.render_device(buf.data_ptr(), p)
    torch.cuda.synchronize()
    r.timer_start(); r.render_device(buf.data_ptr(), p); r.timer_stop()
    one = r.timer_ms()
    K = max(3, min(300, int(60.0 / max(one, 1e-3))))
    res = []
    for rep in range(3):
        for _ in range(max(2, K // 10)):
            r.render_device(buf.data_ptr(), p)
        r.timer_start()
        for _ in range(K):
            r.render_device(buf.data_ptr(), p)
        r.timer_stop()
        res.append(r.timer_ms() / K * 1e3)
    print(f"{os.environ.get('RT_HIP_LIB', 'default'):10s} {spec:34s} {min(res):10.2f} us/frame  ({' '.join(f'{x:.1f}' for x in res)})", flush=True)
r.shutdown()
