"""Device-built BVH (rt_hip_build_bvh_device) against the host's binned-SAH tree: build time and 1080p frame time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import raytracer_amd as R
from test_hip_parity import icosphere, mesh_scene

W, H = 1920, 1080
r = R.HipRenderer(0)
buf = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda:0")
BVH = R._ffi.RT_ACCEL_BVH


def frame_ms(K=20):
    p = R.make_params(accel=BVH)
    for _ in range(3):
        r.render_device(buf.data_ptr(), p)
    r.timer_start()
    for _ in range(K):
        r.render_device(buf.data_ptr(), p)
    r.timer_stop()
    return r.timer_ms() / K


def scenes():
    t0 = time.perf_counter()
    s = R.stock_scene("lighthouse", W, H)
    yield "lighthouse", s, time.perf_counter() - t0
    for level in (6, 7, 8):
        t0 = time.perf_counter()
        s = mesh_scene(icosphere(level), W, H, reflective=float(os.environ.get("LBVH_REFL", "0")))
        yield f"icosphere{level}", s, time.perf_counter() - t0


leaf = int(os.environ.get("LBVH_LEAF", "4"))
for name, s, host_s in scenes():
    r.upload(s)
    host_ms = frame_ms()
    want = buf.clone()
    r.build_bvh_device(leaf)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nodes, depth, build_ms = r.build_bvh_device(leaf)
    call_ms = (time.perf_counter() - t0) * 1e3
    dev_ms = frame_ms()
    same = bool(torch.equal(buf, want))
    print(f"{name:12s} {s.n_tris:8d} triangles  host scene+SAH {host_s * 1e3:9.1f} ms  device tree {build_ms:7.3f} ms on the GPU, {call_ms:7.3f} ms the call (depth {depth})  "
          f"frame {host_ms * 1e3:8.1f} us (SAH) {dev_ms * 1e3:8.1f} us (device tree)  identical={same}", flush=True)
r.shutdown()
