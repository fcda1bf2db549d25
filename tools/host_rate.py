"""How fast can the host enqueue frames, against how fast the GPU renders them?  Development aid."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as R

hip = R.HipRenderer(0)
buf = torch.zeros((1080, 1920, 3), dtype=torch.uint8, device="cuda:0")
for scene in sys.argv[1:] or ["cube"]:
    hip.upload(R.stock_scene(scene, 1920, 1080))
    p = R.make_params()
    for _ in range(50):
        hip.render_device(buf.data_ptr(), p)
    torch.cuda.synchronize()
    n = 2000
    hip.timer_start()
    t0 = time.perf_counter()
    for _ in range(n):
        hip.render_device(buf.data_ptr(), p)
    t1 = time.perf_counter()
    hip.timer_stop()
    gpu = hip.timer_ms() / n * 1e3
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{scene}: host enqueue {1e6 * (t1 - t0) / n:.2f} us/frame, GPU span {gpu:.2f} us/frame, wall to idle {1e6 * (t2 - t0) / n:.2f} us/frame")
hip.shutdown()
