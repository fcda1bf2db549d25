"""Kernel time of ONE rank's share of the 1080p frame for N = 1, 2, 4, 8 ranks (measured on one GPU):
the render-side ceiling of strong scaling, and how fast Python can issue launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as R
from raytracer_amd.dist import default_band_rows

K = 400
scene = sys.argv[1] if len(sys.argv) > 1 else "cube"
r = R.HipRenderer(0)
r.upload(R.stock_scene(scene, 1920, 1080))
buf = torch.zeros((1080, 1920, 3), dtype=torch.uint8, device="cuda:0")
for n in (1, 2, 4, 8):
    p = R.make_params(band_rows=default_band_rows(1080, n), n_ranks=n, rank=0)
    for _ in range(20):
        r.render_device(buf.data_ptr(), p)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r.timer_start()
    for _ in range(K):
        r.render_device(buf.data_ptr(), p)
    t_issue = time.perf_counter() - t0
    r.timer_stop()
    torch.cuda.synchronize()
    print(f"{scene} N={n}: {r.timer_ms() / K * 1e3:7.2f} us/launch on the GPU, {t_issue / K * 1e6:6.2f} us/launch to issue from Python", flush=True)
r.shutdown()
