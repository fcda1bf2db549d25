"""Back-to-back launch cost of the trace kernel on images of growing size (the per-launch floor)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as R
K = 500
r = R.HipRenderer(0)
for (w, h) in ((16, 16), (256, 16), (1920, 16), (1920, 128), (1920, 540), (1920, 1080)):
    r.upload(R.stock_scene("cube", w, h))
    buf = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda:0")
    p = R.make_params()
    for _ in range(20):
        r.render_device(buf.data_ptr(), p)
    torch.cuda.synchronize()
    r.timer_start()
    for _ in range(K):
        r.render_device(buf.data_ptr(), p)
    r.timer_stop()
    print(f"{w}x{h}: {r.timer_ms() / K * 1e3:7.2f} us/launch", flush=True)
r.shutdown()
