"""Device P3 formatter: time per frame and HBM throughput (algorithmic bytes: image read twice + text written)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import raytracer_amd as R

r = R.HipRenderer(0)
K = 50
for (w, h) in ((1920, 1080), (3840, 2160), (7680, 4320)):
    for kind in ("random", "black"):
        if kind == "random":
            img = torch.randint(0, 256, (h, w, 3), dtype=torch.uint8, device="cuda:0")
        else:
            img = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda:0")
        cap = r.ppm_capacity(w, h)
        text = torch.empty(cap, dtype=torch.uint8, device="cuda:0")
        n = r.format_ppm_device(img.data_ptr(), w, h, text.data_ptr(), cap)
        for _ in range(5):
            r.format_ppm_device(img.data_ptr(), w, h, text.data_ptr(), cap, wait=False)
        r.timer_start()
        for _ in range(K):
            r.format_ppm_device(img.data_ptr(), w, h, text.data_ptr(), cap, wait=False)
        r.timer_stop()
        ms = r.timer_ms() / K
        bytes_alg = 2 * 3 * w * h + n
        print(f"{w}x{h} {kind:6s}: text {n / 1e6:8.1f} MB, {ms * 1e3:8.1f} us/frame, {bytes_alg / ms / 1e6:8.1f} GB/s algorithmic "
              f"({bytes_alg / ms / 1e6 / 8000 * 100:.1f}% of 8 TB/s)", flush=True)
r.shutdown()
