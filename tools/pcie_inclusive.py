"""Headline frame through the host-buffer entry points (device-to-host copy included), next to the device-resident rate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as R
r = R.HipRenderer(0)
r.upload(R.stock_scene("cube", 1920, 1080))
p = R.make_params()
for _ in range(5):
    img, _ = r.render(p, stats=False)
K = 50
t0 = time.perf_counter()
for _ in range(K):
    img, _ = r.render(p, stats=False)
dt = (time.perf_counter() - t0) / K
print(f"rt_hip_render (8-bit image to host memory): {dt * 1e3:.3f} ms/frame = {2306823 / dt / 1e6:.0f} Mrays/s")
for _ in range(3):
    txt, _ = r.render_ppm(p)
t0 = time.perf_counter()
for _ in range(20):
    txt, _ = r.render_ppm(p)
dt = (time.perf_counter() - t0) / 20
print(f"rt_hip_render_ppm (P3 text, {len(txt)} bytes, to host memory): {dt * 1e3:.3f} ms/frame")
t0 = time.perf_counter()
for _ in range(5):
    ref = R.ppm_bytes(img)
print(f"host C writer rt_format_ppm on the same image: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms/frame; identical: {ref == txt}")
r.shutdown()
