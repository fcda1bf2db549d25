"""1080p frames of the worlds that go through the general kernel (planes, patterns, cube, cylinder, glass)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as R
r = R.HipRenderer(0)
buf = torch.zeros((1080, 1920, 3), dtype=torch.uint8, device="cuda:0")
for scene in sys.argv[1:] or ["refraction", "world_with_plane", "dual_spheres"]:
    r.upload(R.stock_scene(scene, 1920, 1080))
    st = R._ffi.Stats()
    r.render_device(buf.data_ptr(), R.make_params(count_rays=True), stats=st)
    p = R.make_params()
    for _ in range(3):
        r.render_device(buf.data_ptr(), p)
    K = 20
    r.timer_start()
    for _ in range(K):
        r.render_device(buf.data_ptr(), p)
    r.timer_stop()
    ms = r.timer_ms() / K
    print(f"{scene:18s} {st.rays:10d} rays  {ms:8.3f} ms/frame  {st.rays / ms / 1e3:9.1f} Mrays/s", flush=True)
r.shutdown()
