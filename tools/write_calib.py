"""Calibration of WRITE_SIZE on this image's access patterns: a frame whose tiles are all background (16-byte fill
stores only) and torch's own fill of the same 6.2 MB buffer.  Run under rocprofv3 --pmc WRITE_SIZE."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import raytracer_amd as R
from raytracer_amd.scene import SceneBuilder

b = SceneBuilder()
b.set_camera(1920, 1080, 0.5, (0.0, 0.0, -5.0), (0.0, 0.0, 0.0))
b.set_light((-10.0, 10.0, -10.0), (1.0, 1.0, 1.0))
b.add_triangles_f32(np.array([[40, 40, 5, 41, 40, 5, 40, 41, 5]], dtype=np.float32))  # far outside the view
s = b.finish(4)
r = R.HipRenderer(0)
r.upload(s)
buf = torch.zeros((1080, 1920, 3), dtype=torch.uint8, device="cuda:0")
p = R.make_params()
for _ in range(5):
    r.render_device(buf.data_ptr(), p)
torch.cuda.synchronize()
for _ in range(5):
    buf.fill_(7)
torch.cuda.synchronize()
r.shutdown()
