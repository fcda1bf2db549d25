"""Frames of one scene issued round-robin on S streams (each with its own output image): how much of a frame's
fill/drain latency does the next frame hide?  Whole-batch time from HIP events on the default stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as R

r = R.HipRenderer(0)
K = 400
for scene, accel in (("cube", R._ffi.RT_ACCEL_AUTO), ("spheres", R._ffi.RT_ACCEL_AUTO), ("lighthouse", R._ffi.RT_ACCEL_BVH),
                     ("lighthouse_refl", R._ffi.RT_ACCEL_BVH)):
    r.upload(R.stock_scene(scene.replace("_refl", ""), 1920, 1080, 0.3 if scene.endswith("_refl") else 0.0))
    for S in (1, 2, 3, 4, 8):
        p = R.make_params(accel=accel, frames_overlap=S > 1)
        streams = [torch.cuda.Stream() for _ in range(S)]
        bufs = [torch.zeros((1080, 1920, 3), dtype=torch.uint8, device="cuda:0") for _ in range(S)]
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            for i in range(2 * S):
                r.render_device(bufs[i % S].data_ptr(), p, stream=streams[i % S].cuda_stream)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for s in streams:
                s.wait_event(e0)
            for i in range(K):
                r.render_device(bufs[i % S].data_ptr(), p, stream=streams[i % S].cuda_stream)
            for s in streams:
                torch.cuda.current_stream().wait_stream(s)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / K * 1e3)
        same = all(bool(torch.equal(b, bufs[0])) for b in bufs)
        print(f"{scene:16s} streams {S}: {best:7.2f} us/frame  images identical={same}", flush=True)
r.shutdown()
