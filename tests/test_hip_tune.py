"""The switches around the measured tile plan (include/rt_hip.h: RT_RENDER_NO_TUNE, rt_hip_tune) and the promise of
rt_hip_render_device without stats: it only enqueues.

The reference's use is one process and one render() (RayTracer.c:7236-7251): such a caller renders with the plan as it
stands.  A caller with a run of frames tunes once and then only enqueues.  The image never depends on any of it.
"""
import ctypes as C
import hashlib

import numpy as np
import pytest

import raytracer_amd as R
from conftest import load_golden

pytestmark = pytest.mark.gpu


def _hip_runtime():
    """The HIP runtime this process already holds (the copy librt_hip.so and torch resolved): a second copy must not
    be loaded."""
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64.so" in line:
                return C.CDLL(line.split()[-1])
    raise RuntimeError("no HIP runtime mapped")


def _lighthouse_1080p(manifest):
    e = manifest["lighthouse_1080p"]
    return e, R.stock_scene(e["scene"], e["width"], e["height"], e["tri_reflective"], e["sph_reflective"])


def test_no_tune_on_a_caller_stream_gives_the_reference_image(hip, manifest):
    """RT_RENDER_NO_TUNE, device output, the caller's stream, no stats: the bytes of the reference's 1080p lighthouse."""
    import torch

    e, scene = _lighthouse_1080p(manifest)
    hip.upload(scene)
    want = load_golden(e, "rgb")
    buf = torch.zeros((e["height"], e["width"], 3), dtype=torch.uint8, device="cuda:0")
    stream = torch.cuda.Stream()
    p = R.make_params(no_tune=True)
    L = R._ffi.hip_lib()
    # straight through the C ABI (the Python wrapper would tune first)
    rc = L.rt_hip_render_device(C.byref(p), C.c_void_p(buf.data_ptr()), C.c_void_p(stream.cuda_stream), None)
    assert rc == 0
    stream.synchronize()
    got = buf.cpu().numpy()
    assert np.array_equal(got, want)
    assert hashlib.sha256(R.ppm_bytes(got)).hexdigest() == e["ppm_sha256"]


def test_tune_then_render_gives_the_same_bytes_and_is_not_slower(hip, manifest):
    import torch

    e, scene = _lighthouse_1080p(manifest)
    hip.upload(scene)
    want = load_golden(e, "rgb")
    buf = torch.zeros((e["height"], e["width"], 3), dtype=torch.uint8, device="cuda:0")
    L = R._ffi.hip_lib()
    p = R.make_params(no_tune=True)

    def timed(n=20):
        best = 1e9
        for _ in range(3):
            hip.timer_start()
            for _ in range(n):
                assert L.rt_hip_render_device(C.byref(p), C.c_void_p(buf.data_ptr()), None, None) == 0
            hip.timer_stop()
            best = min(best, hip.timer_ms() / n)
        return best

    static_ms = timed()
    assert np.array_equal(buf.cpu().numpy(), want)
    assert L.rt_hip_tune(C.byref(R.make_params())) == 0
    buf.zero_()
    tuned_ms = timed()
    assert np.array_equal(buf.cpu().numpy(), want)
    print(f"lighthouse 1080p: static plan {static_ms * 1e3:.1f} us, tuned plan {tuned_ms * 1e3:.1f} us")
    assert tuned_ms <= static_ms * 1.10


def test_render_device_without_stats_is_legal_under_stream_capture(hip, manifest):
    """It allocates nothing, creates no event and waits for nothing - with a plan at hand (made at upload) and
    without one (another band layout: raster order).  A captured graph replays to the same bytes."""
    import torch

    e = manifest["lighthouse_300"]
    hip.upload(R.stock_scene(e["scene"], e["width"], e["height"], e["tri_reflective"], e["sph_reflective"]))
    want = load_golden(e, "rgb")
    L = R._ffi.hip_lib()
    hiprt = _hip_runtime()
    buf = torch.zeros((e["height"], e["width"], 3), dtype=torch.uint8, device="cuda:0")
    half = torch.zeros((R.HipRenderer.rank_rows(hip, R.make_params(n_ranks=2, band_rows=8, rank=1)), e["width"], 3),
                       dtype=torch.uint8, device="cuda:0")
    stream = torch.cuda.Stream()
    s = C.c_void_p(stream.cuda_stream)
    p_full = R.make_params(no_tune=True)
    p_band = R.make_params(no_tune=True, n_ranks=2, band_rows=8, rank=1)  # no plan exists for this layout
    graph, exe = C.c_void_p(), C.c_void_p()
    assert hiprt.hipStreamBeginCapture(s, 0) == 0  # hipStreamCaptureModeGlobal
    rc1 = L.rt_hip_render_device(C.byref(p_full), C.c_void_p(buf.data_ptr()), s, None)
    rc2 = L.rt_hip_render_device(C.byref(p_band), C.c_void_p(half.data_ptr()), s, None)
    assert hiprt.hipStreamEndCapture(s, C.byref(graph)) == 0
    assert rc1 == 0 and rc2 == 0
    assert hiprt.hipGraphInstantiate(C.byref(exe), graph, None, None, 0) == 0
    buf.zero_()
    half.zero_()
    assert hiprt.hipGraphLaunch(exe, s) == 0
    stream.synchronize()
    assert np.array_equal(buf.cpu().numpy(), want)
    ys = R.rank_row_map(e["height"], 8, 2, 1)
    assert np.array_equal(half.cpu().numpy(), want[ys])
    hiprt.hipGraphExecDestroy(exe)
    hiprt.hipGraphDestroy(graph)


def test_general_kernel_without_a_plan_renders_in_raster_order(hip, manifest):
    """Worlds of the general kernel order their tiles by a measured estimate; without stats and without a plan the
    tiles go in raster order and the image is the same."""
    import torch

    scene = R.stock_scene("refraction", 160, 120)
    hip.upload(scene)
    want, _ = hip.render()
    buf = torch.zeros((120, 160, 3), dtype=torch.uint8, device="cuda:0")
    L = R._ffi.hip_lib()
    p = R.make_params(no_tune=True, n_ranks=3, band_rows=16, rank=0)
    rows = hip.rank_rows(p)
    assert L.rt_hip_render_device(C.byref(p), C.c_void_p(buf.data_ptr()), None, None) == 0
    torch.cuda.synchronize()
    ys = R.rank_row_map(120, 16, 3, 0)
    assert np.array_equal(buf.cpu().numpy()[:rows], want[ys])
