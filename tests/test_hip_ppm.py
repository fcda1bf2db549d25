"""The P3 text of write_canvas_to_file (RayTracer.c:1333-1348) formatted on the GPU: byte for byte the text of the
host C writer rt_format_ppm (which the CPU tests pin to the reference's own canvas.ppm files)."""
import hashlib

import numpy as np
import pytest
import torch

import raytracer_amd as R
from conftest import load_golden  # noqa: F401  (fixture module)
from test_hip_parity import scene_for

pytestmark = pytest.mark.gpu


def device_text(hip, img):
    t = torch.from_numpy(np.ascontiguousarray(img)).to("cuda:0")
    h, w, _ = img.shape
    out = hip.format_ppm(t.data_ptr(), w, h)
    torch.cuda.synchronize()
    return out


SHAPES = [(1, 1), (1, 7), (3, 5), (16, 128), (1, 2048), (1, 2049), (2047, 1), (300, 300), (33, 1021), (1080, 1920)]


@pytest.mark.parametrize("h,w", SHAPES)
def test_random_images_match_the_host_writer(hip, h, w):
    rng = np.random.default_rng(h * 100003 + w)
    img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    assert device_text(hip, img) == R.ppm_bytes(img)


@pytest.mark.parametrize("fill", [0, 9, 10, 99, 100, 255])
def test_constant_images_every_digit_count(hip, fill):
    img = np.full((37, 211, 3), fill, dtype=np.uint8)
    want = R.ppm_bytes(img)
    assert device_text(hip, img) == want
    per_px = 3 * (len(str(fill)) + 1) + 1
    assert len(want) == len(b"P3\n211 37\n255\n") + per_px * 37 * 211


def test_digit_boundaries_mixed(hip):
    vals = np.array([0, 9, 10, 99, 100, 255, 1, 200], dtype=np.uint8)
    img = vals[np.arange(5000 * 3) % len(vals)].reshape(50, 100, 3)
    assert device_text(hip, img) == R.ppm_bytes(img)


def test_empty_image_is_the_header_alone(hip):
    t = torch.zeros(16, dtype=torch.uint8, device="cuda:0")
    assert hip.format_ppm(t.data_ptr(), 0, 0) == b"P3\n0 0\n255\n"


def test_async_call_and_length_query(hip):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, size=(123, 457, 3), dtype=np.uint8)
    t = torch.from_numpy(img).to("cuda:0")
    cap = hip.ppm_capacity(457, 123)
    text = torch.zeros(cap, dtype=torch.uint8, device="cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    assert hip.format_ppm_device(t.data_ptr(), 457, 123, text.data_ptr(), cap, stream=s, wait=False) is None
    n = hip.ppm_last_length()
    want = R.ppm_bytes(img)
    assert n == len(want)
    assert text[:n].cpu().numpy().tobytes() == want


def test_rejects_bad_arguments(hip):
    t = torch.zeros(64, dtype=torch.uint8, device="cuda:0")
    text = torch.zeros(4096, dtype=torch.uint8, device="cuda:0")
    with pytest.raises(R.RtError):  # capacity below the bound
        hip.format_ppm_device(t.data_ptr(), 4, 4, text.data_ptr(), 10)
    with pytest.raises(R.RtError):  # text buffer not 16-byte aligned
        hip.format_ppm_device(t.data_ptr(), 4, 4, text.data_ptr() + 4, 4000)
    with pytest.raises(R.RtError):  # image not 4-byte aligned
        hip.format_ppm_device(t.data_ptr() + 1, 4, 4, text.data_ptr(), 4096)


@pytest.mark.parametrize("name", ["lighthouse_300", "cube_300", "spheres_300"])
def test_render_ppm_hashes_to_the_reference_file(hip, manifest, name):
    """render + format on the device = the bytes of the canvas.ppm the reference binary writes."""
    e = manifest[name]
    hip.upload(scene_for(e))
    text, st = hip.render_ppm(count_rays=True)
    assert st.rays == e["rays"]
    assert hashlib.sha256(text).hexdigest() == e["ppm_sha256"]


def test_8k_frame_text(hip):
    """The largest benchmark frame (7680x4320): length from the digit counts, content against the host writer."""
    h, w = 4320, 7680
    rng = np.random.default_rng(8)
    img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    digits = 1 + (img >= 10).astype(np.int64) + (img >= 100)
    want_len = len(b"P3\n7680 4320\n255\n") + int(digits.sum()) + 4 * h * w
    got = device_text(hip, img)
    assert len(got) == want_len
    assert hashlib.sha256(got).digest() == hashlib.sha256(R.ppm_bytes(img)).digest()


def test_pipelined_frames_are_the_synchronous_bytes(hip, manifest):
    """Two frames in flight (submit f + 1 before collecting f): every frame's text equals rt_hip_render_ppm's, also
    when the scene changes between submissions and when a slot is misused."""
    e = manifest["cube_300"]
    hip.upload(scene_for(e))
    want, _ = hip.render_ppm()
    assert hashlib.sha256(want).hexdigest() == e["ppm_sha256"]
    with pytest.raises(R.RtError):
        hip.render_ppm_collect(0)  # nothing submitted
    hip.render_ppm_submit(0)
    with pytest.raises(R.RtError):
        hip.render_ppm_submit(0)  # still holds a frame
    for f in range(1, 6):
        hip.render_ppm_submit(f % 2)
        got, st = hip.render_ppm_collect((f - 1) % 2)
        assert got == want and st.kernel_ms > 0
    got, _ = hip.render_ppm_collect(5 % 2)
    assert got == want
    # another scene through the same slots
    e2 = manifest["spheres_300"]
    hip.upload(scene_for(e2))
    hip.render_ppm_submit(0)
    got, _ = hip.render_ppm_collect(0)
    assert hashlib.sha256(got).hexdigest() == e2["ppm_sha256"]
