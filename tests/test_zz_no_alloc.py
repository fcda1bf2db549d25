"""rt_hip_format_ppm_device_noalloc, the entry point for callers that cannot allocate: its text and length equal
rt_hip_format_ppm_device's, byte for byte, on two calls in a row (the second on a smaller image: the fixed scratch
keeps the first call's contents), on another stream, and on an empty image; sizes beyond the scratch are refused."""
import numpy as np
import pytest
import torch

import raytracer_amd as R

pytestmark = pytest.mark.gpu


def both_texts(hip, img, stream=None):
    h, w, _ = img.shape
    rgb = torch.from_numpy(np.ascontiguousarray(img)).to("cuda:0") if img.size else torch.zeros(16, dtype=torch.uint8, device="cuda:0")
    cap = hip.ppm_capacity(w, h)
    want_text = torch.zeros(cap, dtype=torch.uint8, device="cuda:0")
    got_text = torch.full((cap,), 0xAA, dtype=torch.uint8, device="cuda:0")
    length = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    n = hip.format_ppm_device(rgb.data_ptr(), w, h, want_text.data_ptr(), cap)
    hip.format_ppm_device_noalloc(rgb.data_ptr(), w, h, got_text.data_ptr(), cap, length.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    m = int(length.cpu()[0])
    return want_text[:n].cpu().numpy().tobytes(), got_text[:m].cpu().numpy().tobytes(), got_text[m:m + 64].cpu().numpy()


def test_two_calls_in_a_row_second_smaller(hip):
    rng = np.random.default_rng(77)
    big = rng.integers(0, 256, size=(1080, 1920, 3), dtype=np.uint8)
    small = rng.integers(0, 256, size=(33, 1021, 3), dtype=np.uint8)
    for img in (big, small):
        want, got, after = both_texts(hip, img)
        assert got == want
        assert want == R.ppm_bytes(img)
        assert (after == 0xAA).all()  # nothing written past the text


def test_other_stream_and_empty_image(hip):
    rng = np.random.default_rng(78)
    img = rng.integers(0, 256, size=(300, 300, 3), dtype=np.uint8)
    s = torch.cuda.Stream()
    want, got, _ = both_texts(hip, img, stream=s.cuda_stream)
    assert got == want
    want, got, _ = both_texts(hip, np.zeros((0, 0, 3), dtype=np.uint8))
    assert got == want == b"P3\n0 0\n255\n"


def test_refuses_more_pixels_than_the_scratch_holds(hip):
    t = torch.zeros(64, dtype=torch.uint8, device="cuda:0")
    length = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    w, h = 8192, 8193  # one row more than 67,108,864 pixels
    with pytest.raises(R.RtError) as e:
        hip.format_ppm_device_noalloc(t.data_ptr(), w, h, t.data_ptr(), hip.ppm_capacity(w, h), length.data_ptr())
    assert e.value.status == R._ffi.RT_ERR_TOO_LARGE
    with pytest.raises(R.RtError):  # length word not 8-byte aligned
        hip.format_ppm_device_noalloc(t.data_ptr(), 2, 2, t.data_ptr(), hip.ppm_capacity(2, 2), length.data_ptr() + 4)
