"""The C multi-GPU path (rt_hip_multi_*): band arithmetic on the CPU, placement kernel and the whole call on the GPU.

A one-GPU box cannot run a real exchange, so the pieces are pinned separately: (1) where each rank's rows sit in
the gathered buffer (CPU), (2) the placement kernel on bands that ranks 0..n-1 rendered one after the other on the
one device into exactly those places (everything rt_hip_render_multi does except the transport), (3)
rt_hip_render_multi with one device: RCCL is loaded, a communicator made, the same code path runs, and the bytes
are rt_hip_render's."""
import numpy as np
import pytest

import raytracer_amd as R


@pytest.mark.parametrize("n", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("height,band", [(1080, 16), (45, 4), (45, 1), (7, 16), (300, 8)])
def test_gathered_layout_is_a_permutation_of_the_rows(n, height, band):
    """gather_offset + rank_row_map cover every image row exactly once, rank after rank, and the closed form the
    placement kernel uses for 'rows before this rank' equals the host's sum."""
    width = 5
    row_bytes = width * 3
    seen = np.zeros(height, dtype=np.int64)
    total = 0
    for r in range(n):
        rows = R.rank_row_map(height, band, n, r)
        off = R.gather_offset(width, height, band, n, r)
        assert off == total * row_bytes
        stride, full, rem = band * n, height // (band * n), height % (band * n)
        assert off // row_bytes == full * band * r + min(rem, r * band)  # the kernel's expression
        seen[rows] += 1
        # the kernel's mapping from y back to (rank, packed row)
        for i, y in enumerate(rows):
            b = y // band
            assert b % n == r and (b // n) * band + (y - b * band) == i
        total += len(rows)
    assert total == height and (seen == 1).all()
    assert R.gather_offset(width, height, band, n, n) == height * row_bytes


@pytest.mark.gpu
@pytest.mark.parametrize("n,band", [(2, 16), (3, 4), (4, 16), (8, 16), (8, 1)])
def test_bands_of_n_ranks_placed_on_the_device_equal_the_whole_frame(hip, n, band):
    import torch

    W, H = 333, 217
    hip.upload(R.stock_scene("cube_spheres", W, H, tri_reflective=0.3, sph_reflective=0.3))
    full, st = hip.render(count_rays=True)
    gathered = torch.full((H * W * 3,), 0xEE, dtype=torch.uint8, device="cuda:0")
    out = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda:0")
    rays = 0
    for r in range(n):
        p = R.make_params(count_rays=True, band_rows=band, n_ranks=n, rank=r)
        s = R._ffi.Stats()
        hip.render_device(gathered.data_ptr() + R.gather_offset(W, H, band, n, r), p, stats=s)
        rays += s.rays
    hip.place_bands_device(gathered.data_ptr(), out.data_ptr(), W, H, band, n)
    torch.cuda.synchronize()
    assert rays == st.rays
    assert np.array_equal(out.cpu().numpy(), full)


@pytest.mark.gpu
def test_render_multi_with_one_device_runs_the_rccl_path(hip):
    W, H = 640, 360
    scene = R.stock_scene("cube", W, H)
    hip.upload(scene)
    want, st = hip.render(count_rays=True)
    m = R.MultiRenderer(1)  # loads RCCL, makes the communicator; keeps device 0's context
    assert m.n_devices == 1
    m.upload(scene)
    got, ms = m.render(count_rays=True)
    assert np.array_equal(got, want)
    assert ms.rays == st.rays and ms.rows_rendered == H
    got2, _ = m.render(band_rows=4)
    assert np.array_equal(got2, want)
    with pytest.raises(R.RtError):
        m.render(n_ranks=2, band_rows=4, rank=0)  # the library deals the bands out itself


@pytest.mark.parametrize("n", [2, 4, 8])
@pytest.mark.parametrize("chunks", [1, 2, 3, 4])
@pytest.mark.parametrize("height,band", [(1080, 16), (4320, 16), (217, 4), (100, 1)])
def test_parts_of_a_chunked_gather_are_strided_runs_of_rows(n, chunks, height, band):
    """rt_hip_render_multi deals a device's rows out into `chunks` parts: part c of device d is rank c * n + d of a
    round-robin over n * chunks bands.  The rows of part c (over all devices) are then runs of n * band rows every
    n * chunks * band rows, starting at row c * n * band - what the per-part copy to the host relies on - and every
    device renders, in all its parts together, exactly the rows it would render without parts."""
    V = n * chunks
    rows_of_part = [set() for _ in range(chunks)]
    for d in range(n):
        mine = set()
        for c in range(chunks):
            ys = set(int(y) for y in R.rank_row_map(height, band, V, c * n + d))
            assert not (mine & ys)
            mine |= ys
            rows_of_part[c] |= ys
        assert mine == set(int(y) for y in R.rank_row_map(height, band, n, d))
    for c in range(chunks):
        run, period, first = n * band, V * band, c * n * band
        want = set()
        start = first
        while start < height:
            want |= set(range(start, min(start + run, height)))
            start += period
        assert rows_of_part[c] == want
    assert set().union(*rows_of_part) == set(range(height))


def test_gathered_bytes_written_to_the_canvas_convert_back_to_the_same_bytes():
    """The reference-side binding stores a gathered 8-bit value v into canvas[][] as (v + 0.5) / 255
    (rt_reference_binding.c: rt_binding_store_rgb); write_canvas_to_file's color_convert (RayTracer.c:1322:
    (int)(x * 255), clamped) must give v back, for all 256 values."""
    v = np.arange(256, dtype=np.float64)
    x = (v + 0.5) / 255.0
    back = np.clip((x * 255.0).astype(np.int64), 0, 255)  # the cast truncates, as in C
    assert np.array_equal(back, v.astype(np.int64))


@pytest.mark.gpu
@pytest.mark.parametrize("chunks", [2, 3, 4])
def test_render_multi_in_parts_on_one_device(hip, chunks):
    """One device, its rows in several parts: the launches per part, the per-part placement and the strided copies to
    the host are what a group of devices runs too; only the transport between devices is missing."""
    W, H = 640, 360
    scene = R.stock_scene("lighthouse", W, H, tri_reflective=0.3)
    hip.upload(scene)
    want, st = hip.render(count_rays=True)
    m = R.MultiRenderer(1)
    m.upload(scene)
    try:
        m.set_chunks(chunks)
        for band in (16, 4, 7):
            got, ms = m.render(band_rows=band)
            assert np.array_equal(got, want), (chunks, band)
            assert ms.rows_rendered == H
        got, ms = m.render(count_rays=True)  # the counting build renders in one part
        assert np.array_equal(got, want) and ms.rays == st.rays
        got, _ = m.render(no_tune=True, stats=False)
        assert np.array_equal(got, want)
    finally:
        m.set_chunks(0)
    with pytest.raises(R.RtError):
        m.set_chunks(5)


@pytest.mark.gpu
def test_render_multi_over_two_or_more_devices():
    """The exchange itself (grouped RCCL send / receive into device 0): runs where a node with several GPUs is
    available, skipped on a one-GPU box."""
    L = R._ffi.hip_lib()
    n = L.rt_hip_device_count()
    if n < 2:
        pytest.skip("one GPU: the transport between devices cannot run here")
    W, H = 640, 360
    scene = R.stock_scene("cube_spheres", W, H, tri_reflective=0.3, sph_reflective=0.3)
    one = R.HipRenderer(0)
    one.upload(scene)
    want, st = one.render(count_rays=True)
    m = R.MultiRenderer(min(n, 8))
    m.upload(scene)
    for chunks in (1, 2):
        m.set_chunks(chunks)
        got, ms = m.render(count_rays=True)
        assert np.array_equal(got, want)
        assert ms.rays == st.rays
        got, _ = m.render(no_tune=True)
        assert np.array_equal(got, want)
    m.set_chunks(0)
