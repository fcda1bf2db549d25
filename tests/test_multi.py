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
