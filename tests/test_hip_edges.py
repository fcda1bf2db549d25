"""Image and band shapes at the edges of the launch geometry: single pixels, sizes that are not multiples of the
16 x 16 tile, more ranks than rows, sub-pixel grids on tiny images.  Every case against the CPU oracle."""
import numpy as np
import pytest

import oracle_lib as O
import raytracer_amd as R

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scene", ["cube", "spheres", "cube_spheres", "lighthouse", "refraction"])
@pytest.mark.parametrize("w,h", [(1, 1), (17, 1), (1, 33), (31, 15), (65, 47)])
def test_odd_image_sizes(hip, scene, w, h):
    s = R.stock_scene(scene, w, h, 0.3, 0.3)
    hip.upload(s)
    got, st = hip.render(count_rays=True)
    want, ost = O.render_rgb(s)
    assert got.shape == (h, w, 3) and np.array_equal(got, want)
    assert st.rays == ost.rays and st.rays_primary == w * h
    if s.n_tris and not s.n_shapes:
        brute, _ = hip.render(accel=R._ffi.RT_ACCEL_BRUTE)
        assert np.array_equal(brute, want)


@pytest.mark.parametrize("scene", ["cube_spheres", "world_with_plane"])
def test_more_ranks_than_rows_and_tiny_supersampled_images(hip, scene):
    s = R.stock_scene(scene, 9, 5, 0.3, 0.3)
    hip.upload(s)
    full, _ = hip.render(spp_grid=3)
    want, _ = O.render_rgb(s, spp_grid=3)
    assert np.array_equal(full, want)
    out = np.zeros_like(full)
    rows = 0
    for rank in range(8):  # 5 rows in bands of 1 over 8 ranks: ranks 5..7 have nothing to do
        part, st = hip.render(spp_grid=3, n_ranks=8, band_rows=1, rank=rank)
        ys = R.rank_row_map(5, 1, 8, rank)
        assert part.shape[0] == len(ys) == st.rows_rendered
        out[ys] = part
        rows += len(ys)
    assert rows == 5 and np.array_equal(out, full)
