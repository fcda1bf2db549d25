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


def _box_triangles(lo, hi):
    """The 12 triangles of an axis-aligned box."""
    x0, y0, z0 = lo
    x1, y1, z1 = hi
    v = [(x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0), (x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1)]
    quads = [(0, 1, 2, 3), (5, 4, 7, 6), (4, 0, 3, 7), (1, 5, 6, 2), (3, 2, 6, 7), (4, 5, 1, 0)]
    tris = []
    for a, b, c, d in quads:
        tris += [v[a] + v[b] + v[c], v[a] + v[c] + v[d]]
    return np.array(tris, dtype=np.float32)


@pytest.mark.parametrize("eye,to", [((0.0, 0.0, -6.0), (0.0, 0.0, 0.0)), ((0.0, 5.0, 0.0), (0.0, 0.0, 0.0)),
                                    ((-7.0, 0.5, 0.0), (0.0, 0.5, 0.0))])
def test_rays_parallel_to_the_axes_through_the_bvh(hip, eye, to):
    """A camera on an axis looking along it at axis-aligned boxes, odd image size: the middle column and row have
    direction components of exactly zero, mirror bounces keep them, and coordinates of exactly zero abound.  The
    walk's single-precision slab test (fused lo/d - o/d with a clamped reciprocal) must not lose a box."""
    from raytracer_amd.scene import SceneBuilder

    b = SceneBuilder()
    up = (0.0, 0.0, 1.0) if eye[1] != 0.0 and eye[0] == 0.0 and eye[2] == 0.0 else (0.0, 1.0, 0.0)
    b.set_camera(33, 33, 0.9, eye, to, up)
    b.set_light((-4.0, 6.0, -5.0), (1, 1, 1))
    tris = np.concatenate([_box_triangles((-1, -1, -1), (1, 1, 1)), _box_triangles((0.125, 0.0625, 1.25), (2, 2, 3)),
                           _box_triangles((-3, -2, -0.5), (-1.5, 0.375, 0.5)), _box_triangles((-0.25, -3, -0.25), (0.25, 3, 0.25))])
    b.add_triangles_f32(tris)
    b.set_reflective(0.5, 0.0)
    s = b.finish(1)  # one triangle per leaf: every hit goes through several levels of boxes
    assert s.n_bvh_nodes > 48
    hip.upload(s)
    tree, st = hip.render(accel=R._ffi.RT_ACCEL_BVH, count_rays=True)
    brute, sb = hip.render(accel=R._ffi.RT_ACCEL_BRUTE, count_rays=True)
    want, ost = O.render_rgb(s)
    # no ray origin exactly in the plane of a triangle it is tested against: the list quirk of RayTracer.c:168
    # (an entry with t == 0.0 is overwritten by the next one) is outside what the mesh kernels reproduce
    assert O.probe().exact_zero_t == 0 and O.probe().list_overflow == 0
    assert np.array_equal(tree, brute) and np.array_equal(tree, want)
    assert st.rays == sb.rays == ost.rays and st.rays_secondary > 0
    hip.build_bvh_device(1)
    dev, sd = hip.render(accel=R._ffi.RT_ACCEL_BVH, count_rays=True)
    assert np.array_equal(dev, want) and sd.rays == ost.rays


def test_mirror_spheres_far_from_the_mesh(hip):
    """Mirror and shadow rays that start on a sphere hundreds of units away are walked through the mesh's tree: the
    padding of its single-precision boxes has to scale with where those rays start, not only with the mesh."""
    from raytracer_amd.scene import SceneBuilder
    from test_hip_parity import icosphere

    b = SceneBuilder()
    b.set_camera(96, 64, 0.5, (700.0, 40.0, -900.0), (250.0, 0.0, 0.0))
    b.set_light((-300.0, 900.0, -700.0), (1, 1, 1))
    b.add_triangles_f32(icosphere(3) * np.float32(60.0))
    b.add_sphere(R.Mat4.mul(R.Mat4.translate(500.0, 0.0, 0.0), R.Mat4.scale(200.0, 200.0, 200.0)), R.default_material(reflective=0.9))
    b.add_sphere(R.Mat4.mul(R.Mat4.translate(150.0, -400.0, 100.0), R.Mat4.scale(300.0, 300.0, 300.0)), R.default_material(reflective=0.5))
    b.set_reflective(0.3, 0.8)
    s = b.finish(2)
    hip.upload(s)
    tree, st = hip.render(accel=R._ffi.RT_ACCEL_BVH, count_rays=True)
    brute, sb = hip.render(accel=R._ffi.RT_ACCEL_BRUTE, count_rays=True)
    want, ost = O.render_rgb(s)
    assert O.probe().exact_zero_t == 0 and O.probe().list_overflow == 0
    assert st.rays_secondary > 500 and tree.max() > 0
    assert np.array_equal(tree, brute) and np.array_equal(tree, want) and st.rays == sb.rays == ost.rays
    hip.build_bvh_device(2)
    dev, sd = hip.render(accel=R._ffi.RT_ACCEL_BVH, count_rays=True)
    assert np.array_equal(dev, want) and sd.rays == ost.rays
