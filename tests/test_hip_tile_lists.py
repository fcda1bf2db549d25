"""Meshes of more than 64 triangles that are walked exhaustively get, per 16 x 16 tile, a list of the triangles a
camera ray of the tile can hit (the same conservative rasterisation that makes the masks of smaller meshes) with
copies of their single-precision forms.  A list that misses a triangle changes pixels, so: random soups and closed
boxes of 65 .. 700 triangles against the CPU oracle, image sizes that are not multiples of the tile, sub-pixel grids
and row bands, mirrors (whose rays must not use the lists), the reference's lighthouse at 1080p against the walk
through the tree, and scenes whose triangle order the device tree build has changed."""
import numpy as np
import pytest

import oracle_lib as O
import raytracer_amd as R
from test_hip_edges import _box_triangles
from test_hip_tile_masks import build, soup

pytestmark = pytest.mark.gpu

BRUTE, BVH, REFERENCE = R._ffi.RT_ACCEL_BRUTE, R._ffi.RT_ACCEL_BVH, R._ffi.RT_ACCEL_REFERENCE


def check(hip, s, **kw):
    hip.upload(s)
    got, st = hip.render(accel=BRUTE, count_rays=True, **kw)
    want, ost = O.render_rgb(s, **kw)
    assert np.array_equal(got, want)
    assert st.rays == ost.rays
    if s.n_bvh_nodes > 0:  # the walk, whose camera rays run down the same lists
        walked, wst = hip.render(accel=BVH, count_rays=True, **kw)
        assert np.array_equal(walked, want) and wst.rays == ost.rays
    return st


@pytest.mark.parametrize("seed", range(10))
def test_random_soups_beyond_64_triangles(hip, seed):
    rng = np.random.default_rng(4000 + seed)
    n = int([65, 66, 127, 128, 129, 300, 513, 700, 97, 211][seed])
    tris = soup(rng, n)
    direction = rng.normal(size=3)
    eye = tuple(direction / np.linalg.norm(direction) * rng.uniform(9.0, 14.0))
    light = [tuple(np.array(eye) + rng.uniform(-2.0, 2.0, size=3)), tuple(-np.array(eye) * rng.uniform(0.5, 1.5)),
             tuple(rng.uniform(-1.0, 1.0, size=3)), tuple(np.round(rng.uniform(-3.0, 3.0, size=3) * 4.0) / 4.0)][seed % 4]
    w, h = [(150, 97), (161, 64), (96, 131), (203, 77)][seed % 4]
    st = check(hip, build(tris, w, h, eye, light, reflective=0.4 if seed % 3 == 0 else 0.0))
    if seed % 3 != 0:  # the lists are in use: without them every camera ray near the mesh tests every triangle
        assert st.prim_tests < (st.rays_primary // 2 + st.rays_shadow) * n


def test_boxes_sharing_edges_with_sub_pixel_grid_and_row_bands(hip):
    rng = np.random.default_rng(11)
    boxes = []
    for k in range(9):
        lo = np.round(rng.uniform(-2.5, 1.5, 3) * 8.0) / 8.0
        boxes.append(_box_triangles(tuple(lo), tuple(lo + np.round(rng.uniform(0.25, 1.5, 3) * 8.0) / 8.0)))
    tris = np.concatenate(boxes)
    assert len(tris) == 108
    s = build(tris, 131, 83, (4.0, 3.0, -10.0), (-6.0, 7.0, -4.0))
    check(hip, s, spp_grid=2)
    full, _ = hip.render(accel=BRUTE, spp_grid=2)
    out = np.zeros_like(full)
    for rank in range(3):
        part, st = hip.render(accel=BRUTE, spp_grid=2, n_ranks=3, band_rows=5, rank=rank, count_rays=True)
        ys = R.rank_row_map(83, 5, 3, rank)
        assert part.shape[0] == len(ys) == st.rows_rendered
        out[ys] = part
    assert np.array_equal(out, full)


def test_lighthouse_1080p_lists_equal_the_walk_and_skip_most_tests(hip):
    s = R.stock_scene("lighthouse", 1920, 1080)
    hip.upload(s)
    walked, wst = hip.render(accel=BVH, count_rays=True)
    got, st = hip.render(accel=BRUTE, count_rays=True)
    assert np.array_equal(got, walked) and st.rays == wst.rays
    # every shadow ray still tests the whole mesh (minus what it skips after its first blocker); the camera rays no longer do
    assert st.prim_tests < 1.3 * st.rays_shadow * s.n_tris


def test_lighthouse_reflective_lists_equal_the_walk(hip):
    s = R.stock_scene("lighthouse", 640, 360, 0.3, 0.0)
    hip.upload(s)
    walked, wst = hip.render(accel=BVH, count_rays=True)
    got, st = hip.render(accel=BRUTE, count_rays=True)
    assert np.array_equal(got, walked) and st.rays == wst.rays and st.rays_secondary > 0


@pytest.mark.parametrize("n", [40, 200])
def test_exhaustive_loop_after_the_device_tree_build_reordered_the_triangles(hip, n):
    """rt_hip_build_bvh_device sorts the triangles along a space-filling curve: masks and lists made from the order of
    the upload would name the wrong triangles."""
    rng = np.random.default_rng(77 + n)
    s = build(soup(rng, n), 203, 131, (5.0, 4.0, -11.0), (-8.0, 9.0, -9.0))
    hip.upload(s)
    before, bst = hip.render(accel=BRUTE, count_rays=True)
    hip.build_bvh_device(4)
    after, ast = hip.render(accel=BRUTE, count_rays=True)
    walked, _ = hip.render(accel=BVH)
    want, ost = O.render_rgb(s)
    assert np.array_equal(before, want) and np.array_equal(after, want) and np.array_equal(walked, want)
    assert bst.rays == ast.rays == ost.rays


def test_counting_build_still_sees_every_entry_of_a_listed_camera_ray(hip):
    """104 large triangles stacked behind each other: a camera ray through the middle appends 104 entries, of which
    the reference keeps 100 (RayTracer.c:161).  The plain kernel leaves the 103 behind the first to the
    single-precision test; the counting kernel evaluates all of them and reports the rays that went over the cap."""
    from raytracer_amd.scene import SceneBuilder

    tris = np.array([[-3, -3, z, 3, -3, z, 0, 4, z] for z in np.arange(104) * 0.125], dtype=np.float32)
    b = SceneBuilder()
    b.set_camera(64, 48, 0.6, (0.0, 0.0, -10.0), (0.0, 0.0, 0.0))
    b.set_light((-30.0, 0.0, -0.5), (1.0, 1.0, 1.0))  # the lines of the shadow rays pass beside the triangles behind
    b.add_triangles_f32(tris)
    s = b.finish(4)
    hip.upload(s)
    O.render_rgb(s)
    assert O.probe().list_overflow > 0
    got, st = hip.render(accel=BRUTE, count_rays=True)
    assert st.list_cap_rays > 0 and st.zero_t_entries == 0
    assert st.tri_exact < st.rays_primary * 2  # ... while it counts the double-precision tests of the plain kernel
    ref, _ = hip.render(accel=REFERENCE)
    want, _ = O.render_rgb(s)
    assert np.array_equal(ref, want)


@pytest.mark.parametrize("level,w,h,reflective", [(3, 203, 131, 0.0), (4, 171, 99, 0.25), (5, 160, 120, 0.0)])
def test_tessellated_spheres_with_many_small_triangles_per_tile(hip, level, w, h, reflective):
    """1280, 5120 and 20,480 triangles on a sphere: tiles with hundreds of small triangles in their lists (front and
    back of the sphere), lists longer than what a workgroup keeps in LDS, shadow rays that graze the silhouette."""
    from test_hip_parity import icosphere, mesh_scene

    s = mesh_scene(icosphere(level), w, h, reflective=reflective)
    hip.upload(s)
    want, ost = O.render_rgb(s)
    walked, wst = hip.render(accel=BVH, count_rays=True)
    assert np.array_equal(walked, want) and wst.rays == ost.rays
    if level < 5:  # the exhaustive loop over 20,480 triangles is left to the smaller meshes
        got, st = hip.render(accel=BRUTE, count_rays=True)
        assert np.array_equal(got, want) and st.rays == ost.rays
        assert st.prim_tests < st.rays * s.n_tris // 2
