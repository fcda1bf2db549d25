"""Worlds of at most 64 triangles and 64 spheres are rendered with masks per 16 x 16 tile: the primitives a
camera ray of the tile can hit, and the primitives that can stand between those and the light (built on the host by
conservative rasterisation and plane separation; a sphere counts as the box that holds it).  A mask that misses a triangle changes pixels, so: random
triangle soups and closed boxes, lights in front, behind, inside and in the plane of things, image sizes that
are not multiples of the tile, sub-pixel grids and row bands; every image against the CPU oracle, and at a larger
size against the kernel that keeps the reference's list (no masks, no filter)."""
import numpy as np
import pytest

import oracle_lib as O
import raytracer_amd as R
from raytracer_amd.scene import SceneBuilder
from test_hip_edges import _box_triangles

pytestmark = pytest.mark.gpu

BRUTE, REFERENCE = R._ffi.RT_ACCEL_BRUTE, R._ffi.RT_ACCEL_REFERENCE


def soup(rng, n):
    """n triangles of mixed sizes around the origin, corners on a 1/64 grid (so that shared planes and edges occur)."""
    c = rng.uniform(-2.0, 2.0, size=(n, 1, 3))
    size = rng.choice([0.3, 0.8, 2.5], size=(n, 1, 1))
    v = c + rng.uniform(-1.0, 1.0, size=(n, 3, 3)) * size
    return (np.round(v * 64.0) / 64.0).reshape(n, 9).astype(np.float32)


def build(tris, w, h, eye, light, reflective=0.0):
    b = SceneBuilder()
    b.set_camera(w, h, 0.7, eye, (0.0, 0.0, 0.0))
    b.set_light(light, (1.0, 1.0, 1.0))
    b.add_triangles_f32(tris)
    if reflective:
        b.set_reflective(reflective, 0.0)
    return b.finish(4)


def check(hip, s, big=None):
    hip.upload(s)
    got, st = hip.render(accel=BRUTE, count_rays=True)
    want, ost = O.render_rgb(s)
    assert np.array_equal(got, want)
    assert st.rays == ost.rays
    assert st.zero_t_entries == 0 and st.list_cap_rays == 0


@pytest.mark.parametrize("seed", range(16))
def test_random_soups(hip, seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1, 65))
    tris = soup(rng, n)
    direction = rng.normal(size=3)
    eye = tuple(direction / np.linalg.norm(direction) * rng.uniform(9.0, 14.0))
    kind = seed % 4
    if kind == 0:  # near the eye
        light = tuple(np.array(eye) + rng.uniform(-2.0, 2.0, size=3))
    elif kind == 1:  # behind the soup
        light = tuple(-np.array(eye) * rng.uniform(0.5, 1.5))
    elif kind == 2:  # in the middle of it
        light = tuple(rng.uniform(-1.0, 1.0, size=3))
    else:  # exactly on the grid the corners live on: in the plane of some triangles' corners
        light = tuple(np.round(rng.uniform(-3.0, 3.0, size=3) * 4.0) / 4.0)
    w, h = [(150, 97), (161, 64), (96, 131), (203, 77)][seed % 4]
    check(hip, build(tris, w, h, eye, light, reflective=0.4 if seed % 3 == 0 else 0.0))


@pytest.mark.parametrize("light", [(-4.0, 6.0, -5.0), (0.5, 0.5, 0.5), (1.0, 3.0, 1.0), (0.0, 0.0, 9.0), (1.0, 1.0, -7.0)])
def test_boxes_light_inside_on_a_face_plane_and_behind(hip, light):
    """Closed bodies: neighbours share edges and corners with the triangle a shadow ray starts on; (1, 3, 1) lies in the
    planes of two faces of the first box, (0.5, 0.5, 0.5) inside it."""
    tris = np.concatenate([_box_triangles((-1, -1, -1), (1, 1, 1)), _box_triangles((1.5, -1, -0.5), (2.5, 0.5, 0.5)),
                           _box_triangles((-3, -2, -0.5), (-1.5, 0.375, 0.5)), _box_triangles((-0.25, 1.25, -0.25), (0.25, 3, 0.25))])
    assert len(tris) == 48
    for eye in ((3.0, 4.0, -9.0), (-7.0, 0.5, 2.0)):
        check(hip, build(tris, 211, 117, eye, light, reflective=0.3))


def test_masks_with_sub_pixel_grid_and_row_bands(hip):
    rng = np.random.default_rng(7)
    s = build(soup(rng, 40), 131, 83, (4.0, 3.0, -10.0), (-6.0, 7.0, -4.0))
    hip.upload(s)
    full, _ = hip.render(accel=BRUTE, spp_grid=2)
    want, _ = O.render_rgb(s, spp_grid=2)
    assert np.array_equal(full, want)
    out = np.zeros_like(full)
    for rank in range(3):
        part, st = hip.render(accel=BRUTE, spp_grid=2, n_ranks=3, band_rows=5, rank=rank)
        ys = R.rank_row_map(83, 5, 3, rank)
        assert part.shape[0] == len(ys) == st.rows_rendered
        out[ys] = part
    assert np.array_equal(out, full)


@pytest.mark.parametrize("seed", range(4))
def test_masked_kernel_equals_the_list_kernel_at_1080p(hip, seed):
    rng = np.random.default_rng(50 + seed)
    tris = soup(rng, [12, 33, 64, 5][seed])
    s = build(tris, 1920, 1080, (5.0, 4.0, -11.0), [(-8.0, 9.0, -9.0), (0.0, 0.0, 0.0), (3.0, -6.0, 8.0), (5.0, 4.5, -10.0)][seed],
              reflective=0.3 if seed == 1 else 0.0)
    hip.upload(s)
    ref, rst = hip.render(accel=REFERENCE, count_rays=True)
    got, st = hip.render(accel=BRUTE, count_rays=True)
    assert st.zero_t_entries == 0 and st.list_cap_rays == 0
    assert np.array_equal(got, ref) and st.rays == rst.rays
    if seed == 0:  # the masks are in use: without them the camera rays near the soup alone test all 12 triangles each
        assert st.prim_tests < 6 * st.rays_shadow


@pytest.mark.parametrize("seed", range(8))
def test_spheres_and_triangles_shadowing_each_other(hip, seed):
    """Mixed worlds: boxes and loose triangles between spheres of all shapes, among them a floor-like flattened sphere
    that reaches behind the camera (no pixel rectangle) and, in some, an eye inside a sphere's box (no masks for its
    tiles).  Lights between, above and inside the bodies."""
    from raytracer_amd.scene import Mat4, default_material

    rng = np.random.default_rng(300 + seed)
    b = SceneBuilder()
    eye = [(0.0, 1.5, -7.0), (4.0, 3.0, -6.0), (0.0, 0.2, -1.2), (-5.0, 2.0, 3.0)][seed % 4]
    b.set_camera([150, 203][seed % 2], [97, 77][seed % 2], 0.9, eye, (0.0, 0.3, 0.0))
    b.set_light([(-6.0, 9.0, -7.0), (0.0, 0.6, 0.0), (3.0, 0.4, -1.0), (0.0, 12.0, 0.0)][(seed // 2) % 4], (1.0, 1.0, 1.0))
    b.add_sphere(Mat4.mul(Mat4.translate(0.0, -1.0, 0.0), Mat4.scale(12.0, 0.02, 12.0)), default_material(color=(0.9, 0.8, 0.7), reflective=0.3 if seed % 3 == 0 else 0.0))
    for k in range(int(rng.integers(2, 9))):
        t = Mat4.mul(Mat4.translate(*rng.uniform(-2.5, 2.5, 3)), Mat4.mul(Mat4.rotate_z(rng.uniform(0, 3)), Mat4.scale(*rng.uniform(0.2, 1.4, 3))))
        b.add_sphere(t, default_material(color=rng.uniform(0.1, 1.0, 3), shininess=200.0))
    tris = [soup(rng, int(rng.integers(1, 20)))]
    for k in range(int(rng.integers(0, 3))):
        lo = rng.uniform(-2.5, 1.5, 3)
        tris.append(_box_triangles(tuple(lo), tuple(lo + rng.uniform(0.3, 1.2, 3))))
    b.add_triangles_f32(np.concatenate(tris))
    s = b.finish(4)
    assert s.n_tris <= 64 and s.n_spheres >= 3
    hip.upload(s)
    got, st = hip.render_f64(accel=BRUTE, count_rays=True)
    want, ost = O.render_f64(s)
    assert st.rays == ost.rays
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-15)  # the last bits of pow() may differ, a wrong hit would not be that small
    full, _ = hip.render(accel=BRUTE)
    ref, _ = hip.render(accel=REFERENCE)
    assert np.array_equal(full, ref)
