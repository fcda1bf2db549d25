"""Worlds beyond spheres and triangles on the GPU: planes, cube, cylinder, patterns, refraction (RayTracer.c:799-918,
1166-1271, 1378-1406, 1453-1499).  Checked against canvases and images the REFERENCE's own scene functions produced
(render_refraction_scene, render_complete_world_with_plane, render_dual_spheres_refracting_on_floor) and, for seeded
worlds, against the CPU oracle.  Tolerance: the double canvas within 1e-12 relative (pow() is the only operation
whose last bits may differ), 8-bit images >= 99.9 % identical pixels and no channel off by more than 1."""
import hashlib
import math

import numpy as np
import pytest

import oracle_lib as O
import raytracer_amd as R
from conftest import load_golden, quantise
from raytracer_amd.scene import Mat4, SceneBuilder, default_material, material_ext
from test_hip_parity import assert_image_parity, scene_for

pytestmark = pytest.mark.gpu

CANVAS = ["refraction_96x64", "world_with_plane_96x64", "dual_spheres_96x64", "refraction_ss2_48x32", "world_with_plane_77x45"]


@pytest.mark.parametrize("name", CANVAS)
def test_canvas_matches_reference_canvas(hip, manifest, name):
    e = manifest[name]
    hip.upload(scene_for(e))
    got, st = hip.render_f64(spp_grid=e["spp_grid"], count_rays=True)
    want = load_golden(e, "canvas")
    assert st.rays == e["rays"], "ray count (intersect_world calls)"
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-15)
    ex, npx = assert_image_parity(quantise(got), quantise(want), name)
    assert ex == npx


@pytest.mark.parametrize("name", ["refraction_300", "world_with_plane_300", "dual_spheres_300"])
def test_reference_scene_functions_300(hip, manifest, name):
    """The image the reference binary's own scene function renders; the PPM hashes to its canvas.ppm."""
    e = manifest[name]
    hip.upload(scene_for(e))
    got, st = hip.render(count_rays=True)
    want = load_golden(e, "rgb")
    assert_image_parity(got, want, name)
    assert st.rays == e["rays"]
    if np.array_equal(got, want):
        assert hashlib.sha256(R.ppm_bytes(got)).hexdigest() == e["ppm_sha256"]
        text, _ = hip.render_ppm()
        assert hashlib.sha256(text).hexdigest() == e["ppm_sha256"]


@pytest.mark.parametrize("name", ["refraction_1080p", "world_with_plane_1080p", "dual_spheres_1080p"])
def test_reference_scene_functions_1080p(hip, manifest, name):
    e = manifest[name]
    hip.upload(scene_for(e))
    got, st = hip.render(count_rays=True)
    assert st.rays == e["rays"]
    assert_image_parity(got, load_golden(e, "rgb"), name)


def random_general_world(seed, w, h):
    """Every shape kind, every pattern kind, mirrors and glass, with rotated / scaled transforms."""
    rng = np.random.default_rng(seed)
    b = SceneBuilder()
    b.set_camera(w, h, 0.9, (rng.uniform(-1, 1), rng.uniform(1.0, 2.5), -6.0), (0.0, 0.6, 0.0))
    b.set_light((rng.uniform(-6, 6), rng.uniform(4, 9), rng.uniform(-8, -3)), (1.0, 0.95, 0.9))

    def mat(**kw):
        m = default_material()
        m.color[:] = list(rng.uniform(0.1, 1.0, 3))
        m.diffuse, m.specular, m.shininess = rng.uniform(0.3, 0.9), rng.uniform(0.0, 0.9), float(rng.integers(5, 300))
        for k, v in kw.items():
            setattr(m, k, v)
        return m

    def pat(kind):
        return material_ext(pattern_type=kind, pattern_from=rng.uniform(0, 1, 3), pattern_to=rng.uniform(0, 1, 3))

    def xf():
        t = Mat4.translate(rng.uniform(-2.5, 2.5), rng.uniform(0.2, 1.6), rng.uniform(-1.5, 3.0))
        r = Mat4.mul(Mat4.rotate_y(rng.uniform(0, math.pi)), Mat4.rotate_x(rng.uniform(-0.6, 0.6)))
        s = Mat4.scale(*rng.uniform(0.3, 0.9, 3))
        return Mat4.mul(Mat4.mul(t, r), s)

    P = R._ffi
    b.add_shape("plane", Mat4.identity(), mat(reflective=0.3), pat(P.RT_PATTERN_CHECKER), Mat4.scale(0.7, 0.7, 0.7))
    b.add_shape("plane", Mat4.mul(Mat4.translate(0, 0, 6.0), Mat4.rotate_x(math.pi / 2)), mat(), pat(P.RT_PATTERN_RING))
    b.add_sphere_ext(xf(), mat(), pat(P.RT_PATTERN_STRIPE), Mat4.scale(0.2, 0.2, 0.2))
    b.add_sphere_ext(xf(), mat(), pat(P.RT_PATTERN_GRADIENT), Mat4.rotate_z(0.4))
    b.add_sphere_ext(xf(), mat(), pat(P.RT_PATTERN_TEST))
    b.add_sphere_ext(xf(), mat(diffuse=0.1, reflective=0.9), material_ext(transparency=0.9, refractive_index=1.5))
    b.add_sphere_ext(xf(), mat(diffuse=0.2), material_ext(transparency=0.7, refractive_index=1.2))
    b.add_sphere(xf(), mat(reflective=1.0))
    b.add_shape("cube", xf(), mat(reflective=0.2))
    b.add_shape("cube", xf(), mat(), material_ext(transparency=0.5, refractive_index=1.3))
    b.add_shape("cylinder", xf(), mat(), minimum=-1.0, maximum=1.0, closed=True)
    b.add_shape("cylinder", xf(), mat(reflective=0.5), minimum=0.0, maximum=0.7, closed=False)
    b.add_shape("cylinder", xf(), mat(), pat(P.RT_PATTERN_CHECKER), minimum=-0.5, maximum=2.0, closed=True)
    for _ in range(3):
        c = rng.uniform(-2, 2, 3) + np.array([0, 1.5, 1.0])
        p = (c + rng.uniform(-0.8, 0.8, (3, 3))).astype(np.float32).astype(np.float64)
        b.add_triangle(p[0], p[1], p[2], mat(reflective=0.4))
    p = np.array([[-3, 0.2, 2.5], [3, 0.2, 2.5], [0, 3.5, 2.0]], dtype=np.float32).astype(np.float64)
    b.add_triangle(p[0], p[1], p[2], mat(diffuse=0.3), material_ext(transparency=0.6, refractive_index=1.1))
    b.add_triangle(p[0] + [0, 0, 1], p[2] + [0, 0, 1], p[1] + [0, 0, 1], mat(), pat(P.RT_PATTERN_STRIPE), Mat4.scale(0.3, 0.3, 0.3))
    return b.finish(0)


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_seeded_general_worlds_match_oracle(hip, seed):
    s = random_general_world(seed, 96, 72)
    hip.upload(s)
    got, st = hip.render_f64(count_rays=True)
    want, ost = O.render_f64(s)
    assert O.probe().list_overflow == 0
    assert (st.rays_primary, st.rays_shadow, st.rays_secondary) == (ost.rays_primary, ost.rays_shadow, ost.rays_secondary)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-15)
    assert (want.sum(axis=2) > 0).mean() > 0.5


@pytest.mark.parametrize("seed", [11, 12])
def test_silhouettes_of_stretched_shapes_match_oracle(hip, seed):
    """Spheres, cubes and cylinders scaled by 0.05..2 per axis and rotated: many rays pass close to the edge of a
    shape, where the kernel's single-precision early-out (line against a bounding sphere) has to stay on the safe
    side of the reference's discriminant and slab tests.  Canvases agree to the last bits or the test fails."""
    rng = np.random.default_rng(seed)
    b = SceneBuilder()
    b.set_camera(256, 192, 0.8, (0.5, 1.5, -7.0), (0.0, 0.8, 0.0))
    b.set_light((-5.0, 8.0, -6.0), (1.0, 1.0, 1.0))
    b.add_shape("plane", Mat4.identity(), default_material())
    for k in range(15):
        t = Mat4.translate(rng.uniform(-3, 3), rng.uniform(0.1, 2.5), rng.uniform(-2, 4))
        r = Mat4.mul(Mat4.rotate_y(rng.uniform(0, math.pi)), Mat4.mul(Mat4.rotate_x(rng.uniform(0, math.pi)),
                                                                     Mat4.rotate_z(rng.uniform(0, math.pi))))
        xf = Mat4.mul(Mat4.mul(t, r), Mat4.scale(*rng.uniform(0.05, 2.0, 3)))
        m = default_material()
        m.color[:] = list(rng.uniform(0.2, 1.0, 3))
        m.reflective = float(rng.choice([0.0, 0.5]))
        ext = material_ext(transparency=0.6, refractive_index=1.3) if k % 5 == 0 else None
        if k % 3 == 0:
            b.add_sphere_ext(xf, m, ext)
        elif k % 3 == 1:
            b.add_shape("cube", xf, m, ext)
        else:
            b.add_shape("cylinder", xf, m, ext, minimum=float(rng.uniform(-2, 0)), maximum=float(rng.uniform(0.1, 2)),
                        closed=bool(k % 2))
    s = b.finish(0)
    hip.upload(s)
    got, st = hip.render_f64(count_rays=True)
    want, ost = O.render_f64(s)
    assert O.probe().list_overflow == 0
    assert (st.rays_primary, st.rays_shadow, st.rays_secondary) == (ost.rays_primary, ost.rays_shadow, ost.rays_secondary)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-15)


def test_supersampled_general_world_and_row_bands(hip):
    s = random_general_world(7, 64, 48)
    hip.upload(s)
    full, _ = hip.render(spp_grid=2)
    want, _ = O.render_rgb(s, spp_grid=2)
    assert_image_parity(full, want, "general ss2")
    out = np.zeros_like(full)
    for rank in range(3):
        part, st = hip.render(spp_grid=2, n_ranks=3, band_rows=5, rank=rank)
        ys = [y for y in range(48) if (y // 5) % 3 == rank]
        assert part.shape[0] == len(ys) == st.rows_rendered
        out[ys] = part
    assert np.array_equal(out, full)


def test_long_lists_use_the_100_slot_list(hip):
    """More than 32 possible intersections per ray: the kernel switches to the reference's list size."""
    b = SceneBuilder()
    b.set_camera(48, 32, 0.6, (0.0, 0.5, -9.0), (0.0, 0.0, 0.0))
    b.set_light((3.0, 6.0, -8.0), (1.0, 1.0, 1.0))
    m = default_material()
    for k in range(20):  # 20 nested glass shells: up to 40 entries on one ray
        b.add_sphere_ext(Mat4.scale(*(3 * [0.4 + 0.1 * k])), m, material_ext(transparency=0.3, refractive_index=1.0 + 0.01 * k))
    s = b.finish(0)
    hip.upload(s)
    got, st = hip.render_f64(max_depth=3, count_rays=True)
    want, ost = O.render_f64(s, max_depth=3)
    assert O.probe().max_list == 40
    assert st.rays == ost.rays
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize("scene,w,h,tr,sr", [("cube", 1920, 1080, 0.0, 0.0), ("spheres", 1920, 1080, 0.0, 0.0),
                                              ("cube_spheres", 1280, 720, 0.3, 0.3), ("lighthouse", 960, 540, 0.3, 0.0)])
def test_fast_kernels_equal_the_list_kernel_on_the_gpu(hip, scene, w, h, tr, sr):
    """RT_ACCEL_REFERENCE runs the kernel that keeps the reference's 100-slot list, stable sort and 4-component
    tuples on a sphere / triangle world: the running-minimum kernels (exhaustive and BVH) must give the same bytes and
    ray counts at sizes the CPU reference needs minutes for."""
    hip.upload(R.stock_scene(scene, w, h, tr, sr))
    ref, rst = hip.render(accel=R._ffi.RT_ACCEL_REFERENCE, count_rays=True)
    assert rst.accel_used == R._ffi.RT_ACCEL_REFERENCE
    for accel in (R._ffi.RT_ACCEL_BRUTE, R._ffi.RT_ACCEL_BVH):
        got, st = hip.render(accel=accel, count_rays=True)
        assert st.accel_used != R._ffi.RT_ACCEL_REFERENCE  # reflective materials alone must not select the list kernel
        assert (st.rays_primary, st.rays_shadow, st.rays_secondary) == (rst.rays_primary, rst.rays_shadow, rst.rays_secondary)
        assert_image_parity(got, ref, f"{scene} accel {accel}")
        assert np.array_equal(got, ref)


def test_counting_build_reports_what_a_running_minimum_cannot_reproduce(hip):
    """The sphere / triangle kernels keep a running minimum, not the reference's list.  Two worlds that make the list
    behave unlike a minimum: the counting build reports both, the list kernel (RT_ACCEL_REFERENCE) renders them as the
    reference does, and the ordinary scenes report zero."""
    # an eye exactly in the plane of the first triangle: every camera ray gets an entry with t == 0.0, which the next
    # append overwrites (RayTracer.c:168)
    b = SceneBuilder()
    b.set_camera(32, 24, 0.9, (0.0, 0.0, 0.0), (0.0, 0.0, -1.0))
    b.set_light((-3.0, 4.0, 2.0), (1.0, 1.0, 1.0))
    b.add_triangles_f32(np.array([[-1, -1, 0, 1, -1, 0, 0, 1, 0], [-4, -4, -3, 4, -4, -3, 0, 5, -3]], dtype=np.float32))
    s = b.finish(1)
    hip.upload(s)
    want, _ = O.render_rgb(s)
    for accel in (R._ffi.RT_ACCEL_BRUTE, R._ffi.RT_ACCEL_BVH):
        fast, st = hip.render(accel=accel, count_rays=True)
        assert st.zero_t_entries >= 32 * 24 and st.list_cap_rays == 0
        assert not np.array_equal(fast, want)  # the divergence the counter is there to flag
    ref, rst = hip.render(accel=R._ffi.RT_ACCEL_REFERENCE, count_rays=True)
    assert np.array_equal(ref, want) and rst.zero_t_entries == 0  # the list kernel reproduces, so it has nothing to report

    # 51 nested spheres: 102 entries on a ray through the middle, the reference keeps the first 100 (RayTracer.c:161)
    b = SceneBuilder()
    b.set_camera(40, 30, 0.5, (0.0, 0.0, -12.0), (0.0, 0.0, 0.0))
    b.set_light((-5.0, 8.0, -10.0), (1.0, 1.0, 1.0))
    for k in range(51):
        b.add_sphere(Mat4.scale(*(3 * [0.5 + 0.05 * k])), default_material())
    s = b.finish(0)
    hip.upload(s)
    want, _ = O.render_rgb(s)
    assert O.probe().list_overflow > 0
    _, st = hip.render(accel=R._ffi.RT_ACCEL_BRUTE, count_rays=True)
    assert st.list_cap_rays > 0 and st.zero_t_entries == 0
    ref, _ = hip.render(accel=R._ffi.RT_ACCEL_REFERENCE)
    assert np.array_equal(ref, want)

    for scene in ("cube", "spheres", "cube_spheres", "lighthouse"):
        hip.upload(R.stock_scene(scene, 160, 120, 0.3, 0.3))
        for accel in (R._ffi.RT_ACCEL_BRUTE, R._ffi.RT_ACCEL_BVH):
            _, st = hip.render(accel=accel, count_rays=True)
            assert st.zero_t_entries == 0 and st.list_cap_rays == 0
