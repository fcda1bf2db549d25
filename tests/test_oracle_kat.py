"""The oracle against the reference's OWN unit-test values (known answers).

Each case repeats a test of RayTracer/RayTracer.c (`#if defined _DEBUG`, lines
1710-7044; all 174 pass when the reference is built with -D_DEBUG) on the CPU
restatement in oracle/rt_oracle.c.  Tolerance is the reference's own:
equal(a, b) = |a - b| < 1e-6 (RayTracer.c:262).
"""
import ctypes as C
import math

import numpy as np
import pytest

import oracle_lib as O
import raytracer_amd as R

EPS = 1e-6
S2 = math.sqrt(2.0) / 2.0


def close(a, b):
    return np.allclose(np.asarray(a, dtype=float), np.asarray(b, dtype=float), rtol=0, atol=EPS)


def sphere_ts(transform, origin, direction):
    inv = R.Mat4.inverse(transform)
    t = (C.c_double * 2)()
    n = O.lib().rto_intersect_sphere(O.m16(inv), C.byref(O.ray(origin, direction)), t)
    return n, list(t[:n])


def default_world(w=11, h=11, outer_ambient=None, inner_ambient=None, light=(-10.0, 10.0, -10.0), extra=None):
    """create_default_world, RayTracer.c:1092-1126 (note the 0.8f float literal at :1113)."""
    b = R.SceneBuilder()
    b.set_camera(w, h, math.pi / 2, (0, 0, -5), (0, 0, 0))
    b.set_light(light, (1.0, 1.0, 1.0))
    m1 = R.default_material(color=(float(np.float32(0.8)), 1.0, 0.6), diffuse=0.7, specular=0.2)
    if outer_ambient is not None:
        m1.ambient = outer_ambient
    b.add_sphere(R.Mat4.identity(), m1)
    m2 = R.default_material()
    if inner_ambient is not None:
        m2.ambient = inner_ambient
    b.add_sphere(R.Mat4.scale(0.5, 0.5, 0.5), m2)
    if extra:
        extra(b)
    return b.finish(0)


# ---- ray / sphere, RayTracer.c:2916-2981, 3179-3218 ----
@pytest.mark.parametrize("origin,expect", [((0, 0, -5), [4.0, 6.0]), ((0, 1, -5), [5.0, 5.0]), ((0, 2, -5), []),
                                           ((0, 0, 0), [-1.0, 1.0]), ((0, 0, 5), [-6.0, -4.0])])
def test_ray_sphere(origin, expect):
    n, ts = sphere_ts(R.Mat4.identity(), origin, (0, 0, 1))
    assert n == len(expect) and close(ts, expect)


def test_ray_scaled_and_translated_sphere():
    n, ts = sphere_ts(R.Mat4.scale(2, 2, 2), (0, 0, -5), (0, 0, 1))
    assert n == 2 and close(ts, [3.0, 7.0])
    n, _ = sphere_ts(R.Mat4.translate(5, 0, 0), (0, 0, -5), (0, 0, 1))
    assert n == 0


# ---- hit() and the list, RayTracer.c:3028-3088 ----
@pytest.mark.parametrize("ts,expect_t", [([1, 2], 1), ([-1, 1], 1), ([-2, -1], None), ([5, 7, -3, 2], 2)])
def test_hit_selection(ts, expect_t):
    arr = (C.c_double * len(ts))(*[float(t) for t in ts])
    i = O.lib().rto_hit_list(arr, len(ts))
    assert (None if i < 0 else ts[i]) == expect_t


def test_list_capacity_is_100():
    assert O.lib().rto_list_capacity(101) == 100


def test_hit_takes_first_of_equal_ts_and_accepts_zero():
    arr = (C.c_double * 3)(2.0, 2.0, 3.0)
    assert O.lib().rto_hit_list(arr, 3) == 0  # stable sort + strict '<' in hit(): earlier list entry wins
    arr = (C.c_double * 2)(-1.0, 5.0)
    assert O.lib().rto_hit_list(arr, 2) == 1


# ---- normals, RayTracer.c:3220-3303 ----
def test_sphere_normals():
    def normal(transform, p):
        out = (C.c_double * 3)()
        O.lib().rto_normal_at_sphere(O.m16(R.Mat4.inverse(transform)), O.d3(p), out)
        return list(out)

    I = R.Mat4.identity()
    assert close(normal(I, (1, 0, 0)), (1, 0, 0))
    assert close(normal(I, (0, 1, 0)), (0, 1, 0))
    r3 = math.sqrt(3) / 3
    assert close(normal(I, (r3, r3, r3)), (r3, r3, r3))
    assert close(normal(R.Mat4.translate(0, 1, 0), (0, 1.70711, -0.70711)), (0, 0.70711, -0.70711)) or np.allclose(
        normal(R.Mat4.translate(0, 1, 0), (0, 1.70711, -0.70711)), (0, 0.70711, -0.70711), atol=1e-5)
    m = R.Mat4.mul(R.Mat4.scale(1, 0.5, 1), R.Mat4.rotate_z(math.pi / 5))
    assert close(normal(m, (0, S2, -S2)), (0, 0.97014250, -0.24253563))


def test_reflect():  # RayTracer.c:3307-3326
    out = (C.c_double * 3)()
    O.lib().rto_reflect(O.d3((1, -1, 0)), O.d3((0, 1, 0)), out)
    assert close(out[:], (1, 1, 0))
    O.lib().rto_reflect(O.d3((0, -1, 0)), O.d3((S2, S2, 0)), out)
    assert close(out[:], (1, 0, 0))


# ---- lighting, RayTracer.c:3394-3476, 4256-4270 ----
@pytest.mark.parametrize("eye,light,shadow,expect", [
    ((0, 0, -1), (0, 0, -10), 0, 1.9),
    ((0, S2, -S2), (0, 0, -10), 0, 1.0),
    ((0, 0, -1), (0, 10, -10), 0, 0.73639608769926945),
    ((0, -S2, -S2), (0, 10, -10), 0, 1.6363960638574115),
    ((0, 0, -1), (0, 0, 10), 0, 0.1),
    ((0, 0, -1), (0, 0, -10), 1, 0.1),
])
def test_lighting(eye, light, shadow, expect):
    m = R.default_material()
    L = R._ffi.Light()
    L.position[:] = [float(x) for x in light]
    L.intensity[:] = [1.0, 1.0, 1.0]
    out = (C.c_double * 3)()
    O.lib().rto_lighting(C.byref(m), C.byref(L), O.d3((0, 0, 0)), O.d3(eye), O.d3((0, 0, -1)), shadow, out)
    assert close(out[:], [expect] * 3)


# ---- world, RayTracer.c:3806-4092 ----
def test_intersect_default_world():
    w = default_world()
    ts = (C.c_double * 8)()
    n = C.c_int()
    t = C.c_double()
    hit = O.lib().rto_hit_world(C.byref(w), C.byref(O.ray((0, 0, -5), (0, 0, 1))), C.byref(t), C.byref(n), ts, 8)
    assert n.value == 4 and close(ts[:4], [4, 4.5, 5.5, 6]) and hit == 0 and close(t.value, 4.0)


def test_prepare_computations():
    w = default_world()
    out = (C.c_double * 16)()
    inside = O.lib().rto_prepare(C.byref(w), C.byref(O.ray((0, 0, -5), (0, 0, 1))), 0, 4.0, out)
    assert inside == 0 and close(out[0:3], (0, 0, -1)) and close(out[3:6], (0, 0, -1)) and close(out[6:9], (0, 0, -1))
    inside = O.lib().rto_prepare(C.byref(w), C.byref(O.ray((0, 0, 0), (0, 0, 1))), 0, 1.0, out)
    assert inside == 1 and close(out[0:3], (0, 0, 1)) and close(out[3:6], (0, 0, -1)) and close(out[6:9], (0, 0, -1))


def test_hit_should_offset_the_point():  # RayTracer.c:4343-4356
    b = R.SceneBuilder()
    b.set_camera(4, 4, 1.0, (0, 0, -5), (0, 0, 0)).set_light((0, 0, -10), (1, 1, 1))
    b.add_sphere(R.Mat4.translate(0, 0, 1))
    w = b.finish(0)
    out = (C.c_double * 16)()
    O.lib().rto_prepare(C.byref(w), C.byref(O.ray((0, 0, -5), (0, 0, 1))), 0, 5.0, out)
    assert out[11] < -EPS / 2 and out[2] > out[11]


def test_shade_hit_and_color_at():
    col = (C.c_double * 3)()
    w = default_world()
    O.lib().rto_color_at(C.byref(w), C.byref(O.ray((0, 0, -5), (0, 0, 1))), 5, col)
    assert close(col[:], (0.38066119994542108, 0.47582649284140904, 0.28549590704943306))
    O.lib().rto_color_at(C.byref(w), C.byref(O.ray((0, 0, -5), (0, 1, 0))), 5, col)
    assert close(col[:], (0, 0, 0))
    w = default_world(light=(0.0, 0.25, 0.0))  # shading an intersection from the inside
    O.lib().rto_color_at(C.byref(w), C.byref(O.ray((0, 0, 0), (0, 0, 1))), 5, col)
    assert close(col[:], [0.90498445224856761] * 3)
    w = default_world(outer_ambient=1.0, inner_ambient=1.0)  # intersection behind the ray: inner colour
    O.lib().rto_color_at(C.byref(w), C.byref(O.ray((0, 0, 0.75), (0, 0, -1))), 5, col)
    assert close(col[:], (1, 1, 1))


@pytest.mark.parametrize("point,expect", [((0, 10, 0), 0), ((10, -10, 10), 1), ((-20, 20, -20), 0), ((-2, 2, -2), 0)])
def test_is_shadowed(point, expect):  # RayTracer.c:4273-4310
    w = default_world()
    assert O.lib().rto_is_shadowed(C.byref(w), O.d3(point)) == expect


def test_shade_hit_in_shadow():  # RayTracer.c:4313-4340
    b = R.SceneBuilder()
    b.set_camera(4, 4, 1.0, (0, 0, -5), (0, 0, 0)).set_light((0, 0, -10), (1, 1, 1))
    b.add_sphere(R.Mat4.identity())
    b.add_sphere(R.Mat4.translate(0, 0, 10))
    w = b.finish(0)
    col = (C.c_double * 3)()
    O.lib().rto_color_at(C.byref(w), C.byref(O.ray((0, 0, 5), (0, 0, 1))), 5, col)
    assert close(col[:], (0.1, 0.1, 0.1))


# ---- camera, RayTracer.c:4095-4236 ----
def test_camera_and_ray_for_pixel():
    b = R.SceneBuilder()
    b.set_light((0, 0, 0), (1, 1, 1))
    b.set_camera_view(200, 125, math.pi / 2, R.Mat4.identity())
    assert close(b.finish(0).camera.pixel_size, 0.01)
    b.set_camera_view(125, 200, math.pi / 2, R.Mat4.identity())
    assert close(b.finish(0).camera.pixel_size, 0.01)
    b.set_camera_view(201, 101, math.pi / 2, R.Mat4.identity())
    s = b.finish(0)
    r = O.Ray()
    O.lib().rto_ray_for_pixel(C.byref(s.camera), 100.0, 50.0, C.byref(r))
    assert close(r.origin[:3], (0, 0, 0)) and close(r.direction[:3], (0, 0, -1))
    O.lib().rto_ray_for_pixel(C.byref(s.camera), 0.0, 0.0, C.byref(r))
    assert close(r.direction[:3], (0.66518642611945078, 0.33259321305972539, -0.66851235825004807))
    b.set_camera_view(201, 101, math.pi / 2, R.Mat4.mul(R.Mat4.rotate_y(math.pi / 4), R.Mat4.translate(0, -2, 5)))
    s = b.finish(0)
    O.lib().rto_ray_for_pixel(C.byref(s.camera), 100.0, 50.0, C.byref(r))
    assert close(r.origin[:3], (0, 2, -5)) and close(r.direction[:3], (S2, 0, -S2))


def test_view_transform():
    assert close(R.Mat4.view_transform((0, 0, 0), (0, 0, -1), (0, 1, 0)), np.eye(4))
    assert close(R.Mat4.view_transform((0, 0, 0), (0, 0, 1), (0, 1, 0)), R.Mat4.scale(-1, 1, -1))
    assert close(R.Mat4.view_transform((0, 0, 8), (0, 0, 0), (0, 1, 0)), R.Mat4.translate(0, 0, -8))
    m = R.Mat4.view_transform((1, 3, 2), (4, -2, 8), (1, 1, 0))
    want = [[-0.50709, 0.50709, 0.67612, -2.36643], [0.76772, 0.60609, 0.12122, -2.82843],
            [-0.35857, 0.59761, -0.71714, 0.0], [0, 0, 0, 1]]
    assert np.allclose(m, want, atol=1e-5)


# ---- triangles, RayTracer.c:6789-6890 ----
def test_triangle():
    b = R.SceneBuilder()
    b.set_camera(4, 4, 1.0, (0, 0, -5), (0, 0, 0)).set_light((0, 0, -10), (1, 1, 1))
    b.add_triangle((0, 1, 0), (-1, 0, 0), (1, 0, 0))
    a = R.scene_arrays(b.finish(0))["tri_soa"][:, 0]
    p1, e1, e2, n = a[0:3], a[3:6], a[6:9], a[9:12]
    assert close(e1, (-1, -1, 0)) and close(e2, (1, -1, 0)) and close(n, (0, 0, -1))

    def tri_t(origin, direction):
        t = C.c_double()
        hit = O.lib().rto_intersect_triangle(O.d3(p1), O.d3(e1), O.d3(e2), C.byref(O.ray(origin, direction)), C.byref(t))
        return t.value if hit else None

    assert tri_t((0, -1, -2), (0, 1, 0)) is None   # parallel
    assert tri_t((1, 1, -2), (0, 0, 1)) is None    # beyond the p1-p3 edge
    assert tri_t((-1, 1, -2), (0, 0, 1)) is None   # beyond the p1-p2 edge
    assert tri_t((0, -1, -2), (0, 0, 1)) is None   # beyond the p2-p3 edge
    assert close(tri_t((0, 0.5, -2), (0, 0, 1)), 2.0)


def test_color_convert():  # RayTracer.c:2004-2016
    got = [O.lib().rto_color_convert(x) for x in (0.0, 0.5, 1.0, 2.0, -2.0)]
    assert got == [0, 127, 255, 255, 0]


# ---- reflection, RayTracer.c:4861-4974.  The reference's tests put a PLANE at y = -1; planes are outside
# the accelerated path, so the same surface is given as two large triangles (same point, same normal; the
# hit t agrees to rounding, far inside the 1e-6 tolerance). ----
def reflective_floor(reflective):
    def extra(b):
        m = R.default_material(reflective=reflective)
        b.add_triangle((-100, -1, -100), (-100, -1, 100), (100, -1, 100), m)
        b.add_triangle((-100, -1, -100), (100, -1, 100), (100, -1, -100), m)
    return extra


def test_reflected_colour_and_shade_hit_with_reflective_floor():
    w = default_world(extra=reflective_floor(0.5))
    col = (C.c_double * 3)()
    O.lib().rto_color_at(C.byref(w), C.byref(O.ray((0, 0, -3), (0, -S2, S2))), 5, col)
    assert close(col[:], (0.87675614729320861, 0.92433883683716145, 0.82917345774925577))
    # reflected part = shade_hit - surface; surface is what remaining = 0 leaves
    surf = (C.c_double * 3)()
    O.lib().rto_color_at(C.byref(w), C.byref(O.ray((0, 0, -3), (0, -S2, S2))), 0, surf)
    assert close(np.array(col[:]) - np.array(surf[:]), (0.19033077235655871, 0.23791346190051155, 0.14274808281260587))


def test_mutually_reflective_surfaces_terminate():
    b = R.SceneBuilder()
    b.set_camera(4, 4, 1.0, (0, 0, -5), (0, 0, 0)).set_light((0, 0, 0), (1, 1, 1))
    m = R.default_material(reflective=1.0)
    for y, flip in ((-1.0, False), (1.0, True)):
        a, c, d, e = (-50, y, -50), (-50, y, 50), (50, y, 50), (50, y, -50)
        b.add_triangle(a, c, d, m)
        b.add_triangle(a, d, e, m)
    col = (C.c_double * 3)()
    O.lib().rto_color_at(C.byref(b.finish(0)), C.byref(O.ray((0, 0, 0), (0, 1, 0))), 5, col)
    assert np.isfinite(col[:]).all()


# ---- planes, cubes, cylinders, patterns, Schlick: the reference's tests at RayTracer.c:4456-4850, 5262-5720 ----
def shape(kind, transform=None, minimum=-math.inf, maximum=math.inf, closed=False):
    sh = R._ffi.Shape()
    sh.inv[:] = list(np.asarray(R.Mat4.inverse(R.Mat4.identity() if transform is None else transform)).reshape(16))
    sh.minimum, sh.maximum, sh.closed = minimum, maximum, int(closed)
    sh.type = {"plane": R._ffi.RT_SHAPE_PLANE, "cube": R._ffi.RT_SHAPE_CUBE, "cylinder": R._ffi.RT_SHAPE_CYLINDER}[kind]
    return sh


def shape_ts(sh, origin, direction, normalise=False):
    d = np.asarray(direction, dtype=float)
    if normalise:
        d = d / np.sqrt((d * d).sum())
    t = (C.c_double * 8)()
    n = O.lib().rto_intersect_shape(C.byref(sh), C.byref(O.ray(origin, d)), t, 8)
    return n, list(t[:n])


def test_plane_intersections():  # :4483-4529
    pl = shape("plane")
    assert shape_ts(pl, (0, 10, 0), (0, 0, 1))[0] == 0
    assert shape_ts(pl, (0, 0, 0), (0, 0, 1))[0] == 0
    assert shape_ts(pl, (0, 1, 0), (0, -1, 0)) == (1, [1.0])
    assert shape_ts(pl, (0, -1, 0), (0, 1, 0)) == (1, [1.0])
    n = (C.c_double * 4)()
    for p in ((0, 0, 0), (10, 0, -10), (-5, 0, 150)):  # normal_of_plane_is_const_everywhere_test, :4456
        O.lib().rto_normal_at_shape(C.byref(pl), O.d3(p), n)
        assert close(n[:], (0, 1, 0, 0))


@pytest.mark.parametrize("origin,direction", [((5, 0.5, 0), (-1, 0, 0)), ((-5, 0.5, 0), (1, 0, 0)), ((0.5, 5, 0), (0, -1, 0)),
                                              ((0.5, -5, 0), (0, 1, 0)), ((0.5, 0, 5), (0, 0, -1)), ((0.5, 0, -5), (0, 0, 1))])
def test_ray_intersects_cube(origin, direction):  # ray_intersects_cube_test, :5367
    n, ts = shape_ts(shape("cube"), origin, direction)
    assert n == 2 and close(ts, (4, 6))


def test_ray_inside_cube_and_misses():  # ray inside the cube (:5367-5422), ray_misses_cube_test (:5424)
    n, ts = shape_ts(shape("cube"), (0, 0.5, 0), (0, 0, 1))
    assert n == 2 and close(ts, (-1, 1))
    for o, d in (((-2, 0, 0), (0.2673, 0.5345, 0.8018)), ((0, -2, 0), (0.8018, 0.2673, 0.5345)), ((0, 0, -2), (0.5345, 0.8018, 0.2673)),
                 ((2, 0, 2), (0, 0, -1)), ((0, 2, 2), (0, -1, 0)), ((2, 2, 0), (-1, 0, 0))):
        assert shape_ts(shape("cube"), o, d)[0] == 0


def test_normal_on_surface_of_cube():  # :5462: the point itself decides, unnormalised, returned as a point (w == 1)
    cu = shape("cube")
    n = (C.c_double * 4)()
    for p, want in (((1, 0.5, -0.8), (1, 0, 0)), ((-1, -0.2, 0.9), (-1, 0, 0)), ((-0.4, 1, -0.1), (0, 1, 0)),
                    ((0.3, -1, -0.7), (0, -1, 0)), ((-0.6, 0.3, 1), (0, 0, 1)), ((0.4, 0.4, -1), (0, 0, -1)),
                    ((1, 1, 1), (1, 0, 0)), ((-1, -1, -1), (-1, 0, 0))):
        O.lib().rto_normal_at_shape(C.byref(cu), O.d3(p), n)
        assert close(n[:3], want) and n[3] == 1.0


def test_cylinder_intersections():  # :5517-5720
    cyl = shape("cylinder")
    for o, d in (((1, 0, 0), (0, 1, 0)), ((0, 0, 0), (0, 1, 0)), ((0, 0, -5), (1, 1, 1))):  # ray_misses_cylinder_test
        assert shape_ts(cyl, o, d, normalise=True)[0] == 0
    assert close(shape_ts(cyl, (1, 0, -5), (0, 0, 1), True)[1], (5, 5))
    assert close(shape_ts(cyl, (0, 0, -5), (0, 0, 1), True)[1], (4, 6))
    assert close(shape_ts(cyl, (0.5, 0, -5), (0.1, 1, 1), True)[1], (6.8079819, 7.0887234))
    lim = shape("cylinder", minimum=1.0, maximum=2.0)  # intersecting_constrained_cylinder, :5621
    for o, d, want in (((0, 1.5, 0), (0.1, 1, 0), 0), ((0, 3, -5), (0, 0, 1), 0), ((0, 0, -5), (0, 0, 1), 0),
                       ((0, 2, -5), (0, 0, 1), 0), ((0, 1, -5), (0, 0, 1), 0), ((0, 1.5, -2), (0, 0, 1), 2)):
        assert shape_ts(lim, o, d)[0] == want
    capped = shape("cylinder", minimum=1.0, maximum=2.0, closed=True)  # intersecting_caps_of_closed_cylinder_test, :5671
    for o, d in (((0, 3, 0), (0, -1, 0)), ((0, 3, -2), (0, -1, 2)), ((0, 4, -2), (0, -1, 1)), ((0, 0, -2), (0, 1, 2)),
                 ((0, -1, -2), (0, 1, 1))):
        assert shape_ts(capped, o, d, normalise=True)[0] == 2
    n = (C.c_double * 4)()
    for p, want in (((1, 0, 0), (1, 0, 0)), ((0, 5, -1), (0, 0, -1)), ((0, -2, 1), (0, 0, 1)), ((-1, 1, 0), (-1, 0, 0))):
        O.lib().rto_normal_at_shape(C.byref(cyl), O.d3(p), n)  # normal_vector_on_cylinder_test, :5590
        assert close(n[:], (*want, 0))


def pattern_colour(kind, point, frm=(1, 1, 1), to=(0, 0, 0)):
    x = R.material_ext(pattern_type=kind, pattern_from=frm, pattern_to=to)
    out = (C.c_double * 3)()
    O.lib().rto_pattern_at(C.byref(x), O.d3(point), out)
    return list(out)


def test_patterns():  # :4531-4850
    P = R._ffi
    W, B = [1, 1, 1], [0, 0, 0]
    for p, want in (((0, 0, 0), W), ((0, 1, 0), W), ((0, 2, 0), W), ((0, 0, 1), W), ((0, 0, 2), W), ((0.9, 0, 0), W),
                    ((1, 0, 0), B), ((-0.1, 0, 0), B), ((-1, 0, 0), B), ((-1.1, 0, 0), W)):
        assert pattern_colour(P.RT_PATTERN_STRIPE, p) == want, p
    for p, want in (((0, 0, 0), W), ((1, 0, 0), B), ((0, 0, 1), B), ((float(np.float32(0.708)), 0, float(np.float32(0.708))), B)):
        assert pattern_colour(P.RT_PATTERN_RING, p) == want, p
    for p, want in (((0, 0, 0), W), ((0.99, 0, 0), W), ((1.01, 0, 0), B), ((0, 0.99, 0), W), ((0, 1.01, 0), B),
                    ((0, 0, 0.99), W), ((0, 0, 1.01), B)):
        assert pattern_colour(P.RT_PATTERN_CHECKER, p) == want, p
    assert close(pattern_colour(P.RT_PATTERN_GRADIENT, (0.25, 0, 0)), (0.75, 0.75, 0.75))  # gradiant_linearly_interpolates_between_colors_test, :4722
    assert close(pattern_colour(P.RT_PATTERN_GRADIENT, (0.5, 0, 0)), (0.5, 0.5, 0.5))
    assert close(pattern_colour(P.RT_PATTERN_GRADIENT, (0.75, 0, 0)), (0.25, 0.25, 0.25))
    assert pattern_colour(P.RT_PATTERN_TEST, (2, 3, 4)) == [2, 3, 4]


def test_schlick():  # :5262-5298: eyev / normalv as prepare_computations leaves them for those rays
    L = O.lib()
    assert close(L.rto_schlick(O.d3((0, -1, 0)), O.d3((0, -1, 0)), 1.5, 1.0), 0.04)  # perpendicular, from inside
    assert close(L.rto_schlick(O.d3((0, -1, 0)), O.d3((-S2, 0, -S2)), 1.5, 1.0), 1.0)  # total internal reflection
    # small angle, n2 > n1: ray (0, 0.99, -2) -> +z hits the unit glass sphere at t = 1.8589
    pz = -2 + 1.8589
    nrm = np.array([0, 0.99, pz]) / np.sqrt(0.99 ** 2 + pz ** 2)
    assert close(L.rto_schlick(O.d3((0, 0, -1)), O.d3(nrm), 1.0, 1.5), 0.48873)


def test_finding_n1_and_n2_at_various_intersections():  # finding_n1_and_n2_at_various_intersections_test, :5066: three overlapping glass spheres
    b = R.SceneBuilder()
    b.set_camera(8, 8, 1.0, (0, 0, -5), (0, 0, 0))
    b.set_light((-10, 10, -10), (1, 1, 1))
    m = R.default_material()
    b.add_sphere_ext(R.Mat4.scale(2, 2, 2), m, R.material_ext(transparency=1.0, refractive_index=1.5))
    b.add_sphere_ext(R.Mat4.translate(0, 0, -0.25), m, R.material_ext(transparency=1.0, refractive_index=2.0))
    b.add_sphere_ext(R.Mat4.translate(0, 0, 0.25), m, R.material_ext(transparency=1.0, refractive_index=2.5))
    s = b.finish(0)
    r = O.ray((0, 0, -4), (0, 0, 1))
    want = [(1.0, 1.5), (1.5, 2.0), (2.0, 2.5), (2.5, 2.5), (2.5, 1.5), (1.5, 1.0)]
    out = (C.c_double * 2)()
    for i, w in enumerate(want):
        assert O.lib().rto_refractive_indices(C.byref(s), C.byref(r), i, out) == 6
        assert tuple(out) == w, i
