"""The sphere loops decide in single precision which spheres need the reference's double-precision roots at all: not
one whose discriminant is negative, not one that lies behind the ray (origin outside, direction away from the
centre), not the sphere a shadow ray leaves on its lit side.  A wrong decision changes pixels, so: worlds built to
sit on the edges of those decisions (eye and light inside spheres and on their surfaces, concentric and touching
spheres, flattened and sheared ones, tiny ones far away, dozens of spheres, mirrors), each against the CPU oracle
(the colours before quantisation, to 1e-12), and at 1080p against the kernel that keeps the reference's list (no
single-precision stage)."""
import numpy as np
import pytest

import oracle_lib as O
from conftest import quantise
import raytracer_amd as R
from raytracer_amd.scene import Mat4, SceneBuilder, default_material

pytestmark = pytest.mark.gpu

REFERENCE = R._ffi.RT_ACCEL_REFERENCE


def world(seed, w, h, n, eye, to, light, scale_lo=0.2, scale_hi=2.0, spread=3.0, mirror=0.0, shear=False):
    rng = np.random.default_rng(seed)
    b = SceneBuilder()
    b.set_camera(w, h, 0.9, eye, to)
    b.set_light(light, (1.0, 0.95, 0.9))
    for k in range(n):
        m = default_material(color=rng.uniform(0.1, 1.0, 3), diffuse=rng.uniform(0.4, 0.9), specular=rng.uniform(0.0, 0.9),
                             shininess=rng.uniform(5, 300), reflective=mirror if k % 2 else 0.0)
        t = Mat4.mul(Mat4.translate(*rng.uniform(-spread, spread, 3)),
                     Mat4.mul(Mat4.rotate_y(rng.uniform(0, 3)), Mat4.scale(*np.exp(rng.uniform(np.log(scale_lo), np.log(scale_hi), 3)))))
        if shear and k % 3 == 0:
            xy, xz, yx, yz, zx, zy = rng.uniform(-0.7, 0.7, 6)
            t = Mat4.mul(t, np.array([[1, xy, xz, 0], [yx, 1, yz, 0], [zx, zy, 1, 0], [0, 0, 0, 1]], dtype=np.float64))
        b.add_sphere(t, m)
    return b.finish(0)


def check(hip, s, max_depth=5):
    hip.upload(s)
    got, st = hip.render_f64(max_depth=max_depth, count_rays=True)
    want, ost = O.render_f64(s, max_depth=max_depth)
    assert st.rays == ost.rays
    # a wrong hit moves a colour by percents; the last bits of pow() (device library against glibc) by parts in 1e16
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-15)
    assert np.array_equal(quantise(got), quantise(want))
    return st


CASES = [
    # seed, n, eye, to, light, kwargs
    (1, 6, (0.0, 1.0, -8.0), (0.0, 0.5, 0.0), (-6.0, 9.0, -7.0), {}),
    (2, 12, (0.0, 0.0, 0.0), (0.0, 0.0, 1.0), (0.5, 0.5, 0.5), dict(scale_lo=1.0, scale_hi=6.0, spread=1.0)),  # eye and light inside
    (3, 40, (5.0, 3.0, -9.0), (0.0, 0.0, 0.0), (-6.0, 9.0, -7.0), dict(scale_lo=0.1, scale_hi=1.0, mirror=0.5)),
    (4, 17, (0.0, 2.0, -12.0), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), dict(scale_lo=0.3, scale_hi=3.0)),  # light in the middle
    (5, 9, (0.0, 1.0, -30.0), (0.0, 0.0, 0.0), (-20.0, 30.0, -25.0), dict(scale_lo=0.01, scale_hi=10.0, spread=5.0)),  # flattened
    (6, 10, (3.0, 1.0, -7.0), (0.0, 0.0, 0.0), (8.0, 2.0, -3.0), dict(shear=True, mirror=0.8)),
    (7, 8, (0.0, 0.5, -400.0), (0.0, 0.0, 0.0), (-300.0, 500.0, -400.0), dict(scale_lo=0.5, scale_hi=3.0, spread=6.0)),  # far away
]


@pytest.mark.parametrize("seed,n,eye,to,light,kw", CASES)
def test_random_sphere_worlds_against_the_oracle(hip, seed, n, eye, to, light, kw):
    check(hip, world(seed, 150, 97, n, eye, to, light, **kw))


def test_concentric_touching_and_coincident_spheres(hip):
    """Equal roots on different spheres (ties go to the earlier one), spheres that touch in a point on the axis of view,
    an eye a hair outside a surface and a light exactly on one."""
    b = SceneBuilder()
    b.set_camera(161, 99, 0.8, (0.0, 0.0, -5.0), (0.0, 0.0, 0.0))
    b.set_light((0.0, 3.0, 0.0), (1.0, 1.0, 1.0))  # on the surface of the radius-3 sphere
    for k, (tr, sc) in enumerate([((0, 0, 0), 1.0), ((0, 0, 0), 1.0), ((0, 0, 0), 3.0), ((2, 0, 0), 1.0), ((-2, 0, 0), 1.0),
                                  ((0, 0, 0), 0.5), ((0, 0, -3.9999), 1.0)]):  # the last one has the eye 1e-4 off its surface
        b.add_sphere(Mat4.mul(Mat4.translate(*tr), Mat4.scale(sc, sc, sc)),
                     default_material(color=(0.2 + 0.1 * k, 0.9 - 0.1 * k, 0.5), reflective=0.3 if k == 2 else 0.0))
    check(hip, b.finish(0))


def test_the_reference_worlds_and_their_shadows(hip):
    for name in ("spheres", "cube_spheres"):
        for tr, sr in ((0.0, 0.0), (0.3, 0.3)):
            check(hip, R.stock_scene(name, 160, 120, tr, sr))


def test_single_precision_stage_spares_most_double_precision_roots(hip):
    """The reference's sphere world at 1080p: same bytes as the list kernel, and fewer than two double-precision
    solves per ray (six spheres; solving every sphere whose line the ray's line meets takes 2.5 per ray)."""
    hip.upload(R.stock_scene("spheres", 1920, 1080))
    ref, rst = hip.render(accel=REFERENCE, count_rays=True)
    got, st = hip.render(count_rays=True)
    assert np.array_equal(got, ref) and st.rays == rst.rays
    assert st.sphere_exact < 1.8 * st.rays
    assert st.sphere_tests < 4.5 * st.rays  # per-tile masks: not every ray is tested against all six


@pytest.mark.parametrize("seed", range(3))
def test_many_spheres_against_the_list_kernel_at_1080p(hip, seed):
    s = world(20 + seed, 1920, 1080, [6, 24, 50][seed], (2.0, 2.0, -10.0), (0.0, 0.0, 0.0), [(-6.0, 9.0, -7.0), (0.0, 0.0, 0.0), (5.0, -4.0, 6.0)][seed],
              scale_lo=0.2, scale_hi=1.5, spread=3.5, mirror=0.4 if seed == 1 else 0.0)
    hip.upload(s)
    ref, rst = hip.render(accel=REFERENCE, count_rays=True)
    got, st = hip.render(count_rays=True)
    assert st.rays == rst.rays
    assert np.array_equal(got, ref), f"{int((got != ref).any(axis=-1).sum())} pixels differ"
