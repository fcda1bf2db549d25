"""The BVH built on the GPU (rt_hip_build_bvh_device): whatever tree the walk is given it only prunes, so
the image must equal the one from the host-built tree, the exhaustive loop and the oracle.

The reference has no acceleration structure (intersect_world tests every object, RayTracer/RayTracer.c:1043-1055);
the oracle and the exhaustive kernel are the truth here, the trees are candidates.
"""
import numpy as np
import pytest

import oracle_lib as O
import raytracer_amd as R
from test_hip_parity import icosphere, mesh_scene, random_scene

pytestmark = pytest.mark.gpu

BVH, BRUTE = R._ffi.RT_ACCEL_BVH, R._ffi.RT_ACCEL_BRUTE


@pytest.mark.parametrize("leaf_size", [1, 2, 4, 7])
def test_lighthouse_device_tree_equals_host_tree_and_exhaustive(hip, leaf_size):
    s = R.stock_scene("lighthouse", 320, 200, tri_reflective=0.3)
    hip.upload(s)
    host_tree, st_h = hip.render(accel=BVH, count_rays=True)
    brute, st_b = hip.render(accel=BRUTE, count_rays=True)
    nodes, depth, ms = hip.build_bvh_device(leaf_size)
    assert 1 <= nodes <= s.n_tris - 1 and 1 <= depth <= 48 and ms > 0
    dev_tree, st_d = hip.render(accel=BVH, count_rays=True)
    assert st_d.accel_used == BVH
    assert np.array_equal(dev_tree, host_tree) and np.array_equal(dev_tree, brute)
    assert st_d.rays == st_h.rays == st_b.rays
    assert st_d.prim_tests < st_b.prim_tests / 10
    # the exhaustive loop still works on the re-ordered triangles, and a second build on top of the first
    again, _ = hip.render(accel=BRUTE)
    assert np.array_equal(again, brute)
    hip.build_bvh_device(leaf_size)
    twice, _ = hip.render(accel=BVH)
    assert np.array_equal(twice, brute)


def test_scene_uploaded_without_a_host_tree(hip):
    """bvh_leaf_size = 0 skips the host builder: RT_ACCEL_BVH is refused until the device build has run."""
    s = R.stock_scene("lighthouse", 200, 160, bvh_leaf_size=0)
    assert s.n_bvh_nodes == 0
    hip.upload(s)
    brute, sb = hip.render(accel=BRUTE, count_rays=True)
    with pytest.raises(R.RtError):
        hip.render(accel=BVH)
    hip.build_bvh_device(4)
    tree, st = hip.render(accel=R._ffi.RT_ACCEL_AUTO, count_rays=True)
    assert st.accel_used == BVH and np.array_equal(tree, brute) and st.rays == sb.rays
    hip.upload(s)  # a new upload drops the device tree
    with pytest.raises(R.RtError):
        hip.render(accel=BVH)


@pytest.mark.parametrize("seed,ns,nt,refl", [(2, 0, 40, 0.0), (3, 4, 60, 0.4), (4, 7, 200, 0.8), (7, 3, 2, 0.5)])
def test_seeded_worlds_device_tree_matches_oracle(hip, seed, ns, nt, refl):
    """Spheres stay outside the tree; canvases agree with the oracle to the last bits."""
    s = random_scene(seed, 120, 80, ns, nt, refl)
    want, ost = O.render_f64(s)
    hip.upload(s)
    hip.build_bvh_device(2)
    got, st = hip.render_f64(accel=BVH, count_rays=True)
    assert (st.rays_primary, st.rays_shadow, st.rays_secondary) == (ost.rays_primary, ost.rays_shadow, ost.rays_secondary)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-15)


def test_duplicate_and_degenerate_triangles(hip):
    """Equal Morton codes (the same triangle many times, zero-area triangles at one point) are split by position.
    92 triangles in all: a ray cannot fill the reference's 100-entry intersection list (RayTracer.c:108)."""
    rng = np.random.default_rng(5)
    base = rng.uniform(-1, 1, (6, 9)).astype(np.float32)
    point = np.tile(np.float32([0.2, 0.1, -0.3]), 3)
    tris = np.concatenate([np.repeat(base, 12, axis=0), np.tile(point, (20, 1))]).astype(np.float32)
    s = mesh_scene(tris, 96, 64, reflective=0.2)
    hip.upload(s)
    brute, sb = hip.render(accel=BRUTE, count_rays=True)
    nodes, depth, _ = hip.build_bvh_device(4)
    assert 1 <= nodes <= len(tris) - 1 and depth <= 48
    tree, st = hip.render(accel=BVH, count_rays=True)
    assert np.array_equal(tree, brute) and st.rays == sb.rays
    want, _ = O.render_rgb(s)
    assert O.probe().list_overflow == 0
    assert np.array_equal(tree, want)


def test_fewer_triangles_than_a_leaf_and_argument_errors(hip):
    s = mesh_scene(icosphere(0)[:3], 48, 32)
    hip.upload(s)
    brute, _ = hip.render(accel=BRUTE)
    assert hip.build_bvh_device(4)[0] == 0
    one_leaf, st = hip.render(accel=BVH)
    assert np.array_equal(one_leaf, brute)
    for bad in (0, 8, -1):
        with pytest.raises(R.RtError):
            hip.build_bvh_device(bad)
    hip.upload(R.stock_scene("refraction", 48, 32))  # planes and glass: the reference's list is walked as it is
    with pytest.raises(R.RtError):
        hip.build_bvh_device(4)


def test_large_mesh_without_a_tree_gets_one_at_upload(hip):
    """From 4096 triangles up an upload without a host tree builds the device tree itself."""
    s = mesh_scene(icosphere(5), 96, 64, reflective=0.25, bvh_leaf_size=0)
    assert s.n_tris == 20480 and s.n_bvh_nodes == 0
    hip.upload(s)
    got, st = hip.render(count_rays=True)
    assert st.accel_used == BVH and st.prim_tests < st.rays * 100
    brute, sb = hip.render(accel=BRUTE, count_rays=True)
    assert sb.accel_used == BRUTE and np.array_equal(got, brute) and st.rays == sb.rays
    want, _ = O.render_rgb(s)
    assert np.array_equal(got, want)


def test_camera_inside_a_large_mesh(hip):
    """No pixel rectangle bounds the mesh when the camera sits inside it: every tile is a mesh tile, ordered in
    per-XCD runs because the tree is read from memory."""
    from raytracer_amd.scene import SceneBuilder

    b = SceneBuilder()
    b.set_camera(96, 64, 1.2, (0.1, 0.05, -0.2), (0.3, 0.2, 1.0))
    b.set_light((0.2, 0.3, 0.1), (1, 1, 1))
    b.add_triangles_f32(icosphere(5))
    b.set_reflective(0.2, 0.0)
    s = b.finish(4)
    assert s.n_bvh_nodes > 2 * 768
    hip.upload(s)
    got, st = hip.render(accel=BVH, count_rays=True)
    brute, sb = hip.render(accel=BRUTE, count_rays=True)
    assert np.array_equal(got, brute) and st.rays == sb.rays and st.rays_shadow == 96 * 64 + st.rays_secondary
    want, ost = O.render_rgb(s)
    assert np.array_equal(got, want) and st.rays == ost.rays
    hip.build_bvh_device(4)
    dev, sd = hip.render(accel=BVH, count_rays=True)
    assert np.array_equal(dev, want) and sd.rays == ost.rays


def test_spheres_beside_a_large_mesh(hip):
    """Spheres stay outside the tree; with a tree that is read from memory the tiles are still ordered per XCD."""
    from raytracer_amd.scene import SceneBuilder

    b = SceneBuilder()
    b.set_camera(64, 40, 0.8, (0.3, 1.2, -4.5), (0, 0, 0))
    b.set_light((-4.0, 6.0, -5.0), (1, 1, 1))
    b.add_sphere(R.Mat4.mul(R.Mat4.translate(1.6, 0.2, -0.5), R.Mat4.scale(0.5, 0.5, 0.5)), R.default_material(reflective=0.4))
    b.add_triangles_f32(icosphere(5))
    b.add_sphere(R.Mat4.mul(R.Mat4.translate(-1.5, -0.3, -0.8), R.Mat4.scale(0.4, 0.7, 0.4)), R.default_material())
    b.set_reflective(0.2, 0.4)
    for leaf in (4, 0):  # host tree, then the tree built at upload
        s = b.finish(leaf)
        hip.upload(s)
        got, st = hip.render(count_rays=True)
        assert st.accel_used == BVH
        want, ost = O.render_rgb(s)
        assert np.array_equal(got, want) and st.rays == ost.rays


def test_large_meshes_device_tree(hip):
    """81,920 triangles against the oracle, 1.3 million against the host tree; build times are printed."""
    s = mesh_scene(icosphere(6), 64, 48, reflective=0.25)
    hip.upload(s)
    nodes, depth, ms = hip.build_bvh_device(4)
    got, st = hip.render(accel=BVH, count_rays=True)
    want, ost = O.render_rgb(s)
    assert (st.rays_primary, st.rays_shadow, st.rays_secondary) == (ost.rays_primary, ost.rays_shadow, ost.rays_secondary)
    assert np.array_equal(got, want)
    print(f"81920 triangles: device tree {nodes} nodes, depth {depth}, built in {ms:.3f} ms")

    s = mesh_scene(icosphere(8), 640, 400)
    hip.upload(s)
    host, sh = hip.render(accel=BVH, count_rays=True)
    nodes, depth, ms = hip.build_bvh_device(4)
    _, _, ms2 = hip.build_bvh_device(4)
    dev, sd = hip.render(accel=BVH, count_rays=True)
    assert np.array_equal(dev, host) and sd.rays == sh.rays and dev.max() > 0
    print(f"1310720 triangles: device tree depth {depth}, built in {ms:.3f} ms (again: {ms2:.3f} ms); "
          f"kernel {sd.kernel_ms:.3f} ms vs {sh.kernel_ms:.3f} ms with the host SAH tree; "
          f"triangle tests {sd.prim_tests} vs {sh.prim_tests}, box tests {sd.node_tests} vs {sh.node_tests}")
