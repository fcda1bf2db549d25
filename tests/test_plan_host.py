"""The planner's conservativeness, without a GPU.

Before a launch the library decides on the host, per 16 x 16 tile, which triangles and spheres a camera ray of the
tile can hit and which can stand between such a hit and the light (raytracer_amd/csrc/rt_plan.hip.inc: a conservative
rasterisation, clipped pyramids of rays, separating planes), and the kernels test nothing else.  The reference tests
every object for every ray (intersect_world, RayTracer/RayTracer.c:1043-1055), so a set that misses a primitive the
reference's loop would have found changes the image.

Here the plan is made by the development build of the library in its host-only mode (rt_hip_debug_plan_host: no
device is touched) and decoded the way the kernels decode it; the CPU oracle then traces every pixel the reference's
way, and every primitive it hits, and every primitive that blocks the way from that hit to the light
(is_shadowed, RayTracer.c:1410-1422: 0 <= t < distance), has to be in the tile's sets.
"""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O
import raytracer_amd as R
from raytracer_amd import _ffi
from raytracer_amd.scene import Mat4, SceneBuilder, default_material

TILE = 16
NO_LIST = 0xFFFFFFFF


def soup(rng, n):
    """n triangles of mixed sizes around the origin, corners on a 1/64 grid (shared planes and edges occur)."""
    c = rng.uniform(-2.0, 2.0, size=(n, 1, 3))
    size = rng.choice([0.3, 0.8, 2.5], size=(n, 1, 1))
    v = c + rng.uniform(-1.0, 1.0, size=(n, 3, 3)) * size
    return (np.round(v * 64.0) / 64.0).reshape(n, 9).astype(np.float32)


def dev_lib():
    path = os.path.join(_ffi.LIB_DIR, "librt_hip_dev.so")
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run `make -C raytracer_amd`")
    try:
        import torch  # noqa: F401  (one HIP runtime per process: see _ffi.hip_lib)
    except ImportError:
        pass
    L = C.CDLL(path)
    L.rt_hip_debug_plan_host.argtypes = [C.POINTER(_ffi.Scene), C.POINTER(_ffi.RenderParams), C.c_void_p, C.c_size_t,
                                         C.POINTER(C.c_int64)]
    return L


class Plan:
    """The words the library would copy to the device for a launch, read as the kernels read them."""

    def __init__(self, scene, **kw):
        L = dev_lib()
        p = R.make_params(**kw)
        info = (C.c_int64 * 10)()
        words = np.zeros(1 << 22, dtype=np.int32)
        rc = L.rt_hip_debug_plan_host(C.byref(scene), C.byref(p), words.ctypes.data, words.size, info)
        assert rc == 0, rc
        (self.n_words, self.entries, self.heavy, self.mask_offset, self.list_forms, self.list_total, self.tiles_x,
         self.light_background, self.bvh, self.leaf_order) = [int(v) for v in info]
        self.words = words[: self.n_words]
        # the masks count spheres in the device's order: device sphere k is the scene's sphere sphere_of_device[k]
        self.sphere_of_device = [int(v) for v in words[self.n_words: self.n_words + scene.n_spheres]]
        self.entry_of_tile = {}
        for k in range(self.entries):
            self.entry_of_tile.setdefault(int(self.words[k]) & 0x3FFFFF, []).append(k)

    def masks(self, k):
        """the four 64-bit words behind entry k"""
        assert self.mask_offset > 0
        w = self.words[self.mask_offset + 8 * k: self.mask_offset + 8 * k + 8].view(np.uint64)
        return [int(x) for x in w]

    def listed(self, where):
        """triangle numbers of a list word (offset | count << 32), or None where the entry has no list"""
        off, count = where & 0xFFFFFFFF, where >> 32
        if count == NO_LIST:
            return None
        base = self.list_forms + 16 * self.list_total + off
        return set(int(t) for t in self.words[base: base + count])


def part_rect(lg, part):
    """The pixels of part `part` of 2^lg of a tile (rt_kernels.hip.inc: rt_part_rect): x0, y0, w, h inside the tile."""
    bits = [(part >> (lg - 1 - k)) & 1 for k in range(lg)] + [0] * (6 - lg)
    x0 = 8 * bits[0] + 4 * bits[2] + 2 * bits[4] + bits[5]
    w = 1 if lg >= 6 else (2 if lg >= 5 else (4 if lg >= 3 else (8 if lg >= 1 else 16)))
    y0 = 8 * bits[1] + 4 * bits[3]
    h = 4 if lg >= 4 else (8 if lg >= 2 else 16)
    return x0, y0, w, h


def device_numbers(scene, leaf_order, sphere_of_device):
    """world-list position -> ('tri' | 'sph', number the device knows the primitive by)"""
    out = {}
    for k, i in enumerate(sphere_of_device):
        out[int(scene.sph_order[i])] = ("sph", k)
    device_of = {int(scene.bvh_tri[k]): k for k in range(scene.n_tris)} if leaf_order else None
    for i in range(scene.n_tris):
        out[int(scene.tri_order[i])] = ("tri", device_of[i] if device_of else i)
    return out


def check_scene(scene, accel=_ffi.RT_ACCEL_BRUTE):
    plan = Plan(scene, accel=accel)
    names = device_numbers(scene, plan.leaf_order, plan.sphere_of_device)
    sph_device_of = {i: k for k, i in enumerate(plan.sphere_of_device)}
    L = O.lib()
    nt, ns = scene.n_tris, scene.n_spheres
    soa = np.ctypeslib.as_array(scene.tri_soa, shape=(12, nt)).copy() if nt else np.zeros((12, 0))
    inv = np.ctypeslib.as_array(scene.sph_inv, shape=(ns, 16)).copy() if ns else np.zeros((0, 16))
    device_of = {int(scene.bvh_tri[k]): k for k in range(nt)} if plan.leaf_order else {i: i for i in range(nt)}
    light = np.array(list(scene.light.position))
    W, H = scene.camera.hsize, scene.camera.vsize
    ray, sray = O.Ray(), O.Ray()
    t = C.c_double()
    n_int = C.c_int()
    prep = (C.c_double * 16)()
    tt = C.c_double()
    roots = (C.c_double * 2)()
    hits = shadow_checks = 0
    for y in range(H):
        for x in range(W):
            L.rto_ray_for_pixel(C.byref(scene.camera), C.c_double(x), C.c_double(y), C.byref(ray))
            order = L.rto_hit_world(C.byref(scene), C.byref(ray), C.byref(t), C.byref(n_int), None, 0)
            tile = (y // TILE) * plan.tiles_x + x // TILE
            ks = plan.entry_of_tile.get(tile) if plan.entries else None
            if order < 0:
                continue
            hits += 1
            kind, number = names[order]
            if plan.entries == 0:
                continue  # no plan: every tile traces everything
            assert ks, f"pixel ({x}, {y}) sees something but its tile is not in the plan"
            # a tile split over several workgroups has an entry per part, each with the sets of its own pixels
            k = None
            for cand in ks:
                e = int(plan.words[cand])
                px0, py0, pw, ph = part_rect((e >> 28) & 7, (e >> 22) & 63)
                if px0 <= x % TILE < px0 + pw and py0 <= y % TILE < py0 + ph:
                    assert k is None, f"pixel ({x}, {y}) belongs to two parts of its tile"
                    k = cand
            assert k is not None, f"pixel ({x}, {y}) belongs to no part of its tile"
            assert not (k >= plan.heavy and plan.light_background), f"pixel ({x}, {y}) sees something in a tile planned as background"
            if plan.mask_offset == 0:
                continue
            m = plan.masks(k)
            cam_list = shadow_list = None
            if plan.list_forms:
                cam_list, shadow_list = plan.listed(m[0]), plan.listed(m[1])
                if cam_list is not None:
                    assert kind == "tri" and number in cam_list, f"pixel ({x}, {y}): triangle {number} is not in its tile's list"
            elif kind == "tri":
                assert (m[0] >> number) & 1, f"pixel ({x}, {y}): triangle {number} is not in its tile's mask"
            else:
                assert (m[2] >> number) & 1, f"pixel ({x}, {y}): sphere {number} is not in its tile's mask"
            # the way to the light (is_shadowed): from over_point
            L.rto_prepare(C.byref(scene), C.byref(ray), order, t, prep)
            over = np.array(prep[9:12])
            v = light - over
            dist = float(np.sqrt(v @ v))
            d = v / dist
            for c in range(3):
                sray.origin[c], sray.direction[c] = over[c], d[c]
            sray.origin[3], sray.direction[3] = 1.0, 0.0
            for i in range(nt):
                if kind == "tri" and device_of[i] == number:
                    continue  # the surface the ray starts on: left out or not, lighting() returns the same (see below)
                p1, e1, e2 = (C.c_double * 3)(*soa[0:3, i]), (C.c_double * 3)(*soa[3:6, i]), (C.c_double * 3)(*soa[6:9, i])
                if L.rto_intersect_triangle(p1, e1, e2, C.byref(sray), C.byref(tt)) and 0.0 <= tt.value < dist:
                    shadow_checks += 1
                    j = device_of[i]
                    if plan.list_forms:
                        assert shadow_list is None or j in shadow_list, f"pixel ({x}, {y}): blocker {j} is not in the shadow list"
                    else:
                        assert (m[1] >> j) & 1, f"pixel ({x}, {y}): blocker {j} is not in the shadow mask"
            for i in range(ns):
                j = sph_device_of[i]
                if kind == "sph" and j == number:
                    continue
                if L.rto_intersect_sphere((C.c_double * 16)(*inv[i]), C.byref(sray), roots) and any(0.0 <= r_ < dist for r_ in roots):
                    shadow_checks += 1
                    assert (m[3] >> j) & 1, f"pixel ({x}, {y}): sphere {j} is not in the shadow mask"
    return plan, hits, shadow_checks


def build(tris, w, h, eye, light):
    b = SceneBuilder()
    b.set_camera(w, h, 0.7, eye, (0.0, 0.0, 0.0))
    b.set_light(light, (1.0, 1.0, 1.0))
    b.add_triangles_f32(tris)
    return b.finish(4)


def _lights(rng, eye, kind):
    return [tuple(np.array(eye) + rng.uniform(-2.0, 2.0, size=3)), tuple(-np.array(eye) * rng.uniform(0.5, 1.5)),
            tuple(rng.uniform(-1.0, 1.0, size=3)), tuple(np.round(rng.uniform(-3.0, 3.0, size=3) * 4.0) / 4.0)][kind]


# the surface a shadow ray starts on is skipped above: whether the planner keeps it or not cannot change a pixel.
# Its own t is negative when eye and light are on the same side (no entry for hit(), RayTracer.c:150), and positive
# otherwise - but then light . normal < 0 and lighting() returns the ambient term with or without the shadow
# (RayTracer.c:1296-1299, 1314-1318).


@pytest.mark.parametrize("seed", range(6))
def test_masks_of_small_triangle_worlds_hold_every_hit_and_every_blocker(seed):
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.integers(3, 65))
    direction = rng.normal(size=3)
    eye = tuple(direction / np.linalg.norm(direction) * rng.uniform(6.0, 9.0))
    scene = build(soup(rng, n), [80, 97, 64][seed % 3], [64, 45, 83][seed % 3], eye, _lights(rng, eye, seed % 4))
    plan, hits, shadow_checks = check_scene(scene)
    assert plan.mask_offset > 0 and plan.list_forms == 0, "a world of up to 64 triangles is planned with masks"
    assert hits > 10


@pytest.mark.parametrize("seed", range(4))
def test_masks_of_worlds_with_spheres_hold_every_hit_and_every_blocker(seed):
    rng = np.random.default_rng(7300 + seed)
    b = SceneBuilder()
    eye = [(0.0, 1.5, -7.0), (4.0, 3.0, -6.0), (0.0, 0.2, -1.2), (-5.0, 2.0, 3.0)][seed % 4]
    b.set_camera(80, 61, 0.9, eye, (0.0, 0.3, 0.0))
    b.set_light([(-6.0, 9.0, -7.0), (0.0, 0.6, 0.0), (3.0, 0.4, -1.0), (0.0, 12.0, 0.0)][seed % 4], (1.0, 1.0, 1.0))
    b.add_sphere(Mat4.mul(Mat4.translate(0.0, -1.0, 0.0), Mat4.scale(12.0, 0.02, 12.0)), default_material())
    for k in range(int(rng.integers(2, 7))):
        t = Mat4.mul(Mat4.translate(*rng.uniform(-2.5, 2.5, 3)), Mat4.mul(Mat4.rotate_z(rng.uniform(0, 3)), Mat4.scale(*rng.uniform(0.2, 1.4, 3))))
        b.add_sphere(t, default_material())
    b.add_triangles_f32(soup(rng, int(rng.integers(1, 20))))
    scene = b.finish(4)
    plan, hits, shadow_checks = check_scene(scene)
    # seed 2 puts the eye inside a sphere's box: such a world may go unplanned (then there is nothing to miss)
    assert plan.list_forms == 0 and (plan.mask_offset > 0 or seed == 2)
    assert hits > 500 and (shadow_checks > 0 or plan.mask_offset == 0)


@pytest.mark.parametrize("seed,accel", [(0, _ffi.RT_ACCEL_BRUTE), (1, _ffi.RT_ACCEL_BRUTE), (2, _ffi.RT_ACCEL_BVH), (3, _ffi.RT_ACCEL_BRUTE)])
def test_lists_of_larger_meshes_hold_every_hit_and_every_blocker(seed, accel):
    rng = np.random.default_rng(7600 + seed)
    n = [65, 130, 200, 97][seed]
    direction = rng.normal(size=3)
    eye = tuple(direction / np.linalg.norm(direction) * rng.uniform(6.0, 9.0))
    scene = build(soup(rng, n), [80, 64, 97, 70][seed], [48, 64, 45, 50][seed], eye, _lights(rng, eye, seed % 4))
    plan, hits, shadow_checks = check_scene(scene, accel)
    assert plan.list_forms > 0, "a mesh of more than 64 triangles is planned with lists"
    assert hits > 10
