"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports what include/rt_hip.h
declares, struct layouts agree with the header, and the host helpers (BVH, OBJ reader, PPM writer, band
arithmetic) behave.  No GPU compute is called here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import raytracer_amd as R
from raytracer_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_hip_library_exports_every_declared_symbol():
    names = declared_functions("rt_hip.h")
    assert set(names) == set(_ffi.HIP_EXPORTS)
    L = _ffi.hip_lib()
    for n in names:
        assert hasattr(L, n), n
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(_ffi.LIB_DIR, "librt_hip.so")],
                         capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (rt_\w+)", out))
    assert set(names) <= exported


def test_sort_library_is_separate_and_the_main_library_small():
    """The sort and scan of the device BVH build (hundreds of library kernels) live in librt_hip_sort.so, which
    librt_hip.so loads on demand; what every process loads at start-up stays under 2 MiB."""
    sort = os.path.join(_ffi.LIB_DIR, "librt_hip_sort.so")
    out = subprocess.run(["nm", "-D", "--defined-only", sort], capture_output=True, text=True, check=True).stdout
    assert {"rt_sort_pairs_u64_u32", "rt_exclusive_sum_u32"} <= set(re.findall(r" T (rt_\w+)", out))
    assert os.path.getsize(os.path.join(_ffi.LIB_DIR, "librt_hip.so")) < 2 * 1024 * 1024


def test_host_library_exports_every_declared_symbol():
    L = _ffi.host_lib()
    for n in declared_functions("rt_scene.h"):
        assert hasattr(L, n), n


def test_struct_layouts_match_the_header():
    """sizeof/offsetof as a C compiler sees include/rt_hip.h vs the ctypes mirror."""
    src = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "rt_hip.h"
    int main(void) {
      printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(rt_material), sizeof(rt_camera), sizeof(rt_light), sizeof(rt_bvh_node),
             sizeof(rt_scene), sizeof(rt_render_params), sizeof(rt_stats));
      printf("%zu %zu %zu %zu %zu\n", offsetof(rt_scene, materials), offsetof(rt_scene, sph_inv), offsetof(rt_scene, tri_soa),
             offsetof(rt_scene, bvh_nodes), offsetof(rt_stats, kernel_ms));
      return 0;
    }'''
    exe = "/tmp/rt_layout_probe"
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe], input=src, text=True, check=True)
    lines = subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split("\n")
    sizes = [int(x) for x in lines[0].split()]
    offs = [int(x) for x in lines[1].split()]
    assert sizes == [C.sizeof(t) for t in (_ffi.Material, _ffi.Camera, _ffi.Light, _ffi.BvhNode, _ffi.Scene,
                                          _ffi.RenderParams, _ffi.Stats)]
    assert offs == [_ffi.Scene.materials.offset, _ffi.Scene.sph_inv.offset, _ffi.Scene.tri_soa.offset,
                    _ffi.Scene.bvh_nodes.offset, _ffi.Stats.kernel_ms.offset]


def test_calls_before_init_return_status_codes():
    """No GPU here (or not initialised): entry points answer with a status, they never abort."""
    L = _ffi.hip_lib()
    p = R.make_params()
    buf = (C.c_uint8 * 16)()
    assert L.rt_hip_render(C.byref(p), buf, None) in (_ffi.RT_ERR_NOT_READY,)
    assert L.rt_hip_render(None, buf, None) == _ffi.RT_ERR_INVALID
    assert L.rt_hip_upload_scene(None) == _ffi.RT_ERR_INVALID
    assert L.rt_hip_strerror(_ffi.RT_ERR_UNSUPPORTED).decode().startswith("scene uses")


@pytest.mark.parametrize("h,band,n", [(1080, 16, 8), (1080, 16, 3), (17, 4, 2), (5, 8, 4), (64, 1, 7), (0, 4, 2)])
def test_band_arithmetic(h, band, n):
    L = _ffi.hip_lib()
    seen = []
    for r in range(n):
        rows = L.rt_hip_rank_rows(h, band, n, r)
        ys = [L.rt_hip_rank_row_to_y(i, band, n, r) for i in range(rows)]
        assert ys == [y for y in range(h) if (y // band) % n == r]
        seen += ys
    assert sorted(seen) == list(range(h))
    assert L.rt_hip_rank_rows(h, band, n, n) < 0 and L.rt_hip_rank_rows(h, 0, 2, 0) < 0


def test_bvh_is_a_valid_conservative_hierarchy():
    s = R.stock_scene("lighthouse", 64, 64, bvh_leaf_size=4)
    n = s.n_bvh_nodes
    nodes = np.ctypeslib.as_array(C.cast(s.bvh_nodes, C.POINTER(C.c_float)), shape=(n, 8)).copy()
    lo, hi = nodes[:, 0:3], nodes[:, 4:7]
    left = nodes[:, 3].view(np.int32)
    right = nodes[:, 7].view(np.int32)
    tri_idx = np.ctypeslib.as_array(s.bvh_tri, shape=(s.n_tris,)).copy()
    assert sorted(tri_idx.tolist()) == list(range(s.n_tris))
    a = R.scene_arrays(s)["tri_soa"]
    p1 = a[0:3].T
    p2 = p1 + a[3:6].T
    p3 = p1 + a[6:9].T
    covered = np.zeros(s.n_tris, dtype=int)

    def visit(i, plo, phi, depth):
        assert (lo[i] >= plo).all() and (hi[i] <= phi).all()  # children nest inside parents
        if right[i] < 0:
            first, cnt = left[i], -right[i]
            assert 1 <= cnt <= 4
            for t in tri_idx[first:first + cnt]:
                for p in (p1[t], p2[t], p3[t]):
                    assert (p > lo[i]).all() and (p < hi[i]).all()  # strictly inside: the boxes are padded
                covered[t] += 1
            return depth
        assert left[i] > i and right[i] > i
        return max(visit(left[i], lo[i], hi[i], depth + 1), visit(right[i], lo[i], hi[i], depth + 1))

    depth = visit(0, lo[0], hi[0], 1)
    assert (covered == 1).all() and depth <= 48


def test_obj_reader_matches_reference_loader():
    """v / f records, (float)atof positions, 1-based indices, quads fan-split (RayTracer.c:1657-1697)."""
    text = "# c\no x\nv 0.1 0.2 0.3\nv 1.000001 0 0\nv 0 1 0\nv 0 0 1\nvn 0 0 1\nvt 0 0\ns 1\nf 1/1/1 2/1/1 3/1/1\nf 1 2 3 4\n"
    b = R.SceneBuilder()
    b.set_camera(4, 4, 1.0, (0, 0, -5), (0, 0, 0)).set_light((0, 0, 0), (1, 1, 1))
    b.load_obj_text(text)
    a = R.scene_arrays(b.finish(0))
    assert a["tri_soa"].shape == (12, 3)
    f = np.float32
    assert np.array_equal(a["tri_soa"][0:3, 0], np.array([f(0.1), f(0.2), f(0.3)], dtype=np.float64))
    p2 = a["tri_soa"][0:3, 0] + a["tri_soa"][3:6, 0]
    assert np.allclose(p2, [float(f(1.000001)), 0, 0], atol=1e-12)
    # quad 1 2 3 4 -> (1,2,3), (1,3,4)
    assert np.allclose(a["tri_soa"][0:3, 2] + a["tri_soa"][6:9, 2], [0, 0, 1])
    lh = R.lighthouse_triangles()
    assert lh.shape == (1336, 9)


def test_obj_reader_stays_inside_the_length_it_was_given():
    """(text, len) need not be terminated: a buffer that goes on with digits right behind the last face index must
    not leak them into the index; relative indices and an oversized BVH leaf are refused."""
    obj = b"v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3"
    raw = obj + b"99999\nf 9 9 9\n"
    b = R.SceneBuilder()
    b.set_camera(4, 4, 1.0, (0, 0, -5), (0, 0, 0)).set_light((0, 0, 0), (1, 1, 1))
    L = _ffi.host_lib()
    assert L.rt_scene_load_obj(b._b, raw, len(obj), None) == _ffi.RT_OK
    a = R.scene_arrays(b.finish(0))
    assert a["tri_soa"].shape == (12, 1)
    assert np.allclose(a["tri_soa"][0:3, 0] + a["tri_soa"][6:9, 0], [0, 1, 0])  # p3 is vertex 3, not 399999
    assert L.rt_scene_load_obj(b._b, b"f -3 -2 -1\n", 11, None) == _ffi.RT_ERR_INVALID
    out = _ffi.Scene()
    assert L.rt_scene_finish(b._b, 8, C.byref(out)) == _ffi.RT_ERR_TOO_LARGE


def test_teapot_h_mesh_is_the_reference_cube(manifest):
    import refdump
    from conftest import GOLDEN
    _, shapes = refdump.load_npz(f"{GOLDEN}/{manifest['cube_96x64']['scene_dump']}")
    a = R.scene_arrays(R.stock_scene("cube", 96, 64))
    assert a["tri_soa"].shape[1] == 12
    assert np.array_equal(a["tri_soa"][0:3].T, shapes["p1"][:, :3])


def test_ppm_writer_bytes(tmp_path):
    rgb = np.array([[[0, 5, 10], [99, 100, 255]]], dtype=np.uint8)
    assert R.ppm_bytes(rgb) == b"P3\n2 1\n255\n0 5 10 \n99 100 255 \n"
    path = tmp_path / "c.ppm"
    assert _ffi.host_lib().rt_write_ppm(os.fsencode(str(path)), rgb.ctypes.data, 2, 1) == 0
    assert path.read_bytes() == R.ppm_bytes(rgb)


def test_unsupported_scenes_are_rejected():
    b = R.SceneBuilder()
    b.set_camera(4, 4, 1.0, (0, 0, -5), (0, 0, 0)).set_light((0, 0, 0), (1, 1, 1))
    b.add_sphere(np.diag([1e-4, 1e-3, 1.0, 1.0]))  # |det| < 1e-6: the reference's inverse refuses it
    with pytest.raises(RuntimeError):
        b.finish(0)
    b2 = R.SceneBuilder()
    b2.set_light((0, 0, 0), (1, 1, 1))
    with pytest.raises(RuntimeError):
        b2.finish(0)  # no camera


def test_plain_worlds_do_not_carry_the_material_extension():
    """Overriding reflectivity must not turn a sphere / triangle world into one for the general kernel."""
    for name, tr, sr in (("cube", 0.3, 0.0), ("cube_spheres", 0.3, 0.3), ("spheres", 0.0, 0.5)):
        s = R.stock_scene(name, 32, 24, tr, sr)
        assert not s.materials_ext and s.n_shapes == 0
    assert R.stock_scene("refraction", 32, 24).materials_ext


def test_threaded_bvh_build_equals_the_single_threaded_one(monkeypatch):
    """From 65,536 triangles up the host builder hands subtrees to worker threads; the tree (walked from the root)
    and the triangle order must not depend on the number of threads."""
    import ctypes as C

    from raytracer_amd.scene import SceneBuilder
    from test_hip_parity import icosphere

    def build(threads):
        monkeypatch.setenv("RT_HOST_THREADS", str(threads))
        b = SceneBuilder()
        b.set_camera(64, 48, 0.7, (0.3, 1.2, -4.0), (0, 0, 0))
        b.set_light((-4.0, 6.0, -5.0), (1, 1, 1))
        b.add_triangles_f32(icosphere(6))
        s = b.finish(4)
        words = np.ctypeslib.as_array(C.cast(s.bvh_nodes, C.POINTER(C.c_int32)), shape=(s.n_bvh_nodes, 8)).copy()
        order = np.ctypeslib.as_array(s.bvh_tri, shape=(s.n_tris,)).copy()
        walk, stack = [], [0]
        while stack:
            i = stack.pop()
            left, right = int(words[i][3]), int(words[i][7])  # lo[3] left hi[3] right
            walk.append(tuple(words[i][[0, 1, 2, 4, 5, 6]]) + ((left, right) if right < 0 else (-1, 0)))
            if right >= 0:
                assert left > i and right > i  # children follow their parent in the array
                stack += [right, left]
        assert len(walk) == s.n_bvh_nodes  # no node is left unreachable by the splice
        return walk, order

    one, order_one = build(1)
    for threads in (3, 8):
        many, order_many = build(threads)
        assert many == one and np.array_equal(order_many, order_one)
