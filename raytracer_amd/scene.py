"""Thin Python handle on the C scene builder (include/rt_scene.h).

The names follow the reference's constructors (create_camera, view_transform,
create_shape + set_transform, create_triangle, add_shape_to_world;
RayTracer/RayTracer.c:1135, 1501, 968, 1057, 998, 1078) so that scene code
reads like the reference's scene functions.  All arithmetic happens in
librt_host.so.
"""
import ctypes as C
import math
import os

import numpy as np

from . import _ffi
from ._ffi import Material, Scene

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def _d3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


def _m16(m):
    a = np.asarray(m, dtype=np.float64).reshape(16)
    return (C.c_double * 16)(*a.tolist())


def check(rc, what="call"):
    if rc < 0:
        raise RuntimeError(f"{what} failed: rt_status {rc}")
    return rc


def default_material(**kw):
    m = _ffi.host_lib().rt_material_default()
    for k, v in kw.items():
        if k == "color":
            m.color[0], m.color[1], m.color[2] = [float(x) for x in v]
        else:
            setattr(m, k, float(v))
    return m


class Mat4:
    """Row-major 4x4 helpers with the reference's operation order."""

    @staticmethod
    def _call(fn, *args):
        out = (C.c_double * 16)()
        fn(*args, out)
        return np.array(out[:], dtype=np.float64).reshape(4, 4)

    @staticmethod
    def identity():
        return Mat4._call(_ffi.host_lib().rt_mat4_identity)

    @staticmethod
    def translate(x, y, z):
        return Mat4._call(_ffi.host_lib().rt_mat4_translate, x, y, z)

    @staticmethod
    def scale(x, y, z):
        return Mat4._call(_ffi.host_lib().rt_mat4_scale, x, y, z)

    @staticmethod
    def rotate_x(r):
        return Mat4._call(_ffi.host_lib().rt_mat4_rotate_x, r)

    @staticmethod
    def rotate_y(r):
        return Mat4._call(_ffi.host_lib().rt_mat4_rotate_y, r)

    @staticmethod
    def rotate_z(r):
        return Mat4._call(_ffi.host_lib().rt_mat4_rotate_z, r)

    @staticmethod
    def mul(a, b):
        return Mat4._call(_ffi.host_lib().rt_mat4_mul, _m16(a), _m16(b))

    @staticmethod
    def inverse(a):
        out = (C.c_double * 16)()
        ok = _ffi.host_lib().rt_mat4_inverse(_m16(a), out)
        if not ok:
            return None
        return np.array(out[:], dtype=np.float64).reshape(4, 4)

    @staticmethod
    def view_transform(frm, to, up):
        return Mat4._call(_ffi.host_lib().rt_view_transform, _d3(frm), _d3(to), _d3(up))


class SceneBuilder:
    """World + camera under construction; finish() yields the flat rt_scene."""

    def __init__(self):
        self._L = _ffi.host_lib()
        self._b = C.c_void_p(self._L.rt_scene_create())
        if not self._b:
            raise MemoryError("rt_scene_create")
        self._scene = None

    def close(self):
        if self._b:
            self._L.rt_scene_destroy(self._b)
            self._b = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # camera / light -------------------------------------------------------
    def set_camera(self, hsize, vsize, fov, frm, to, up=(0.0, 1.0, 0.0)):
        check(self._L.rt_scene_set_camera(self._b, hsize, vsize, fov, _d3(frm), _d3(to), _d3(up)), "set_camera")
        return self

    def set_camera_view(self, hsize, vsize, fov, view):
        check(self._L.rt_scene_set_camera_view(self._b, hsize, vsize, fov, _m16(view)), "set_camera_view")
        return self

    def set_light(self, position, intensity):
        check(self._L.rt_scene_set_light(self._b, _d3(position), _d3(intensity)), "set_light")
        return self

    # shapes ---------------------------------------------------------------
    def add_sphere(self, transform=None, material=None):
        t = Mat4.identity() if transform is None else transform
        m = C.byref(material) if material is not None else None
        return check(self._L.rt_scene_add_sphere(self._b, _m16(t), m), "add_sphere")

    def add_sphere_ext(self, transform=None, material=None, ext=None, pattern_transform=None):
        """A sphere whose material has a pattern and/or transparency (ext: _ffi.MaterialExt)."""
        t = Mat4.identity() if transform is None else transform
        m = C.byref(material) if material is not None else None
        x = C.byref(ext) if ext is not None else None
        pt = _m16(pattern_transform) if pattern_transform is not None else None
        return check(self._L.rt_scene_add_sphere_ext(self._b, _m16(t), m, x, pt), "add_sphere_ext")

    def add_shape(self, kind, transform=None, material=None, ext=None, pattern_transform=None,
                  minimum=-math.inf, maximum=math.inf, closed=False):
        """kind: 'plane' | 'cube' | 'cylinder' (create_shape, RayTracer.c:968)."""
        ty = {"plane": _ffi.RT_SHAPE_PLANE, "cube": _ffi.RT_SHAPE_CUBE, "cylinder": _ffi.RT_SHAPE_CYLINDER}[kind]
        t = Mat4.identity() if transform is None else transform
        m = C.byref(material) if material is not None else None
        x = C.byref(ext) if ext is not None else None
        pt = _m16(pattern_transform) if pattern_transform is not None else None
        return check(self._L.rt_scene_add_shape(self._b, ty, _m16(t), m, x, pt, minimum, maximum, int(closed)), "add_shape")

    def add_triangle(self, p1, p2, p3, material=None, ext=None, pattern_transform=None):
        m = C.byref(material) if material is not None else None
        x = C.byref(ext) if ext is not None else None
        pt = _m16(pattern_transform) if pattern_transform is not None else None
        return check(self._L.rt_scene_add_triangle_ext(self._b, _d3(p1), _d3(p2), _d3(p3), m, x, pt), "add_triangle")

    def add_triangles_f32(self, xyz9, material=None):
        a = np.ascontiguousarray(xyz9, dtype=np.float32).reshape(-1, 9)
        m = C.byref(material) if material is not None else None
        check(
            self._L.rt_scene_add_triangles_f32(self._b, a.ctypes.data_as(C.POINTER(C.c_float)), a.shape[0], m),
            "add_triangles_f32",
        )
        return self

    def load_obj_text(self, text, material=None):
        raw = text.encode() if isinstance(text, str) else bytes(text)
        m = C.byref(material) if material is not None else None
        check(self._L.rt_scene_load_obj(self._b, raw, len(raw), m), "load_obj")
        return self

    def load_obj_file(self, path, material=None):
        m = C.byref(material) if material is not None else None
        check(self._L.rt_scene_load_obj_file(self._b, os.fsencode(path), m), "load_obj_file")
        return self

    def set_reflective(self, tri_reflective=0.0, sph_reflective=0.0):
        check(self._L.rt_scene_set_reflective(self._b, tri_reflective, sph_reflective), "set_reflective")
        return self

    def stock(self, name, hsize, vsize):
        check(self._L.rt_scene_stock(self._b, name.encode(), hsize, vsize), f"stock scene {name}")
        return self

    # result ---------------------------------------------------------------
    def finish(self, bvh_leaf_size=4):
        s = Scene()
        check(self._L.rt_scene_finish(self._b, bvh_leaf_size, C.byref(s)), "rt_scene_finish")
        s._owner = self  # arrays live in the builder
        self._scene = s
        return s


def lighthouse_triangles():
    """The 1336 triangles of the reference's lighthouse mesh as float32 (p1 p2 p3), in file order.

    Dumped from the reference's own loader (load_model_file, RayTracer.c:1621) by
    tests/golden/make_golden.py; shipped as data because the mesh file itself is not.
    """
    path = os.path.join(DATA_DIR, "lighthouse_tris_f32.bin")
    return np.fromfile(path, dtype="<f4").reshape(-1, 9)


def material_ext(**kw):
    """Pattern / transparency half of a material, defaults of create_material_default (RayTracer.c:712-717)."""
    x = _ffi.host_lib().rt_material_ext_default()
    for k, v in kw.items():
        if k in ("pattern_from", "pattern_to"):
            getattr(x, k)[:] = list(v)
        else:
            setattr(x, k, v)
    return x


def stock_scene(name, hsize, vsize, tri_reflective=0.0, sph_reflective=0.0, bvh_leaf_size=4):
    """The reference's scenes at any size: 'lighthouse', 'cube' (the teapot.h mesh), 'spheres', 'cube_spheres',
    'refraction', 'world_with_plane', 'dual_spheres'."""
    b = SceneBuilder()
    b.stock(name, hsize, vsize)
    if name == "lighthouse":
        b.add_triangles_f32(lighthouse_triangles())
    if tri_reflective or sph_reflective:
        b.set_reflective(tri_reflective, sph_reflective)
    return b.finish(bvh_leaf_size)


def scene_arrays(s):
    """numpy views of a flat scene (for tests and tools)."""
    ns, nt = s.n_spheres, s.n_tris
    out = {
        "inv_view": np.array(s.camera.inv_view[:]).reshape(4, 4),
        "sph_inv": np.ctypeslib.as_array(s.sph_inv, shape=(max(ns, 1), 16))[:ns].reshape(ns, 4, 4).copy(),
        "sph_order": np.ctypeslib.as_array(s.sph_order, shape=(max(ns, 1),))[:ns].copy(),
        "sph_material": np.ctypeslib.as_array(s.sph_material, shape=(max(ns, 1),))[:ns].copy(),
        "tri_soa": np.ctypeslib.as_array(s.tri_soa, shape=(12, max(nt, 1)))[:, :nt].copy() if nt else np.zeros((12, 0)),
        "tri_order": np.ctypeslib.as_array(s.tri_order, shape=(max(nt, 1),))[:nt].copy(),
        "tri_material": np.ctypeslib.as_array(s.tri_material, shape=(max(nt, 1),))[:nt].copy(),
        "first_object_inv": np.array(s.first_object_inv[:]).reshape(4, 4),
        "shapes": [s.shapes[i] for i in range(s.n_shapes)],
        "materials_ext": [s.materials_ext[i] for i in range(s.n_materials)] if s.materials_ext else None,
    }
    return out


__all__ = [
    "SceneBuilder",
    "Mat4",
    "default_material",
    "stock_scene",
    "material_ext",
    "scene_arrays",
    "lighthouse_triangles",
    "Material",
    "Scene",
]
_ = math
