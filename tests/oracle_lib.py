"""ctypes handle on oracle/librt_oracle.so (TEST INFRASTRUCTURE: the CPU restatement of the reference path)."""
import ctypes as C
import os

import numpy as np

from raytracer_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "librt_oracle.so")


class Ray(C.Structure):
    _fields_ = [("origin", C.c_double * 4), ("direction", C.c_double * 4)]


class Probe(C.Structure):
    _fields_ = [("max_list", C.c_uint64), ("exact_zero_t", C.c_uint64), ("list_overflow", C.c_uint64)]


_dp = C.POINTER(C.c_double)
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            raise RuntimeError(f"{ORACLE_SO} missing: make -C oracle")
        L = C.CDLL(ORACLE_SO)
        SP, PP, ST = C.POINTER(_ffi.Scene), C.POINTER(_ffi.RenderParams), C.POINTER(_ffi.Stats)
        L.rto_render_f64.argtypes = [SP, PP, C.c_void_p, ST]
        L.rto_render_rgb.argtypes = [SP, PP, C.c_void_p, ST]
        L.rto_get_probe.argtypes = [C.POINTER(Probe)]
        L.rto_get_probe.restype = None
        L.rto_color_convert.argtypes = [C.c_double]
        L.rto_ray_for_pixel.argtypes = [C.POINTER(_ffi.Camera), C.c_double, C.c_double, C.POINTER(Ray)]
        L.rto_ray_for_pixel.restype = None
        L.rto_intersect_sphere.argtypes = [_dp, C.POINTER(Ray), _dp]
        L.rto_intersect_triangle.argtypes = [_dp, _dp, _dp, C.POINTER(Ray), _dp]
        L.rto_normal_at_sphere.argtypes = [_dp, _dp, _dp]
        L.rto_normal_at_sphere.restype = None
        L.rto_reflect.argtypes = [_dp, _dp, _dp]
        L.rto_reflect.restype = None
        L.rto_lighting.argtypes = [C.POINTER(_ffi.Material), C.POINTER(_ffi.Light), _dp, _dp, _dp, C.c_int, _dp]
        L.rto_lighting.restype = None
        L.rto_hit_world.argtypes = [SP, C.POINTER(Ray), _dp, C.POINTER(C.c_int), _dp, C.c_int]
        L.rto_is_shadowed.argtypes = [SP, _dp]
        L.rto_color_at.argtypes = [SP, C.POINTER(Ray), C.c_int, _dp]
        L.rto_color_at.restype = None
        L.rto_prepare.argtypes = [SP, C.POINTER(Ray), C.c_int, C.c_double, _dp]
        L.rto_intersect_shape.argtypes = [C.POINTER(_ffi.Shape), C.POINTER(Ray), _dp, C.c_int]
        L.rto_normal_at_shape.argtypes = [C.POINTER(_ffi.Shape), _dp, _dp]
        L.rto_normal_at_shape.restype = None
        L.rto_pattern_at.argtypes = [C.POINTER(_ffi.MaterialExt), _dp, _dp]
        L.rto_pattern_at.restype = None
        L.rto_schlick.argtypes = [_dp, _dp, C.c_double, C.c_double]
        L.rto_schlick.restype = C.c_double
        L.rto_refractive_indices.argtypes = [SP, C.POINTER(Ray), C.c_int, _dp]
        L.rto_hit_list.argtypes = [_dp, C.c_int]
        L.rto_list_capacity.argtypes = [C.c_int]
        L.rto_write_ppm.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
        _lib = L
    return _lib


def d3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


def m16(m):
    return (C.c_double * 16)(*np.asarray(m, dtype=np.float64).reshape(16).tolist())


def ray(origin, direction):
    r = Ray()
    r.origin[:] = [float(x) for x in origin] + [1.0]
    r.direction[:] = [float(x) for x in direction] + [0.0]
    return r


def params(max_depth=5, spp_grid=1, accel=_ffi.RT_ACCEL_AUTO, count_rays=1, band_rows=1, n_ranks=1, rank=0):
    return _ffi.RenderParams(max_depth, spp_grid, accel, count_rays, band_rows, n_ranks, rank, 0)


def rank_rows(vsize, band_rows, n_ranks, rank):
    return sum(1 for y in range(vsize) if n_ranks <= 1 or (y // band_rows) % n_ranks == rank)


def render_f64(scene, **kw):
    p = params(**kw)
    rows = rank_rows(scene.camera.vsize, p.band_rows, p.n_ranks, p.rank)
    out = np.zeros((rows, scene.camera.hsize, 3), dtype=np.float64)
    st = _ffi.Stats()
    rc = lib().rto_render_f64(C.byref(scene), C.byref(p), out.ctypes.data, C.byref(st))
    assert rc == 0, rc
    return out, st


def render_rgb(scene, **kw):
    p = params(**kw)
    rows = rank_rows(scene.camera.vsize, p.band_rows, p.n_ranks, p.rank)
    out = np.zeros((rows, scene.camera.hsize, 3), dtype=np.uint8)
    st = _ffi.Stats()
    rc = lib().rto_render_rgb(C.byref(scene), C.byref(p), out.ctypes.data, C.byref(st))
    assert rc == 0, rc
    return out, st


def probe():
    p = Probe()
    lib().rto_get_probe(C.byref(p))
    return p
