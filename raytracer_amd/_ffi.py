"""ctypes mirror of include/rt_hip.h and include/rt_scene.h.

Plumbing only: the structs below are the C structs of the drop-in boundary,
field for field; nothing here computes anything.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")

RT_OK = 0
RT_ERR_INVALID = -1
RT_ERR_NO_DEVICE = -2
RT_ERR_HIP = -3
RT_ERR_UNSUPPORTED = -4
RT_ERR_NOT_READY = -5
RT_ERR_TOO_LARGE = -6

RT_ACCEL_AUTO = 0
RT_ACCEL_BRUTE = 1
RT_ACCEL_BVH = 2
RT_ACCEL_REFERENCE = 3

RT_RENDER_FRAMES_OVERLAP = 1
RT_RENDER_NO_TUNE = 2
RT_RENDER_COUNT_AS_RUN = 4


class Material(C.Structure):
    _fields_ = [
        ("color", C.c_double * 3),
        ("ambient", C.c_double),
        ("diffuse", C.c_double),
        ("specular", C.c_double),
        ("shininess", C.c_double),
        ("reflective", C.c_double),
    ]


RT_PATTERN_NONE, RT_PATTERN_TEST, RT_PATTERN_STRIPE, RT_PATTERN_GRADIENT, RT_PATTERN_RING, RT_PATTERN_CHECKER = -1, 0, 1, 2, 3, 4
RT_SHAPE_PLANE, RT_SHAPE_CUBE, RT_SHAPE_CYLINDER = 0, 3, 4


class MaterialExt(C.Structure):
    _fields_ = [
        ("transparency", C.c_double),
        ("refractive_index", C.c_double),
        ("pattern_from", C.c_double * 3),
        ("pattern_to", C.c_double * 3),
        ("pattern_inv", C.c_double * 16),
        ("pattern_type", C.c_int32),
        ("reserved", C.c_int32),
    ]


class Shape(C.Structure):
    _fields_ = [
        ("inv", C.c_double * 16),
        ("minimum", C.c_double),
        ("maximum", C.c_double),
        ("type", C.c_int32),
        ("closed", C.c_int32),
        ("material", C.c_int32),
        ("order", C.c_int32),
    ]


class Camera(C.Structure):
    _fields_ = [
        ("hsize", C.c_int32),
        ("vsize", C.c_int32),
        ("pixel_size", C.c_double),
        ("half_width", C.c_double),
        ("half_height", C.c_double),
        ("inv_view", C.c_double * 16),
    ]


class Light(C.Structure):
    _fields_ = [("position", C.c_double * 3), ("intensity", C.c_double * 3)]


class BvhNode(C.Structure):
    _fields_ = [
        ("lo", C.c_float * 3),
        ("left", C.c_int32),
        ("hi", C.c_float * 3),
        ("right", C.c_int32),
    ]


class Scene(C.Structure):
    _fields_ = [
        ("camera", Camera),
        ("light", Light),
        ("n_materials", C.c_int32),
        ("materials", C.POINTER(Material)),
        ("n_spheres", C.c_int32),
        ("sph_inv", C.POINTER(C.c_double)),
        ("sph_material", C.POINTER(C.c_int32)),
        ("sph_order", C.POINTER(C.c_int32)),
        ("n_tris", C.c_int32),
        ("tri_soa", C.POINTER(C.c_double)),
        ("tri_material", C.POINTER(C.c_int32)),
        ("tri_order", C.POINTER(C.c_int32)),
        ("n_bvh_nodes", C.c_int32),
        ("bvh_nodes", C.POINTER(BvhNode)),
        ("bvh_tri", C.POINTER(C.c_int32)),
        ("n_shapes", C.c_int32),
        ("shapes", C.POINTER(Shape)),
        ("materials_ext", C.POINTER(MaterialExt)),
        ("first_object_inv", C.c_double * 16),
    ]


class RenderParams(C.Structure):
    _fields_ = [
        ("max_depth", C.c_int32),
        ("spp_grid", C.c_int32),
        ("accel", C.c_int32),
        ("count_rays", C.c_int32),
        ("band_rows", C.c_int32),
        ("n_ranks", C.c_int32),
        ("rank", C.c_int32),
        ("flags", C.c_int32),
    ]


class BvhDeviceInfo(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("depth", C.c_int32), ("build_ms", C.c_double)]


class Stats(C.Structure):
    _fields_ = [
        ("rays_primary", C.c_uint64),
        ("rays_shadow", C.c_uint64),
        ("rays_secondary", C.c_uint64),
        ("prim_tests", C.c_uint64),
        ("node_tests", C.c_uint64),
        ("sphere_tests", C.c_uint64),
        ("kernel_ms", C.c_double),
        ("rows_rendered", C.c_int32),
        ("accel_used", C.c_int32),
        ("tri_exact", C.c_uint64),
        ("sphere_exact", C.c_uint64),
        ("exchange_ms", C.c_double),
        ("zero_t_entries", C.c_uint64),
        ("list_cap_rays", C.c_uint64),
    ]

    @property
    def rays(self):
        return self.rays_primary + self.rays_shadow + self.rays_secondary


_dp = C.POINTER(C.c_double)
_host = None
_hip = None


def host_lib():
    """librt_host.so: scene flattening, BVH build, PPM writer (plain C, no GPU needed)."""
    global _host
    if _host is None:
        path = os.path.join(LIB_DIR, "librt_host.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make -C raytracer_amd` (or __graft_entry__.build())")
        L = C.CDLL(path)
        L.rt_scene_create.restype = C.c_void_p
        L.rt_scene_destroy.argtypes = [C.c_void_p]
        L.rt_scene_destroy.restype = None
        L.rt_scene_set_camera.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, _dp, _dp, _dp]
        L.rt_scene_set_camera_view.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, _dp]
        L.rt_scene_set_light.argtypes = [C.c_void_p, _dp, _dp]
        L.rt_scene_add_sphere.argtypes = [C.c_void_p, _dp, C.POINTER(Material)]
        L.rt_scene_add_sphere_ext.argtypes = [C.c_void_p, _dp, C.POINTER(Material), C.POINTER(MaterialExt), _dp]
        L.rt_scene_add_shape.argtypes = [C.c_void_p, C.c_int, _dp, C.POINTER(Material), C.POINTER(MaterialExt), _dp,
                                         C.c_double, C.c_double, C.c_int]
        L.rt_scene_add_triangle_ext.argtypes = [C.c_void_p, _dp, _dp, _dp, C.POINTER(Material), C.POINTER(MaterialExt), _dp]
        L.rt_material_ext_default.restype = MaterialExt
        L.rt_scene_add_triangle.argtypes = [C.c_void_p, _dp, _dp, _dp, C.POINTER(Material)]
        L.rt_scene_add_triangles_f32.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.POINTER(Material)]
        L.rt_scene_load_obj.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(Material)]
        L.rt_scene_load_obj_file.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(Material)]
        L.rt_scene_set_reflective.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.rt_scene_finish.argtypes = [C.c_void_p, C.c_int, C.POINTER(Scene)]
        L.rt_scene_stock.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
        L.rt_material_default.restype = Material
        L.rt_mat4_inverse.argtypes = [_dp, _dp]
        L.rt_mat4_mul.argtypes = [_dp, _dp, _dp]
        L.rt_mat4_mul.restype = None
        for name in ("rt_mat4_translate", "rt_mat4_scale"):
            getattr(L, name).argtypes = [C.c_double, C.c_double, C.c_double, _dp]
            getattr(L, name).restype = None
        for name in ("rt_mat4_rotate_x", "rt_mat4_rotate_y", "rt_mat4_rotate_z"):
            getattr(L, name).argtypes = [C.c_double, _dp]
            getattr(L, name).restype = None
        L.rt_mat4_identity.argtypes = [_dp]
        L.rt_mat4_identity.restype = None
        L.rt_view_transform.argtypes = [_dp, _dp, _dp, _dp]
        L.rt_view_transform.restype = None
        L.rt_write_ppm.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
        L.rt_format_ppm.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int]
        L.rt_format_ppm.restype = C.c_size_t
        _host = L
    return _host


def hip_lib():
    """librt_hip.so: the C-ABI over the gfx950 kernels.  No fallback: missing library is an error."""
    global _hip
    if _hip is None:
        tag = os.environ.get("RT_HIP_LIB")  # development: an alternative build of the same library
        path = os.path.join(LIB_DIR, f"librt_hip_{tag}.so" if tag else "librt_hip.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make -C raytracer_amd` (or __graft_entry__.build())")
        # PyTorch ships its own libamdhip64.so.7.  A process must hold ONE HIP runtime: load torch's first so
        # that this library's NEEDED libamdhip64.so.7 resolves to the copy torch (device memory, RCCL) uses.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(path)
        L.rt_hip_init.argtypes = [C.c_int]
        L.rt_hip_set_device.argtypes = [C.c_int]
        L.rt_hip_upload_scene.argtypes = [C.POINTER(Scene)]
        L.rt_hip_build_bvh_device.argtypes = [C.c_int, C.POINTER(BvhDeviceInfo)]
        L.rt_hip_render.argtypes = [C.POINTER(RenderParams), C.c_void_p, C.POINTER(Stats)]
        L.rt_hip_render_device.argtypes = [C.POINTER(RenderParams), C.c_void_p, C.c_void_p, C.POINTER(Stats)]
        L.rt_hip_render_f64.argtypes = [C.POINTER(RenderParams), C.c_void_p, C.POINTER(Stats)]
        L.rt_hip_tune.argtypes = [C.POINTER(RenderParams)]
        L.rt_hip_multi_init.argtypes = [C.c_int]
        L.rt_hip_multi_upload_scene.argtypes = [C.POINTER(Scene)]
        L.rt_hip_render_multi.argtypes = [C.POINTER(RenderParams), C.c_void_p, C.POINTER(Stats)]
        L.rt_hip_multi_set_chunks.argtypes = [C.c_int]
        L.rt_hip_gather_offset.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.rt_hip_gather_offset.restype = C.c_size_t
        L.rt_hip_place_bands_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.rt_hip_rank_rows.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
        L.rt_hip_rank_row_to_y.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
        L.rt_hip_event_record.argtypes = [C.c_int, C.c_void_p]
        L.rt_hip_event_elapsed_ms.argtypes = [C.POINTER(C.c_double)]
        L.rt_hip_ppm_capacity.argtypes = [C.c_int, C.c_int]
        L.rt_hip_ppm_capacity.restype = C.c_size_t
        L.rt_hip_format_ppm_device.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                               C.POINTER(C.c_size_t), C.c_void_p]
        L.rt_hip_ppm_last_length.argtypes = [C.POINTER(C.c_size_t)]
        L.rt_hip_format_ppm_device_noalloc.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p,
                                                       C.c_void_p]
        L.rt_hip_format_ppm.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.rt_hip_render_ppm.argtypes = [C.POINTER(RenderParams), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                        C.POINTER(Stats)]
        L.rt_hip_render_ppm_submit.argtypes = [C.POINTER(RenderParams), C.c_int]
        L.rt_hip_render_ppm_collect.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(Stats)]
        L.rt_hip_shutdown.restype = None
        L.rt_hip_strerror.argtypes = [C.c_int]
        L.rt_hip_strerror.restype = C.c_char_p
        L.rt_hip_last_error.restype = C.c_char_p
        _hip = L
    return _hip


HIP_EXPORTS = [
    "rt_hip_init",
    "rt_hip_set_device",
    "rt_hip_upload_scene",
    "rt_hip_build_bvh_device",
    "rt_hip_render",
    "rt_hip_render_device",
    "rt_hip_render_f64",
    "rt_hip_tune",
    "rt_hip_ppm_capacity",
    "rt_hip_format_ppm_device",
    "rt_hip_ppm_last_length",
    "rt_hip_format_ppm_device_noalloc",
    "rt_hip_format_ppm",
    "rt_hip_render_ppm",
    "rt_hip_render_ppm_submit",
    "rt_hip_render_ppm_collect",
    "rt_hip_device_count",
    "rt_hip_multi_init",
    "rt_hip_multi_devices",
    "rt_hip_multi_upload_scene",
    "rt_hip_render_multi",
    "rt_hip_multi_set_chunks",
    "rt_hip_gather_offset",
    "rt_hip_place_bands_device",
    "rt_hip_rank_rows",
    "rt_hip_rank_row_to_y",
    "rt_hip_event_record",
    "rt_hip_event_elapsed_ms",
    "rt_hip_shutdown",
    "rt_hip_strerror",
    "rt_hip_last_error",
]
