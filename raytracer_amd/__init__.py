"""raytracer_amd: MI355X-native trace loop behind the render() seam of steven-schronk/RayTracer.

The product is the C-ABI library (include/rt_hip.h, raytracer_amd/csrc) and the
host C helpers (include/rt_scene.h, raytracer_amd/host).  This package is the
Python plumbing used by tests, bench.py and the multi-GPU gather.
"""
from . import _ffi
from .render import HipRenderer, MultiRenderer, RtError, gather_offset, make_params, ppm_bytes, rank_row_map
from .scene import Mat4, SceneBuilder, default_material, lighthouse_triangles, material_ext, scene_arrays, stock_scene

__all__ = [
    "_ffi", "HipRenderer", "RtError", "make_params", "ppm_bytes", "rank_row_map",
    "Mat4", "SceneBuilder", "default_material", "lighthouse_triangles", "scene_arrays", "stock_scene",
]
