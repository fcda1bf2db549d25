"""Python handle on the C ABI of librt_hip.so (include/rt_hip.h).

Plumbing only: every call goes straight to the HIP library; there is no CPU
fallback and a missing library or device raises.
"""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import RenderParams, Stats


class RtError(RuntimeError):
    def __init__(self, what, rc):
        L = _ffi.hip_lib()
        msg = L.rt_hip_strerror(rc).decode()
        detail = L.rt_hip_last_error().decode()
        super().__init__(f"{what}: {msg} ({rc})" + (f" [{detail}]" if detail else ""))
        self.status = rc


def _check(rc, what):
    if rc != _ffi.RT_OK:
        raise RtError(what, rc)


def make_params(max_depth=5, spp_grid=1, accel=_ffi.RT_ACCEL_AUTO, count_rays=False, band_rows=1, n_ranks=1, rank=0,
                flags=0, frames_overlap=False, no_tune=False, count_as_run=False):
    """rt_render_params; frames_overlap=True states that the caller keeps several frames in flight on different
    streams (RT_RENDER_FRAMES_OVERLAP); no_tune=True that the tile plan is to be used as it stands
    (RT_RENDER_NO_TUNE: a one-frame caller)."""
    if frames_overlap:
        flags |= _ffi.RT_RENDER_FRAMES_OVERLAP
    if no_tune:
        flags |= _ffi.RT_RENDER_NO_TUNE
    if count_as_run:  # with count_rays: the counters of what the production kernel executes (RT_RENDER_COUNT_AS_RUN)
        flags |= _ffi.RT_RENDER_COUNT_AS_RUN
    return RenderParams(max_depth, spp_grid, accel, 1 if count_rays else 0, band_rows, n_ranks, rank, flags)


class HipRenderer:
    """One process, one GPU: init -> upload(scene) -> render*()."""

    def __init__(self, device=0):
        self._L = _ffi.hip_lib()
        _check(self._L.rt_hip_init(int(device)), "rt_hip_init")
        self.device = int(device)
        self._scene = None
        self._tuned = set()

    def upload(self, scene):
        _check(self._L.rt_hip_upload_scene(C.byref(scene)), "rt_hip_upload_scene")
        self._scene = scene  # keeps the host arrays alive; the device copy is independent
        self.width, self.height = scene.camera.hsize, scene.camera.vsize
        self._tuned = set()
        return self

    def build_bvh_device(self, leaf_size=4):
        """Replace the scene's BVH by one built on the GPU; returns (inner nodes, depth, build ms)."""
        info = _ffi.BvhDeviceInfo()
        _check(self._L.rt_hip_build_bvh_device(int(leaf_size), C.byref(info)), "rt_hip_build_bvh_device")
        self._tuned = set()
        return info.n_nodes, info.depth, info.build_ms

    def scene_counts(self):
        """(spheres, triangles) of the uploaded scene."""
        return (self._scene.n_spheres, self._scene.n_tris) if self._scene is not None else (0, 0)

    def rank_rows(self, p):
        n = self._L.rt_hip_rank_rows(self.height, p.band_rows, p.n_ranks, p.rank)
        if n < 0:
            raise RtError("rt_hip_rank_rows", n)
        return n

    def render(self, params=None, stats=True, **kw):
        """Host image: uint8 [rows, width, 3] of this rank's rows."""
        p = params or make_params(**kw)
        out = np.empty((self.rank_rows(p), self.width, 3), dtype=np.uint8)
        st = Stats()
        _check(self._L.rt_hip_render(C.byref(p), out.ctypes.data, C.byref(st) if stats else None), "rt_hip_render")
        return out, st

    def render_f64(self, params=None, **kw):
        p = params or make_params(**kw)
        out = np.empty((self.rank_rows(p), self.width, 3), dtype=np.float64)
        st = Stats()
        _check(self._L.rt_hip_render_f64(C.byref(p), out.ctypes.data, C.byref(st)), "rt_hip_render_f64")
        return out, st

    def tune(self, params=None, **kw):
        """Make and tune the tile plan for these parameters now (rt_hip_tune), so that render_device() without
        stats only enqueues from its first frame on."""
        p = params or make_params(**kw)
        _check(self._L.rt_hip_tune(C.byref(p)), "rt_hip_tune")
        self._tuned.add(self._plan_key(p))

    @staticmethod
    def _plan_key(p):
        return (p.max_depth, p.spp_grid, p.accel, p.band_rows, p.n_ranks, p.rank, p.flags & ~_ffi.RT_RENDER_NO_TUNE)

    def render_device(self, data_ptr, params, stream=None, stats=None):
        """Enqueue a render into device memory at `data_ptr` (e.g. a torch uint8 tensor's data_ptr()).  Without
        stats the C call only enqueues and uses the tile plan at hand; this wrapper makes that plan first
        (tune(), once per scene and parameter set) unless the parameters carry RT_RENDER_NO_TUNE."""
        if stats is None and not (params.flags & _ffi.RT_RENDER_NO_TUNE) and self._plan_key(params) not in self._tuned:
            self.tune(params)
        _check(
            self._L.rt_hip_render_device(
                C.byref(params), C.c_void_p(data_ptr), C.c_void_p(stream or 0), C.byref(stats) if stats is not None else None
            ),
            "rt_hip_render_device",
        )

    # ---- write_canvas_to_file's P3 text (RayTracer.c:1333-1348), formatted on the device ----
    def ppm_capacity(self, width, height):
        return int(self._L.rt_hip_ppm_capacity(width, height))

    def format_ppm_device(self, rgb_ptr, width, height, text_ptr, capacity, stream=None, wait=True):
        """Text of the image at device address rgb_ptr into device memory at text_ptr; returns its length
        (None with wait=False: ask ppm_last_length() later)."""
        n = C.c_size_t(0)
        _check(self._L.rt_hip_format_ppm_device(C.c_void_p(rgb_ptr), width, height, C.c_void_p(text_ptr), capacity,
                                                C.byref(n) if wait else None, C.c_void_p(stream or 0)),
               "rt_hip_format_ppm_device")
        return int(n.value) if wait else None

    def format_ppm_device_noalloc(self, rgb_ptr, width, height, text_ptr, capacity, length_ptr, stream=None):
        """The allocation-free variant: the text length goes to the 8-byte device word at length_ptr; never waits."""
        _check(self._L.rt_hip_format_ppm_device_noalloc(C.c_void_p(rgb_ptr), width, height, C.c_void_p(text_ptr), capacity,
                                                        C.c_void_p(length_ptr), C.c_void_p(stream or 0)),
               "rt_hip_format_ppm_device_noalloc")

    def ppm_last_length(self):
        n = C.c_size_t(0)
        _check(self._L.rt_hip_ppm_last_length(C.byref(n)), "rt_hip_ppm_last_length")
        return int(n.value)

    def format_ppm(self, rgb_ptr, width, height):
        """bytes of canvas.ppm for the 8-bit image at device address rgb_ptr."""
        cap = self.ppm_capacity(width, height)
        buf = np.empty(cap, dtype=np.uint8)
        n = C.c_size_t(0)
        _check(self._L.rt_hip_format_ppm(C.c_void_p(rgb_ptr), width, height, buf.ctypes.data, cap, C.byref(n)),
               "rt_hip_format_ppm")
        return buf[: n.value].tobytes()

    def render_ppm(self, params=None, **kw):
        """render + format on the device: (bytes of canvas.ppm, stats)."""
        p = params or make_params(**kw)
        cap = self.ppm_capacity(self.width, self.height)
        buf = np.empty(cap, dtype=np.uint8)
        n = C.c_size_t(0)
        st = Stats()
        _check(self._L.rt_hip_render_ppm(C.byref(p), buf.ctypes.data, cap, C.byref(n), C.byref(st)), "rt_hip_render_ppm")
        return buf[: n.value].tobytes(), st

    def place_bands_device(self, gathered_ptr, out_ptr, width, height, band_rows, n_ranks, stream=None):
        """Gathered buffer (every rank's packed rows, rank after rank) -> image order, device to device."""
        _check(self._L.rt_hip_place_bands_device(C.c_void_p(gathered_ptr), C.c_void_p(out_ptr), width, height, band_rows,
                                                 n_ranks, C.c_void_p(stream or 0)), "rt_hip_place_bands_device")

    def render_ppm_submit(self, slot, params=None, **kw):
        """Enqueue render + formatting of a frame into slot 0 or 1 (returns at once)."""
        p = params or make_params(**kw)
        _check(self._L.rt_hip_render_ppm_submit(C.byref(p), int(slot)), "rt_hip_render_ppm_submit")

    def render_ppm_collect(self, slot, copy=True):
        """Wait for the slot's frame: (bytes of canvas.ppm, stats); copy=False returns its length only (timing)."""
        ptr, n, st = C.c_void_p(), C.c_size_t(0), Stats()
        _check(self._L.rt_hip_render_ppm_collect(int(slot), C.byref(ptr), C.byref(n), C.byref(st)), "rt_hip_render_ppm_collect")
        return (C.string_at(ptr.value, n.value) if copy else n.value), st

    def timer_start(self, stream=None):
        _check(self._L.rt_hip_event_record(0, C.c_void_p(stream or 0)), "rt_hip_event_record")

    def timer_stop(self, stream=None):
        _check(self._L.rt_hip_event_record(1, C.c_void_p(stream or 0)), "rt_hip_event_record")

    def timer_ms(self):
        ms = C.c_double()
        _check(self._L.rt_hip_event_elapsed_ms(C.byref(ms)), "rt_hip_event_elapsed_ms")
        return ms.value

    def shutdown(self):
        self._L.rt_hip_shutdown()


class MultiRenderer:
    """Several GPUs driven by this one process (rt_hip_multi_*): bands rendered on every device, gathered on
    device 0 with RCCL, one image back.  n_devices=0: every visible device."""

    def __init__(self, n_devices=0):
        self._L = _ffi.hip_lib()
        _check(self._L.rt_hip_multi_init(int(n_devices)), "rt_hip_multi_init")
        self.n_devices = int(self._L.rt_hip_multi_devices())
        self._scene = None

    def upload(self, scene):
        _check(self._L.rt_hip_multi_upload_scene(C.byref(scene)), "rt_hip_multi_upload_scene")
        self._scene = scene
        self.width, self.height = scene.camera.hsize, scene.camera.vsize
        return self

    def set_chunks(self, chunks):
        """In how many parts a device's rows travel to device 0 (0: the library decides per frame)."""
        _check(self._L.rt_hip_multi_set_chunks(int(chunks)), "rt_hip_multi_set_chunks")

    def render(self, params=None, stats=True, **kw):
        p = params or make_params(**kw)
        out = np.empty((self.height, self.width, 3), dtype=np.uint8)
        st = Stats()
        _check(self._L.rt_hip_render_multi(C.byref(p), out.ctypes.data, C.byref(st) if stats else None), "rt_hip_render_multi")
        return out, st

    def shutdown(self):
        self._L.rt_hip_shutdown()


def gather_offset(width, height, band_rows, n_ranks, rank):
    """Byte offset of `rank`'s packed rows in a buffer holding every rank's rows, rank after rank."""
    return int(_ffi.hip_lib().rt_hip_gather_offset(width, height, band_rows, n_ranks, rank))


def rank_row_map(height, band_rows, n_ranks, rank):
    """Image rows owned by `rank`, in packed order."""
    L = _ffi.hip_lib()
    n = L.rt_hip_rank_rows(height, band_rows, n_ranks, rank)
    return np.array([L.rt_hip_rank_row_to_y(i, band_rows, n_ranks, rank) for i in range(n)], dtype=np.int64)


def ppm_bytes(rgb):
    """ASCII P3 bytes identical to the reference's write_canvas_to_file (RayTracer.c:1333)."""
    a = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, _ = a.shape
    L = _ffi.host_lib()
    need = L.rt_format_ppm(None, 0, a.ctypes.data, w, h)
    buf = C.create_string_buffer(need)
    got = L.rt_format_ppm(buf, need, a.ctypes.data, w, h)
    return buf.raw[:got]
