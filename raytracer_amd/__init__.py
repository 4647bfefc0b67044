"""raytracer_amd -- MI355X (gfx950) render core for nealsid/RayTracer scenes.

Python is harness only: the product is `librtmi.so` (HIP kernels + the C ABI declared in
include/rtmi.h).  This module binds that ABI with ctypes.  There is no CPU fallback: if the
shared library is missing or no gfx950 device is present, calls raise.
"""
import ctypes as C
import os

import numpy as np

from . import _abi, scenes  # noqa: F401
from ._abi import Camera, Counters, Params, SceneDesc  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# RTMI_LIB_PATH: another build of the library (kernel experiments); the default is the in-tree build
LIB_PATH = os.environ.get("RTMI_LIB_PATH") or os.path.join(_HERE, "librtmi.so")
_lib = None


class RtmiError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        super().__init__(f"{where}: status {status}" + (f" ({detail})" if detail else ""))


def lib():
    """The loaded librtmi.so with argtypes set; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C raytracer_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.rtmi_abi_version.restype = C.c_int32
        L.rtmi_status_string.restype = C.c_char_p
        L.rtmi_status_string.argtypes = [C.c_int32]
        L.rtmi_last_error.restype = C.c_char_p
        L.rtmi_device_count.restype = C.c_int32
        L.rtmi_samples_per_pixel.restype = C.c_int32
        L.rtmi_samples_per_pixel.argtypes = [C.c_double]
        L.rtmi_bounding_sphere.restype = C.c_int32
        L.rtmi_bounding_sphere.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_double),
                                           C.POINTER(C.c_double)]
        L.rtmi_scene_create.restype = C.c_int32
        L.rtmi_scene_create.argtypes = [C.POINTER(SceneDesc), C.c_int32, C.POINTER(C.c_void_p)]
        L.rtmi_scene_destroy.restype = None
        L.rtmi_scene_destroy.argtypes = [C.c_void_p]
        L.rtmi_render.restype = C.c_int32
        L.rtmi_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Params), C.c_void_p,
                                  C.c_void_p, C.POINTER(Counters)]
        L.rtmi_render_device.restype = C.c_int32
        L.rtmi_render_device.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Params),
                                         C.c_void_p, C.c_void_p, C.POINTER(Counters)]
        L.rtmi_render_begin.restype = C.c_int32
        L.rtmi_render_begin.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Params), C.c_void_p, C.c_void_p]
        L.rtmi_render_end.restype = C.c_int32
        L.rtmi_render_end.argtypes = [C.c_void_p, C.POINTER(Counters)]
        L.rtmi_render_device_noalloc.restype = C.c_int32
        L.rtmi_render_device_noalloc.argtypes = L.rtmi_render_device.argtypes
        L.rtmi_noalloc_max_samples.restype = C.c_int64
        L.rtmi_direction_error_bound.restype = C.c_int32
        L.rtmi_direction_error_bound.argtypes = [C.POINTER(Camera), C.c_int32, C.c_int32, C.POINTER(C.c_double)]
        L.rtmi_scene_wait_event.restype = C.c_int32
        L.rtmi_scene_wait_event.argtypes = [C.c_void_p, C.c_void_p]
        L.rtmi_obj_load.restype = C.c_int32
        L.rtmi_obj_load.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
        L.rtmi_obj_free.restype = None
        L.rtmi_obj_free.argtypes = [C.c_void_p]
        for name in ("rtmi_obj_vertex_count", "rtmi_obj_triangle_count"):
            getattr(L, name).restype = C.c_int64
            getattr(L, name).argtypes = [C.c_void_p]
        L.rtmi_obj_material_count.restype = C.c_int32
        L.rtmi_obj_material_count.argtypes = [C.c_void_p]
        L.rtmi_obj_triangles.restype = C.c_void_p
        L.rtmi_obj_triangles.argtypes = [C.c_void_p]
        L.rtmi_obj_materials.restype = C.c_void_p
        L.rtmi_obj_materials.argtypes = [C.c_void_p]
        L.rtmi_obj_material_name.restype = C.c_char_p
        L.rtmi_obj_material_name.argtypes = [C.c_void_p, C.c_int32]
        for name in ("rtmi_write_ppm", "rtmi_write_png"):
            getattr(L, name).restype = C.c_int32
            getattr(L, name).argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32]
        L.rtmi_scene_from_obj.restype = C.c_int32
        L.rtmi_scene_from_obj.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        L.rtmi_tree_slot_order.restype = C.c_int32
        L.rtmi_tree_slot_order.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        if L.rtmi_abi_version() != _abi.RTMI_ABI_VERSION:
            raise ImportError("librtmi.so ABI version does not match raytracer_amd/_abi.py")
        _lib = L
    return _lib


def _check(rc, where):
    if rc != 0:
        L = lib()
        detail = L.rtmi_status_string(rc).decode()
        last = L.rtmi_last_error().decode()
        raise RtmiError(rc, where, detail + (": " + last if last else ""))


def samples_per_pixel(subdivision: float) -> int:
    n = lib().rtmi_samples_per_pixel(float(subdivision))
    _check(min(n, 0), "rtmi_samples_per_pixel")
    return n


def bounding_sphere(triangles: np.ndarray):
    """Sphere(boundingObjects:) (Renderables.swift:215-246) -> (center tuple, radius)."""
    c, r = (C.c_double * 3)(), C.c_double()
    tri = np.ascontiguousarray(triangles)
    _check(lib().rtmi_bounding_sphere(tri.ctypes.data, len(tri), c, C.byref(r)), "rtmi_bounding_sphere")
    return tuple(c), r.value


def direction_error_bound(camera: Camera, width: int, height: int) -> float:
    """The bound the pixels-only path assumes on |d - d*| for its FP32 sample directions (rtmi_direction_error_bound)."""
    b = C.c_double()
    _check(lib().rtmi_direction_error_bound(C.byref(camera), int(width), int(height), C.byref(b)), "rtmi_direction_error_bound")
    return b.value


def load_obj(obj_path, mtl_path=None):
    """readObjFile + readMtlFile + createTriangleList with the reference's parsing rules.
    Returns (triangles TRI_DTYPE array, materials MAT_DTYPE array, material names, vertex count)."""
    h = C.c_void_p()
    rc = lib().rtmi_obj_load(os.fsencode(obj_path), os.fsencode(mtl_path) if mtl_path else None, C.byref(h))
    _check(rc, "rtmi_obj_load")
    try:
        L = lib()
        nt, nm = L.rtmi_obj_triangle_count(h), L.rtmi_obj_material_count(h)
        tris = np.zeros(nt, scenes.TRI_DTYPE)
        if nt:
            C.memmove(tris.ctypes.data, L.rtmi_obj_triangles(h), nt * scenes.TRI_DTYPE.itemsize)
        mats = np.zeros(nm, scenes.MAT_DTYPE)
        if nm:
            C.memmove(mats.ctypes.data, L.rtmi_obj_materials(h), nm * scenes.MAT_DTYPE.itemsize)
        names = [L.rtmi_obj_material_name(h, i).decode() for i in range(nm)]
        return tris, mats, names, L.rtmi_obj_vertex_count(h)
    finally:
        lib().rtmi_obj_free(h)


def tree_slot_order(triangles: np.ndarray) -> np.ndarray:
    """slot -> list position (-1 = padding) of the cone tree's spatial order (rtmi_tree_slot_order)."""
    tri = np.ascontiguousarray(triangles)
    n = C.c_int64()
    _check(lib().rtmi_tree_slot_order(tri.ctypes.data, len(tri), None, 0, C.byref(n)), "rtmi_tree_slot_order")
    out = np.empty(n.value, np.int32)
    _check(lib().rtmi_tree_slot_order(tri.ctypes.data, len(tri), out.ctypes.data, len(out), C.byref(n)), "rtmi_tree_slot_order")
    return out


def write_image(path, rgba: np.ndarray):
    """Write an [H,W,4] uint8 frame as .ppm or .png (rtmi_write_ppm / rtmi_write_png)."""
    rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
    assert rgba.ndim == 3 and rgba.shape[2] == 4
    fn = lib().rtmi_write_ppm if str(path).lower().endswith(".ppm") else lib().rtmi_write_png
    _check(fn(os.fsencode(path), rgba.ctypes.data, rgba.shape[1], rgba.shape[0]), "rtmi_write_image")


class Scene:
    """A scene resident on one GPU (rtmi_scene_create / rtmi_scene_destroy)."""

    def __init__(self, data: "scenes.SceneData", device: int = 0):
        self._h = C.c_void_p()
        self.data = data
        self.device = device
        desc = data.desc()
        _check(lib().rtmi_scene_create(C.byref(desc), device, C.byref(self._h)), "rtmi_scene_create")

    @classmethod
    def from_obj(cls, obj_path, mtl_path=None, lights=None, device: int = 0):
        """loadObjFile in one call (rtmi_scene_from_obj): the mesh wrapped in Sphere(boundingObjects:)."""
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        self.data = None
        self.device = device
        lights = np.ascontiguousarray(lights if lights is not None else np.zeros(0, scenes.LIGHT_DTYPE))
        rc = lib().rtmi_scene_from_obj(os.fsencode(obj_path), os.fsencode(mtl_path) if mtl_path else None,
                                       lights.ctypes.data if len(lights) else None, len(lights), device, C.byref(self._h))
        _check(rc, "rtmi_scene_from_obj")
        return self

    def close(self):
        if self._h:
            lib().rtmi_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, camera: Camera, params: Params, want_hit_ids=True, rgba=None):
        """raytraceWorld into host arrays.  Returns (rgba [H,W,4] u8, hit_id [H,W,spp] i32|None,
        counters dict)."""
        W, H = params.width, params.height
        spp = samples_per_pixel(params.pixel_subdivision)
        if rgba is None:
            rgba = np.zeros((H, W, 4), np.uint8)
        assert rgba.shape == (H, W, 4) and rgba.dtype == np.uint8 and rgba.flags["C_CONTIGUOUS"]
        hit = np.full((H, W, spp), -1, np.int32) if want_hit_ids else None
        cnt = Counters()
        rc = lib().rtmi_render(self._h, C.byref(camera), C.byref(params), rgba.ctypes.data,
                               hit.ctypes.data if want_hit_ids else None, C.byref(cnt))
        _check(rc, "rtmi_render")
        return rgba, hit, cnt.as_dict()

    def render_device(self, camera: Camera, params: Params, rgba_ptr: int, hit_ptr: int = 0):
        """raytraceWorld into device buffers (raw pointers).  Returns the counters dict."""
        cnt = Counters()
        rc = lib().rtmi_render_device(self._h, C.byref(camera), C.byref(params), C.c_void_p(rgba_ptr),
                                      C.c_void_p(hit_ptr) if hit_ptr else None, C.byref(cnt))
        _check(rc, "rtmi_render_device")
        return cnt.as_dict()

    def render_device_noalloc(self, camera: Camera, params: Params, rgba_ptr: int, hit_ptr: int = 0):
        """render_device without any device allocation, for small frames (rtmi_render_device_noalloc)."""
        cnt = Counters()
        rc = lib().rtmi_render_device_noalloc(self._h, C.byref(camera), C.byref(params), C.c_void_p(rgba_ptr),
                                              C.c_void_p(hit_ptr) if hit_ptr else None, C.byref(cnt))
        _check(rc, "rtmi_render_device_noalloc")
        return cnt.as_dict()

    def render_begin(self, camera: Camera, params: Params, rgba_ptr: int, hit_ptr: int = 0):
        """Enqueue a frame into device buffers and return at once (rtmi_render_begin)."""
        rc = lib().rtmi_render_begin(self._h, C.byref(camera), C.byref(params), C.c_void_p(rgba_ptr),
                                     C.c_void_p(hit_ptr) if hit_ptr else None)
        _check(rc, "rtmi_render_begin")

    def wait_event(self, hip_event):
        """Order the scene's stream after `hip_event` (a raw hipEvent_t, e.g. torch.cuda.Event().cuda_event
        once recorded): rtmi_scene_wait_event."""
        _check(lib().rtmi_scene_wait_event(self._h, C.c_void_p(int(hip_event))), "rtmi_scene_wait_event")

    def render_end(self):
        """Wait for the frame in flight; returns its counters dict (rtmi_render_end)."""
        cnt = Counters()
        _check(lib().rtmi_render_end(self._h, C.byref(cnt)), "rtmi_render_end")
        return cnt.as_dict()
