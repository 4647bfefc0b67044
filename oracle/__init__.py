"""ctypes binding of the CPU oracle (oracle/rt_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (raytracer_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from raytracer_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "rt_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.oracle_render.restype = C.c_int32
        L.oracle_render.argtypes = [C.POINTER(_abi.SceneDesc), C.POINTER(_abi.Camera),
                                    C.POINTER(_abi.Params), C.c_void_p, C.c_void_p,
                                    C.POINTER(_abi.Counters), C.c_int32]
        L.oracle_bounding_sphere.restype = C.c_int32
        L.oracle_bounding_sphere.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_double),
                                             C.POINTER(C.c_double)]
        L.oracle_plane_vectors.restype = None
        L.oracle_plane_vectors.argtypes = [C.POINTER(C.c_double)] * 3
        L.oracle_frame.restype = None
        L.oracle_frame.argtypes = [C.POINTER(_abi.Camera), C.c_int32, C.c_int32] + \
            [C.POINTER(C.c_double)] * 5
        L.oracle_pixel_to_world.restype = None
        L.oracle_pixel_to_world.argtypes = [C.POINTER(C.c_double)] * 3 + [C.c_double] * 4 + \
            [C.POINTER(C.c_double)]
        L.oracle_pixel_samples.restype = C.c_int32
        L.oracle_pixel_samples.argtypes = [C.c_int32, C.c_int32, C.c_double,
                                           C.POINTER(C.c_double), C.c_int32]
        L.oracle_sphere_intersections.restype = C.c_int32
        L.oracle_sphere_intersections.argtypes = [C.POINTER(C.c_double), C.c_double] + \
            [C.POINTER(C.c_double)] * 5
        L.oracle_triangle_intersection.restype = C.c_int32
        L.oracle_triangle_intersection.argtypes = [C.POINTER(C.c_double)] * 6
        L.oracle_light_intersection.restype = C.c_int32
        L.oracle_light_intersection.argtypes = [C.POINTER(C.c_double)] * 5
        _lib = L
    return _lib


def _d(a, n=None):
    arr = (C.c_double * (n or len(a)))(*[float(x) for x in a])
    return arr


def plane_vectors(direction):
    u, v = (C.c_double * 3)(), (C.c_double * 3)()
    lib().oracle_plane_vectors(_d(direction), u, v)
    return list(u), list(v)


def frame(camera, width, height):
    ul, u, v = (C.c_double * 3)(), (C.c_double * 3)(), (C.c_double * 3)()
    ph, pv = C.c_double(), C.c_double()
    lib().oracle_frame(C.byref(camera), width, height, ul, u, v, C.byref(ph), C.byref(pv))
    return list(ul), list(u), list(v), ph.value, pv.value


def pixel_to_world(ul, u, v, ppw_h, ppw_v, i, j):
    out = (C.c_double * 3)()
    lib().oracle_pixel_to_world(_d(ul), _d(u), _d(v), ppw_h, ppw_v, float(i), float(j), out)
    return list(out)


def pixel_samples(x, y, subdivision):
    buf = (C.c_double * (2 * 4096))()
    n = lib().oracle_pixel_samples(x, y, subdivision, buf, 4096)
    return [(buf[2 * k], buf[2 * k + 1]) for k in range(n)]


def sphere_intersections(center, radius, o, d):
    t, P, n = (C.c_double * 2)(), (C.c_double * 6)(), (C.c_double * 6)()
    c = lib().oracle_sphere_intersections(_d(center), float(radius), _d(o), _d(d), t, P, n)
    return [(t[k], list(P[3 * k:3 * k + 3]), list(n[3 * k:3 * k + 3])) for k in range(c)]


def triangle_intersection(v, o, d):
    flat = _d([x for p in v for x in p])
    t, P, n = C.c_double(), (C.c_double * 3)(), (C.c_double * 3)()
    hit = lib().oracle_triangle_intersection(flat, _d(o), _d(d), C.byref(t), P, n)
    return (t.value, list(P), list(n)) if hit else None


def light_intersection(loc, o, d):
    t, P = C.c_double(), (C.c_double * 3)()
    hit = lib().oracle_light_intersection(_d(loc), _d(o), _d(d), C.byref(t), P)
    return (t.value, list(P)) if hit else None


def bounding_sphere(triangles: np.ndarray):
    """Sphere(boundingObjects:) for a TRI_DTYPE array -> (center tuple, radius)."""
    c, r = (C.c_double * 3)(), C.c_double()
    rc = lib().oracle_bounding_sphere(triangles.ctypes.data, len(triangles), c, C.byref(r))
    assert rc == 0
    return tuple(c), r.value


def render(scene, camera, params, want_hit_ids=True, threads=1):
    """Returns (rgba [H,W,4] uint8, hit_id [H,W,spp] int32 or None, counters dict)."""
    W, H = params.width, params.height
    spp = len(pixel_samples(0, 0, params.pixel_subdivision))
    rgba = np.zeros((H, W, 4), np.uint8)
    hit = np.full((H, W, spp), -1, np.int32) if want_hit_ids else None
    cnt = _abi.Counters()
    desc = scene.desc()
    rc = lib().oracle_render(C.byref(desc), C.byref(camera), C.byref(params), rgba.ctypes.data,
                             hit.ctypes.data if want_hit_ids else None, C.byref(cnt), threads)
    if rc != 0:
        raise RuntimeError(f"oracle_render failed: {rc}")
    return rgba, hit, cnt.as_dict()
