"""Scene containers and the deterministic synthetic scenes used by tests and bench.

A `SceneData` holds the same data the reference keeps in its `objects`, `lights` and
material values (ViewController.swift:57-64,155-167) as numpy arrays laid out like the C
structs of include/rtmi.h, and hands out an `rtmi_scene_desc` view of them.
"""
import ctypes as C
import math
from dataclasses import dataclass, field

import numpy as np

from . import _abi

TRI_DTYPE = np.dtype([("v", "<f8", (3, 3)), ("material", "<i4"), ("pad_", "<i4")])
OBJ_DTYPE = np.dtype([("kind", "<i4"), ("material", "<i4"), ("center", "<f8", (3,)),
                      ("radius", "<f8"), ("first", "<i8"), ("count", "<i8")])
LIGHT_DTYPE = np.dtype([("loc", "<f8", (3,)), ("specular", "<f8", (3,)), ("diffuse", "<f8", (3,))])
MAT_DTYPE = np.dtype([("Ns", "<f8"), ("d", "<f8"), ("illum", "<i4"), ("pad_", "<i4"),
                      ("Kd", "<f8", (3,)), ("Ka", "<f8", (3,)), ("Ks", "<f8", (3,))])

assert TRI_DTYPE.itemsize == C.sizeof(_abi.Triangle)
assert OBJ_DTYPE.itemsize == C.sizeof(_abi.Object)
assert LIGHT_DTYPE.itemsize == C.sizeof(_abi.Light)
assert MAT_DTYPE.itemsize == C.sizeof(_abi.Material)


@dataclass
class SceneData:
    lights: np.ndarray = field(default_factory=lambda: np.zeros(0, LIGHT_DTYPE))
    materials: np.ndarray = field(default_factory=lambda: np.zeros(0, MAT_DTYPE))
    objects: np.ndarray = field(default_factory=lambda: np.zeros(0, OBJ_DTYPE))
    triangles: np.ndarray = field(default_factory=lambda: np.zeros(0, TRI_DTYPE))
    name: str = ""

    def desc(self) -> _abi.SceneDesc:
        """rtmi_scene_desc pointing into this object's arrays (keep `self` alive)."""
        for a in (self.lights, self.materials, self.objects, self.triangles):
            assert a.flags["C_CONTIGUOUS"]
        d = _abi.SceneDesc()
        d.lights = self.lights.ctypes.data_as(C.POINTER(_abi.Light))
        d.n_lights = len(self.lights)
        d.materials = self.materials.ctypes.data_as(C.POINTER(_abi.Material))
        d.n_materials = len(self.materials)
        d.objects = self.objects.ctypes.data_as(C.POINTER(_abi.Object))
        d.n_objects = len(self.objects)
        d.triangles = self.triangles.ctypes.data_as(C.POINTER(_abi.Triangle))
        d.n_triangles = len(self.triangles)
        return d

    # -- persistence for golden fixtures (plain arrays, no pickling) --
    def to_arrays(self):
        return {"lights": self.lights.view(np.uint8), "materials": self.materials.view(np.uint8),
                "objects": self.objects.view(np.uint8), "triangles": self.triangles.view(np.uint8)}

    @staticmethod
    def from_arrays(arrs, name=""):
        return SceneData(
            lights=np.ascontiguousarray(arrs["lights"]).view(LIGHT_DTYPE),
            materials=np.ascontiguousarray(arrs["materials"]).view(MAT_DTYPE),
            objects=np.ascontiguousarray(arrs["objects"]).view(OBJ_DTYPE),
            triangles=np.ascontiguousarray(arrs["triangles"]).view(TRI_DTYPE), name=name)


def make_camera(pos, direction, focal_length) -> _abi.Camera:
    cam = _abi.Camera()
    cam.pos[:] = [float(x) for x in pos]
    cam.dir[:] = [float(x) for x in direction]
    cam.focal_length = float(focal_length)
    return cam


def make_params(width, height, subdivision=1.0, ambient=(0.2, 0.2, 0.2), window=None,
                tile_size=0, tile_rank=0, tile_world=1, search=_abi.RTMI_SEARCH_CONE_TREE, flags=0,
                hit_policy=_abi.RTMI_HIT_LIST_ORDER) -> _abi.Params:
    p = _abi.Params()
    p.width, p.height = int(width), int(height)
    p.pixel_subdivision = float(subdivision)
    p.ambient[:] = [float(x) for x in ambient]
    p.hit_policy = int(hit_policy)
    if window is not None:
        p.x0, p.y0, p.x1, p.y1 = [int(x) for x in window]
    p.tile_size, p.tile_rank, p.tile_world = int(tile_size), int(tile_rank), int(tile_world)
    p.search = int(search)
    p.flags = int(flags)
    return p


def material(Ns, d, illum, Kd, Ka, Ks) -> np.ndarray:
    m = np.zeros(1, MAT_DTYPE)
    m["Ns"], m["d"], m["illum"] = Ns, d, illum
    m["Kd"], m["Ka"], m["Ks"] = Kd, Ka, Ks
    return m


def light(loc, specular, diffuse) -> np.ndarray:
    l = np.zeros(1, LIGHT_DTYPE)
    l["loc"], l["specular"], l["diffuse"] = loc, specular, diffuse
    return l


def sphere_object(center, radius, material_index=-1, bounding=False, first=0, count=0) -> np.ndarray:
    o = np.zeros(1, OBJ_DTYPE)
    o["kind"] = _abi.RTMI_BOUNDING_SPHERE if bounding else _abi.RTMI_SPHERE
    o["material"] = material_index
    o["center"], o["radius"] = center, radius
    o["first"], o["count"] = first, count
    return o


def triangles_object(first, count) -> np.ndarray:
    o = np.zeros(1, OBJ_DTYPE)
    o["kind"], o["material"], o["first"], o["count"] = _abi.RTMI_TRIANGLES, -1, first, count
    return o


def light_object(light_index) -> np.ndarray:
    o = np.zeros(1, OBJ_DTYPE)
    o["kind"], o["material"], o["first"] = _abi.RTMI_LIGHT, -1, light_index
    return o


def triangles_from_vertices(verts, faces, material_index=0) -> np.ndarray:
    """createTriangleList (ViewController.swift:107-120) for 0-based `faces`."""
    t = np.zeros(len(faces), TRI_DTYPE)
    t["v"] = np.asarray(verts, np.float64)[np.asarray(faces)]
    t["material"] = material_index
    return t


# the material the reference builds in startRT (ViewController.swift:155)
VC_MATERIAL = dict(Ns=8.0, d=10.0, illum=5, Kd=(0.5, 0.5, 0.5), Ka=(0.7, 0.7, 0.7), Ks=(1.0, 1.0, 1.0))
# the default lights (ViewController.swift:59-64) and a third one for the 3-light configs
L1 = dict(loc=(50.0, 0.0, 50.0), specular=(0.8, 0.8, 0.8), diffuse=(0.5, 0.5, 0.5))
L2 = dict(loc=(-50.0, 0.0, -50.0), specular=(0.8, 0.8, 0.8), diffuse=(0.5, 0.5, 0.5))
L3 = dict(loc=(0.0, 50.0, 50.0), specular=(0.8, 0.8, 0.8), diffuse=(0.5, 0.5, 0.5))


def two_sphere_scene() -> SceneData:
    """The hard-coded scene of startRT (ViewController.swift:155-167)."""
    return SceneData(
        lights=np.concatenate([light(**L1), light(**L2)]),
        materials=material(**VC_MATERIAL),
        objects=np.concatenate([sphere_object((25.0, 0.0, 0.0), 25.0, 0),
                                sphere_object((-25.0, 0.0, 0.0), 25.0, 0)]),
        name="two_sphere")


def single_sphere_scene() -> SceneData:
    """Config 1: one sphere, one light."""
    return SceneData(lights=light(**L1), materials=material(**VC_MATERIAL),
                     objects=sphere_object((0.0, 0.0, 0.0), 25.0, 0), name="single_sphere")


VC_CAMERA = dict(pos=(0.0, 0.0, 40.0), direction=(0.0, 0.0, -1.0), focal_length=1.0)
MESH_CAMERA = dict(pos=(0.0, 0.0, 4.0), direction=(0.0, 0.0, -1.0), focal_length=10.0)


def icosphere(level: int):
    """Unit icosphere: (vertices [V,3] f64, faces [F,3] int64), F = 20 * 4**level.

    Deterministic: faces of level L+1 are, for each face (a,b,c) of level L in order,
    (a,ab,ca), (b,bc,ab), (c,ca,bc), (ab,bc,ca); new vertices are numbered by sorted
    (min,max) edge key.  Winding is counter-clockwise seen from outside.
    """
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t],
                  [0, -1, -t], [0, 1, -t], [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4],
                  [11, 10, 2], [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8],
                  [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], np.int64)
    for _ in range(level):
        nv = len(v)
        e = np.stack([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], axis=1).reshape(-1, 2)
        key = np.minimum(e[:, 0], e[:, 1]) * nv + np.maximum(e[:, 0], e[:, 1])
        uniq, inv = np.unique(key, return_inverse=True)
        a, b = uniq // nv, uniq % nv
        mid = v[a] + v[b]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        v = np.concatenate([v, mid])
        m = (inv + nv).reshape(-1, 3)  # per face: ab, bc, ca
        ab, bc, ca = m[:, 0], m[:, 1], m[:, 2]
        f = np.stack([np.stack([f[:, 0], ab, ca], 1), np.stack([f[:, 1], bc, ab], 1),
                      np.stack([f[:, 2], ca, bc], 1), np.stack([ab, bc, ca], 1)], axis=1).reshape(-1, 3)
    return v, f


def displaced_icosphere(level: int):
    """The stand-in mesh: icosphere with r = 1 + 0.25 sin(3 theta) cos(2 phi) (non-convex)."""
    v, f = icosphere(level)
    theta = np.arccos(np.clip(v[:, 2], -1.0, 1.0))
    phi = np.arctan2(v[:, 1], v[:, 0])
    r = 1.0 + 0.25 * np.sin(3.0 * theta) * np.cos(2.0 * phi)
    return v * r[:, None], f


def mesh_scene(level=4, form="B", n_lights=1, shuffle_seed=None, bounding_sphere=None) -> SceneData:
    """Displaced icosphere wrapped in Sphere(boundingObjects:) (form "B", what loadObjFile
    builds, ViewController.swift:101-104) or listed at top level (form "T").

    `bounding_sphere(triangles) -> (center, radius)` computes the wrapper for form B; tests
    pass the oracle's, the product passes librtmi's rtmi_bounding_sphere.
    """
    v, f = displaced_icosphere(level)
    if shuffle_seed is not None:
        f = f[np.random.default_rng(shuffle_seed).permutation(len(f))]
    tris = triangles_from_vertices(v, f, 0)
    lights = np.concatenate([light(**L) for L in (L1, L2, L3)[:n_lights]])
    if form == "B":
        if bounding_sphere is None:
            raise ValueError("form B needs a bounding_sphere function")
        center, radius = bounding_sphere(tris)
        objs = sphere_object(center, radius, -1, bounding=True, first=0, count=len(tris))
    elif form == "T":
        objs = triangles_object(0, len(tris))
    else:
        raise ValueError(form)
    return SceneData(lights=lights, materials=material(**VC_MATERIAL), objects=objs, triangles=tris,
                     name=f"icosphere{level}_{form}_{n_lights}L")


def orbit_camera(frame: int, n_frames: int = 64, radius: float = 4.0, focal_length: float = 10.0):
    """Config 5: camera on a circle about the y axis, looking at the origin."""
    a = 2.0 * math.pi * frame / n_frames
    pos = (radius * math.sin(a), 0.0, radius * math.cos(a))
    return dict(pos=pos, direction=(-math.sin(a), 0.0, -math.cos(a)), focal_length=focal_length)
