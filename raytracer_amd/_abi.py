"""ctypes mirror of include/rtmi.h (struct layouts only, no behaviour).

Field order and padding follow the header exactly; tests/test_abi.py checks the sizes
against the values the C compiler reports.
"""
import ctypes as C

RTMI_ABI_VERSION = 4

RTMI_OK = 0
RTMI_ERR_INVALID_ARGUMENT = -1
RTMI_ERR_UNSUPPORTED = -2
RTMI_ERR_NO_DEVICE = -3
RTMI_ERR_HIP = -4
RTMI_ERR_OUT_OF_MEMORY = -5

RTMI_SPHERE = 0
RTMI_TRIANGLES = 1
RTMI_BOUNDING_SPHERE = 2
RTMI_LIGHT = 3

RTMI_HIT_LIST_ORDER = 0
RTMI_HIT_NEAREST = 1

D3 = C.c_double * 3


class Camera(C.Structure):
    _fields_ = [("pos", D3), ("dir", D3), ("focal_length", C.c_double)]


class Light(C.Structure):
    _fields_ = [("loc", D3), ("specular", D3), ("diffuse", D3)]


class Material(C.Structure):
    _fields_ = [("Ns", C.c_double), ("d", C.c_double), ("illum", C.c_int32), ("pad_", C.c_int32),
                ("Kd", D3), ("Ka", D3), ("Ks", D3)]


class Triangle(C.Structure):
    _fields_ = [("v", D3 * 3), ("material", C.c_int32), ("pad_", C.c_int32)]


class Object(C.Structure):
    _fields_ = [("kind", C.c_int32), ("material", C.c_int32), ("center", D3), ("radius", C.c_double),
                ("first", C.c_int64), ("count", C.c_int64)]


class SceneDesc(C.Structure):
    _fields_ = [("lights", C.POINTER(Light)), ("n_lights", C.c_int32), ("n_materials", C.c_int32),
                ("materials", C.POINTER(Material)), ("objects", C.POINTER(Object)),
                ("triangles", C.POINTER(Triangle)), ("n_objects", C.c_int64),
                ("n_triangles", C.c_int64)]


class Params(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("pixel_subdivision", C.c_double),
                ("ambient", D3), ("hit_policy", C.c_int32),
                ("x0", C.c_int32), ("y0", C.c_int32), ("x1", C.c_int32), ("y1", C.c_int32),
                ("tile_size", C.c_int32), ("tile_rank", C.c_int32), ("tile_world", C.c_int32),
                ("search", C.c_int32), ("flags", C.c_int32)]


class Counters(C.Structure):
    _fields_ = [("samples", C.c_int64), ("rays_reflection", C.c_int64), ("rays_shadow", C.c_int64),
                ("tri_tests", C.c_int64), ("sphere_tests", C.c_int64),
                ("tri_exact_tests", C.c_int64), ("primary_hits", C.c_int64),
                ("shadow_rays_elided", C.c_int64), ("node_tests", C.c_int64), ("tri_candidates", C.c_int64), ("ms_total", C.c_double), ("ms_tri_search", C.c_double),
                ("tri_search_launches", C.c_int64), ("tri_tests_primary_kernel", C.c_int64),
                ("ms_primary_kernel", C.c_double), ("ms_shade", C.c_double), ("primary_rays_traced", C.c_int64),
                ("hits_decided_fp32", C.c_int64), ("pixels_fixed_fp64", C.c_int64),
                ("tri_edge_tests", C.c_int64), ("work_buffer_bytes", C.c_int64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if n != "reserved_"}


RTMI_FLAG_NO_RAY_STATISTICS = 1
RTMI_FLAG_SHADOW_MAX_DISTANCE = 2
RTMI_FLAG_SHADOWS_ENTER_BOUNDED = 4
RTMI_FLAG_REFLECTIONS_ENTER_BOUNDED = 8

RTMI_SEARCH_CONE_TREE = 0
RTMI_SEARCH_LIST_SCAN = 1

# every symbol include/rtmi.h declares
EXPORTS = [
    "rtmi_abi_version", "rtmi_status_string", "rtmi_last_error", "rtmi_device_count",
    "rtmi_bounding_sphere", "rtmi_scene_create", "rtmi_scene_destroy", "rtmi_render",
    "rtmi_render_device", "rtmi_render_begin", "rtmi_render_end", "rtmi_samples_per_pixel",
    "rtmi_obj_load", "rtmi_obj_free", "rtmi_obj_vertex_count", "rtmi_obj_triangle_count", "rtmi_obj_triangles",
    "rtmi_obj_material_count", "rtmi_obj_materials", "rtmi_obj_material_name",
    "rtmi_write_ppm", "rtmi_write_png", "rtmi_tree_slot_order", "rtmi_scene_from_obj",
    "rtmi_scene_wait_event", "rtmi_render_device_noalloc", "rtmi_noalloc_max_samples", "rtmi_direction_error_bound",
]
