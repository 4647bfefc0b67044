"""Counters and device times of one frame for both search strategies (debugging aid)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt  # noqa: E402
from raytracer_amd import _abi, scenes  # noqa: E402

W, H, sub, level = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
data = scenes.mesh_scene(level, "B", 3, bounding_sphere=rt.bounding_sphere)
sc = rt.Scene(data)
cam = scenes.make_camera(**scenes.MESH_CAMERA)
for search in (_abi.RTMI_SEARCH_CONE_TREE, _abi.RTMI_SEARCH_LIST_SCAN):
    p = scenes.make_params(W, H, sub, search=search)
    sc.render(cam, p, want_hit_ids=False)
    _, _, c = sc.render(cam, p, want_hit_ids=False)
    print("search", search, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in c.items()}, flush=True)
