"""Run by tests/test_gpu_parity.py::test_frames_rendered_in_several_batches in a process of its own, with
RTMI_MAX_BATCH_SAMPLES set: renders frames that need several batches and compares them with the oracle."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
import raytracer_amd as rt  # noqa: E402
from raytracer_amd import _abi, scenes  # noqa: E402

assert int(os.environ["RTMI_MAX_BATCH_SAMPLES"]) <= 4096
cam = scenes.make_camera(**scenes.MESH_CAMERA)
for form in ("B", "T"):
    data = scenes.mesh_scene(3, form, 3, bounding_sphere=rt.bounding_sphere if form == "B" else None)
    sc = rt.Scene(data)
    for size, sub in (((150, 130), 0.5), ((97, 61), 0.25)):
        params = scenes.make_params(size[0], size[1], sub)
        o_rgba, o_hit, o_cnt = oracle.render(data, cam, params, threads=8)
        for search in (_abi.RTMI_SEARCH_CONE_TREE, _abi.RTMI_SEARCH_LIST_SCAN):
            params.search = search
            g_rgba, g_hit, g_cnt = sc.render(cam, params)
            assert g_cnt["tri_search_launches"] >= 4, g_cnt["tri_search_launches"]  # several batches
            assert (g_hit == o_hit).all() and (g_rgba == o_rgba).all(), (form, size, search)
            for k in ("samples", "rays_reflection", "rays_shadow", "primary_hits"):
                assert g_cnt[k] == o_cnt[k], (form, size, search, k)
    sc.close()
print("multibatch ok")
