"""Counters and timing of the C3 frame on the path without a ray-generation launch (and with its fallbacks forced)."""
import sys
import time

import torch

sys.path.insert(0, ".")
import raytracer_amd as rt  # noqa: E402
from raytracer_amd import _abi, scenes  # noqa: E402

data = scenes.mesh_scene(4, "B", 3, bounding_sphere=rt.bounding_sphere)
sc = rt.Scene(data, device=0)
W, H = 3840, 2160
frame = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda:0")
for name, cam in (("still", scenes.make_camera(**scenes.MESH_CAMERA)), ("orbit 5", scenes.make_camera(**scenes.orbit_camera(5)))):
    params = scenes.make_params(W, H, 0.25, flags=_abi.RTMI_FLAG_NO_RAY_STATISTICS)
    sc.render_device(cam, params, frame.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        c = sc.render_device(cam, params, frame.data_ptr())
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"{name}: {ms:.3f} ms/frame  search {c['ms_tri_search']:.3f} shade {c['ms_shade']:.3f}  hits {c['primary_hits']} candidates {c['tri_candidates']} "
          f"decided in FP32 {c['hits_decided_fp32']} ({c['hits_decided_fp32'] / c['primary_hits']:.4f} of hits) in FP64 {c['tri_exact_tests']} "
          f"pixels fixed {c['pixels_fixed_fp64']} ({c['pixels_fixed_fp64'] / (c['primary_hits'] / 16):.4f} of hit pixels)", flush=True)
sc.close()
