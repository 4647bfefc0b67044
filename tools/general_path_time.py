"""Times the C3 frame on the general path (full ray statistics) and form T, still camera."""
import sys
import time

import torch

sys.path.insert(0, ".")
import raytracer_amd as rt  # noqa: E402
from raytracer_amd import scenes  # noqa: E402

W, H = 3840, 2160
frame = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda:0")
cam = scenes.make_camera(**scenes.MESH_CAMERA)
for form in ("B", "T"):
    sc = rt.Scene(scenes.mesh_scene(4, form, 3, bounding_sphere=rt.bounding_sphere), device=0)
    params = scenes.make_params(W, H, 0.25)
    sc.render_device(cam, params, frame.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        c = sc.render_device(cam, params, frame.data_ptr())
    ms = (time.perf_counter() - t0) / 4 * 1e3
    print(f"form {form}: {ms:.3f} ms/frame  walk {c['ms_primary_kernel']:.3f} search {c['ms_tri_search']:.3f} shade {c['ms_shade']:.3f}", flush=True)
    sc.close()
