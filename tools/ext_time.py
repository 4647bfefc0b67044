"""Times the C3 frame (form B, triangles with a material so that shadows show) under the secondary-ray extension flags."""
import hashlib
import sys
import time

import torch

sys.path.insert(0, ".")
import raytracer_amd as rt  # noqa: E402
from raytracer_amd import scenes  # noqa: E402

level = int(sys.argv[1]) if len(sys.argv) > 1 else 4
w, h = (3840, 2160)
data = scenes.mesh_scene(level, "B", 3, bounding_sphere=rt.bounding_sphere)
data.triangles["material"] = 0
sc = rt.Scene(data, device=0)
frame = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda:0")
cam = scenes.make_camera(**scenes.MESH_CAMERA)
for flags, policy in ((0, 0), (2, 0), (4, 0), (6, 0), (8, 0), (14, 0), (0, 1), (14, 1)):
    params = scenes.make_params(w, h, 0.25, flags=flags, hit_policy=policy)
    sc.render_device(cam, params, frame.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        c = sc.render_device(cam, params, frame.data_ptr())
    ms = (time.perf_counter() - t0) / 3 * 1e3
    print(f"flags {flags}{' nearest-hit policy' if policy else ''}: {ms:.3f} ms/frame  ms_shade {c['ms_shade']:.3f}  refl {c['rays_reflection']} shadow {c['rays_shadow']} "
          f"tests {c['tri_tests']}  sha {hashlib.sha256(frame.cpu().numpy().tobytes()).hexdigest()[:16]}", flush=True)
sc.close()
