"""Times the per-rank render of the C3 frame for world = 1, 2, 4, 8 on ONE GPU (each rank's tile set
in turn): shows how evenly the cyclic tiling splits the work and what the fixed per-call cost is."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt  # noqa: E402
from raytracer_amd import scenes, sharding  # noqa: E402

data = scenes.mesh_scene(4, "B", 3, bounding_sphere=rt.bounding_sphere)
sc = rt.Scene(data)
cam = scenes.make_camera(**scenes.MESH_CAMERA)
params = scenes.make_params(3840, 2160, 0.25)
frame = torch.zeros((2160, 3840, 4), dtype=torch.uint8, device="cuda:0")
torch.cuda.synchronize()
for world in (1, 2, 4, 8):
    per_rank = []
    for r in range(world):
        p = sharding.tile_params(params, r, world) if world > 1 else params
        sc.render_device(cam, p, frame.data_ptr())
        best_dev, best_wall = 1e9, 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            c = sc.render_device(cam, p, frame.data_ptr())
            best_wall = min(best_wall, (time.perf_counter() - t0) * 1e3)
            best_dev = min(best_dev, c["ms_total"])
        per_rank.append((best_dev, best_wall, c["ms_primary_kernel"]))
    devs = [x[0] for x in per_rank]
    walls = [x[1] for x in per_rank]
    print(f"world {world}: device ms per rank max {max(devs):.3f} min {min(devs):.3f} | wall max {max(walls):.3f} | "
          f"search {max(x[2] for x in per_rank):.3f} | ideal {per_rank and (7.3 / world):.3f}", flush=True)
