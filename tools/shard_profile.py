"""Renders ONE rank's tile set of the C3 frame repeatedly (for rocprofv3 --kernel-trace --stats):
python3 tools/shard_profile.py WORLD RANK ITERS"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt  # noqa: E402
from raytracer_amd import scenes, sharding  # noqa: E402

world, rank, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
data = scenes.mesh_scene(4, "B", 3, bounding_sphere=rt.bounding_sphere)
sc = rt.Scene(data)
cam = scenes.make_camera(**scenes.MESH_CAMERA)
params = scenes.make_params(3840, 2160, 0.25)
p = sharding.tile_params(params, rank, world) if world > 1 else params
frame = torch.zeros((2160, 3840, 4), dtype=torch.uint8, device="cuda:0")
torch.cuda.synchronize()
tot = 0.0
for _ in range(iters):
    tot += sc.render_device(cam, p, frame.data_ptr())["ms_total"]
print(f"world {world} rank {rank}: {tot / iters:.3f} ms per frame (device)")
