"""How much does keeping two frames in flight (two scenes, each with its own stream and buffers) buy for the
per-rank share of an 8-GPU run?  python3 tools/pipeline_probe.py WORLD"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt  # noqa: E402
from raytracer_amd import scenes, sharding  # noqa: E402

world = int(sys.argv[1])
data = scenes.mesh_scene(4, "B", 3, bounding_sphere=rt.bounding_sphere)
cam = scenes.make_camera(**scenes.MESH_CAMERA)
params = scenes.make_params(3840, 2160, 0.25)
p = sharding.tile_params(params, 0, world) if world > 1 else params
frames = [torch.zeros((2160, 3840, 4), dtype=torch.uint8, device="cuda:0") for _ in range(2)]
N = 200 if world > 1 else 40
for depth in (1, 2):
    scs = [rt.Scene(data) for _ in range(depth)]
    for s in scs:
        s.render_device(cam, p, frames[0].data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pending = []
    for i in range(N):
        s = scs[i % depth]
        if len(pending) == depth:
            pending.pop(0).render_end()
        s.render_begin(cam, p, frames[i % 2].data_ptr())
        pending.append(s)
    for s in pending:
        s.render_end()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N * 1e3
    print(f"world {world} frames in flight {depth}: {dt:.3f} ms per frame", flush=True)
    for s in scs:
        s.close()
