"""What ONE rank of an N-rank tile split has to do per C3 frame, timed on one GPU without any collective: the rank's
share of the 32x32 tiles (sharding.tile_params), two frames in flight as bench.py keeps them.  frame time x N against
the single-rank frame time shows the fixed per-frame cost that the split cannot divide (launches, the per-frame cone
tree, host work); the gather of the finished pixels is not included.

    python tools/rank_share_time.py [frames [N [depth]]]       (N: only the split over N ranks, 0 = all; depth: frames in flight, default 2)
"""
import sys
import time

import torch

sys.path.insert(0, ".")
import raytracer_amd as rt  # noqa: E402
from raytracer_amd import _abi, scenes, sharding  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
data = scenes.mesh_scene(4, "B", 3, bounding_sphere=rt.bounding_sphere)
W, H = 3840, 2160
params = scenes.make_params(W, H, 0.25, flags=_abi.RTMI_FLAG_NO_RAY_STATISTICS)
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 2
pair = [rt.Scene(data, device=0) for _ in range(depth)]
targets = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda:0") for _ in range(depth)]
base = None
for world in ((int(sys.argv[2]),) if len(sys.argv) > 2 and int(sys.argv[2]) else (1, 2, 4, 8)):
    for rank in sorted({0, world - 1}):
        p = sharding.tile_params(params, rank, world) if world > 1 else params
        in_flight = []
        dev_ms = 0.0

        def run(n, record):
            global dev_ms
            for i in range(n):
                if len(in_flight) == depth:
                    c = in_flight.pop(0).render_end()
                    if record:
                        dev_ms += c["ms_total"]
                sc = pair[i % depth]
                sc.render_begin(scenes.make_camera(**scenes.orbit_camera(i % 64)), p, targets[i % depth].data_ptr())
                in_flight.append(sc)
            while in_flight:
                c = in_flight.pop(0).render_end()
                if record:
                    dev_ms += c["ms_total"]

        run(6, False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(frames, True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / frames * 1e3
        if base is None:
            base = ms
        print(f"rank {rank} of {world}: {ms:.3f} ms/frame wall, {dev_ms / frames:.3f} ms device per frame; "
              f"x{world} = {ms * world:.3f} ms (single rank {base:.3f}); upper bound on the split's speed-up {base / ms:.2f}", flush=True)
for sc in pair:
    sc.close()
