"""Tile sharding of one frame over the GPUs of a node (one process per GPU).

The render path has no exchange step: samples are independent.  The only communication is
(1) one byte broadcast of the flattened scene from rank 0 before the first frame and (2) one
byte reduction of the RGBA8 frame to rank 0 after each frame.  Each rank renders the 32x32
tiles whose id % world == rank (cyclic, so a mesh that covers part of the frame is spread
evenly) into a zero-initialised full frame; tiles are disjoint, so a SUM reduction of the
uint8 frames is exactly the composite.  With the `nccl` backend both calls are RCCL over xGMI.
"""
import copy

import numpy as np
import torch
import torch.distributed as dist

from . import scenes

TILE = 32


def tile_params(params, rank: int, world: int, tile_size: int = TILE):
    """Copy of `params` restricted to the tiles owned by `rank`."""
    p = copy.copy(params)
    p.tile_size, p.tile_rank, p.tile_world = tile_size, rank, world
    return p


def owned_pixel_mask(width, height, rank, world, tile_size=TILE):
    """[H,W] bool: pixels of the rendered (W-1)x(H-1) area that `rank` owns."""
    ys, xs = np.mgrid[0:height, 0:width]
    tiles_x = (width - 1 + tile_size - 1) // tile_size
    tile = (ys // tile_size) * tiles_x + (xs // tile_size)
    m = (tile % max(world, 1)) == rank
    m &= (xs < width - 1) & (ys < height - 1)
    return m


def broadcast_scene(data, src: int = 0, device="cpu", group=None):
    """Broadcast the four scene arrays as bytes from `src`; returns a SceneData on every rank."""
    rank = dist.get_rank(group)
    keys = ("lights", "materials", "objects", "triangles")
    sizes = torch.zeros(len(keys), dtype=torch.int64, device=device)
    if rank == src:
        arrs = data.to_arrays()
        sizes = torch.tensor([arrs[k].size for k in keys], dtype=torch.int64, device=device)
    dist.broadcast(sizes, src, group=group)
    out = {}
    for k, n in zip(keys, sizes.tolist()):
        if rank == src:
            t = torch.from_numpy(np.ascontiguousarray(arrs[k]).copy()).to(device)
        else:
            t = torch.empty(n, dtype=torch.uint8, device=device)
        if n:
            dist.broadcast(t, src, group=group)
        out[k] = t.cpu().numpy()
    return data if rank == src else scenes.SceneData.from_arrays(out)


def composite(frame: torch.Tensor, dst: int = 0, group=None):
    """Sum-reduce the per-rank uint8 frames (disjoint tiles, zeros elsewhere) onto `dst`."""
    dist.reduce(frame, dst, op=dist.ReduceOp.SUM, group=group)
    return frame
