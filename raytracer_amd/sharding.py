"""Tile sharding of one frame over the GPUs of a node (one process per GPU).

The render path has no exchange step: samples are independent.  The only communication is
(1) one byte broadcast of the flattened scene from rank 0 before the first frame and (2) one
gather of each rank's OWN pixels to rank 0 after each frame.  Each rank renders the 32x32
tiles whose id % world == rank (cyclic, so a mesh that covers part of the frame is spread
evenly).  A rank sends only the pixels it owns (1/world of the frame, packed), so rank 0
receives (world-1)/world of a frame in total, each peer over its own xGMI link; with the
`nccl` backend both calls are RCCL.
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


class TileComposer:
    """Collects the per-rank tiles of a frame on rank 0.

    Every rank owns `frame`, an [H, W, 4] uint8 tensor that its renderer writes its tiles into.
    `compose()` packs the rank's own pixels (as one int32 per RGBA pixel), gathers the packs on
    rank 0 and scatters them into rank 0's frame.  Index tensors are built once.
    """

    def __init__(self, width, height, rank, world, device, tile_size=TILE, group=None):
        self.rank, self.world, self.group = rank, world, group
        self.frame = torch.zeros((height, width, 4), dtype=torch.uint8, device=device)
        self._flat = self.frame.view(torch.int32).view(-1)  # one element per pixel
        idx = [np.flatnonzero(owned_pixel_mask(width, height, r, world, tile_size).ravel()) for r in range(world)]
        self.counts = [len(i) for i in idx]
        self.maxn = max(self.counts) if idx else 0
        self._own_idx = torch.from_numpy(idx[rank]).to(device)
        self._send = torch.zeros(self.maxn, dtype=torch.int32, device=device)
        if rank == 0:
            self._recv = torch.zeros((world, self.maxn), dtype=torch.int32, device=device)
            # peers' pixels only (rank 0's own tiles are already in its frame)
            self._peer_idx = [torch.from_numpy(idx[r]).to(device) for r in range(world)]

    def compose(self):
        """Returns rank 0's frame with every rank's tiles in place (None on other ranks)."""
        if self.world == 1:
            return self.frame
        n = self.counts[self.rank]
        torch.index_select(self._flat, 0, self._own_idx, out=self._send[:n])
        if self.rank == 0:
            dist.gather(self._send, list(self._recv.unbind(0)), dst=0, group=self.group)
            for r in range(1, self.world):
                self._flat.index_copy_(0, self._peer_idx[r], self._recv[r, : self.counts[r]])
            return self.frame
        dist.gather(self._send, None, dst=0, group=self.group)
        return None
