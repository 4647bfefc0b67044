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

    Every rank owns `frames`, two [H, W, 4] uint8 tensors that its renderer writes its tiles
    into alternately.  `compose(slot)` packs the rank's own pixels of frames[slot] (one int32
    per RGBA pixel), gathers the packs on rank 0 and scatters them into rank 0's frames[slot].
    All of it is enqueued asynchronously, so the caller can render the next frame into the other
    slot while this one travels; `wait_slot(slot)` must be called before a slot is rendered into
    again.  Index tensors are built once.
    """

    def __init__(self, width, height, rank, world, device, tile_size=TILE, group=None):
        self.rank, self.world, self.group = rank, world, group
        self.device = torch.device(device)
        self.frames = [torch.zeros((height, width, 4), dtype=torch.uint8, device=device) for _ in range(2)]
        self.frame = self.frames[0]
        self._flat = [f.view(torch.int32).view(-1) for f in self.frames]  # one element per pixel
        self._packed = [None, None]  # CUDA event per slot: own pixels have been read out
        self._done = [None, None]    # CUDA event per slot: compose() has finished with frames[slot]
        if world == 1:
            return
        idx = [np.flatnonzero(owned_pixel_mask(width, height, r, world, tile_size).ravel()) for r in range(world)]
        self.counts = [len(i) for i in idx]
        self.maxn = max(self.counts)
        self._own_idx = torch.from_numpy(idx[rank]).to(device)
        self._send = [torch.zeros(self.maxn, dtype=torch.int32, device=device) for _ in range(2)]
        if rank == 0:
            self._recv = [torch.zeros((world, self.maxn), dtype=torch.int32, device=device) for _ in range(2)]
            # one scatter for all peers: positions of the valid entries of recv[1:] and their pixel indices
            src = np.concatenate([r * self.maxn + np.arange(self.counts[r]) for r in range(1, world)])
            dst = np.concatenate([idx[r] for r in range(1, world)])
            self._peer_src = torch.from_numpy(src).to(device)
            self._peer_dst = torch.from_numpy(dst).to(device)

    def wait_slot(self, slot):
        """Block the host until the previous compose() of `slot` has read the rank's pixels out of it."""
        ev = self._packed[slot]
        if ev is not None:
            ev.synchronize()

    def slot_event(self, slot):
        """CUDA event behind the last operation the previous compose() of `slot` enqueued on frames[slot] (the
        read-out of the rank's own pixels; on rank 0 also the scatter of the peers' pixels), or None.  A renderer
        orders its stream behind it (rtmi_scene_wait_event) before it writes frames[slot] again; the host does
        not wait."""
        return self._done[slot]

    def compose(self, slot=0):
        """Enqueue the gather of frames[slot]; returns rank 0's frame tensor (None on other ranks).
        The result is complete once the device (and the process group) has drained."""
        if self.world == 1:
            return self.frames[slot]
        n = self.counts[self.rank]
        send = self._send[slot]
        torch.index_select(self._flat[slot], 0, self._own_idx, out=send[:n])
        if self.device.type == "cuda":
            ev = torch.cuda.Event()
            ev.record()
            self._packed[slot] = ev
            self._done[slot] = ev
        if self.rank == 0:
            recv = self._recv[slot]
            dist.gather(send, list(recv.unbind(0)), dst=0, group=self.group)
            self._flat[slot].index_copy_(0, self._peer_dst, recv.view(-1).index_select(0, self._peer_src))
            if self.device.type == "cuda":
                ev = torch.cuda.Event()
                ev.record()
                self._done[slot] = ev
            return self.frames[slot]
        dist.gather(send, None, dst=0, group=self.group)
        return None
