"""Multi-rank tile sharding on CPU (gloo, world_size 2): the composed frame must equal the
single-rank frame byte for byte.  The per-tile renderer here is the oracle (tests may use it);
on the GPU box the same layer drives librtmi.so (tests/test_gpu_parity.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from raytracer_amd import scenes, sharding
        data = scenes.mesh_scene(2, "B", 2, bounding_sphere=oracle.bounding_sphere) if rank == 0 else None
        data = sharding.broadcast_scene(data, 0)
        cam = scenes.make_camera(**scenes.MESH_CAMERA)
        params = scenes.make_params(100, 80, 0.5)
        p = sharding.tile_params(params, rank, world, tile_size=16)
        rgba, hit, cnt = oracle.render(data, cam, p)
        mask = sharding.owned_pixel_mask(100, 80, rank, world, 16)
        assert (rgba[~mask] == 0).all(), "a rank wrote outside its tiles"
        assert (rgba[mask][:, 3] == 255).all()
        comp = sharding.TileComposer(100, 80, rank, world, "cpu", tile_size=16)
        comp.frame.copy_(torch.from_numpy(rgba))
        frame = comp.compose()
        # a second frame through the same composer (stale pixels must be overwritten)
        cam2 = scenes.make_camera((0.4, 0.1, 4.0), (-0.1, 0.0, -1.0), 10.0)
        rgba2, _, _ = oracle.render(data, cam2, p)
        first = frame.numpy().copy() if rank == 0 else None
        comp.frame.copy_(torch.from_numpy(rgba2))
        frame2 = comp.compose()
        samples = torch.tensor([cnt["samples"]])
        dist.all_reduce(samples)
        if rank == 0:
            full, _, full_cnt = oracle.render(data, cam, params)
            full2, _, _ = oracle.render(data, cam2, params)
            q.put((bool((first == full).all() and (frame2.numpy() == full2).all()),
                   int(samples.item()) == full_cnt["samples"], len(data.triangles)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_tile_sharded_frame_equals_single_rank_frame(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    same, samples_ok, ntri = q.get(timeout=5)
    assert same and samples_ok and ntri == 320


def test_tile_ownership_partitions_the_rendered_area():
    from raytracer_amd import sharding
    W, H = 131, 77
    for world in (1, 2, 4, 8):
        total = np.zeros((H, W), int)
        for r in range(world):
            total += sharding.owned_pixel_mask(W, H, r, world)
        assert (total[:H - 1, :W - 1] == 1).all()
        assert (total[H - 1, :] == 0).all() and (total[:, W - 1] == 0).all()


def test_window_renders_equal_crops(oracle_mod):
    """Pixel windows use full-frame scaling (unlike the reference's own sub-range iterator)."""
    from raytracer_amd import scenes
    data = scenes.two_sphere_scene()
    cam = scenes.make_camera(**scenes.VC_CAMERA)
    full, fhit, _ = oracle_mod.render(data, cam, scenes.make_params(120, 120, 0.5))
    part, phit, cnt = oracle_mod.render(data, cam, scenes.make_params(120, 120, 0.5, window=(40, 50, 90, 70)))
    assert cnt["samples"] == 50 * 20 * 4
    assert (part[50:70, 40:90] == full[50:70, 40:90]).all()
    assert (phit[50:70, 40:90] == fhit[50:70, 40:90]).all()
    part[50:70, 40:90] = 0
    assert (part == 0).all()


def _worker_abi(rank, world, port, q, lib_path, tile_size):
    """The frame loop of bench.py (two scene objects per rank, frames alternating between two buffers, the gather of
    frame i enqueued while frame i+1 renders) on CPU ranks: the renderer is the library's own C ABI, from the
    CPU-emulated build of the kernels (tests/emu), loaded in place of librtmi.so IN THIS TEST PROCESS ONLY."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        import raytracer_amd as rt
        from raytracer_amd import scenes, sharding
        rt.LIB_PATH = lib_path
        W, H = 70, 52
        data = scenes.mesh_scene(2, "B", 2, bounding_sphere=oracle.bounding_sphere) if rank == 0 else None
        data = sharding.broadcast_scene(data, 0)
        params = scenes.make_params(W, H, 0.5)
        mine = sharding.tile_params(params, rank, world, tile_size=tile_size)
        comp = sharding.TileComposer(W, H, rank, world, "cpu", tile_size=tile_size)
        in_flight_scenes = [rt.Scene(data), rt.Scene(data)]
        cams = [scenes.make_camera(**scenes.orbit_camera(f, 64)) for f in (0, 0, 3, 5, 5)]
        in_flight, ok, composed = [], True, []

        def retire():
            sc, slot = in_flight.pop(0)
            sc.render_end()
            frame = comp.compose(slot)
            if rank == 0:
                composed.append(frame.numpy().copy())

        for i, cam in enumerate(cams):
            slot = i & 1
            if len(in_flight) == 2:
                retire()
            comp.wait_slot(slot)
            sc = in_flight_scenes[i % 2]
            sc.render_begin(cam, mine, comp.frames[slot].data_ptr())
            in_flight.append((sc, slot))
        while in_flight:
            retire()
        for sc in in_flight_scenes:
            sc.close()
        if rank == 0:
            for cam, frame in zip(cams, composed):
                full, _, _ = oracle.render(data, cam, params)
                ok = ok and bool((frame == full).all())
            q.put((ok, len(composed)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,tile_size", [(2, 16), (3, 12)])
def test_tile_sharded_frames_through_the_c_abi(world, tile_size):
    """world ranks x the real host code and kernels (CPU-emulated build) x the real composer: every composed frame
    equals the oracle's full frame.  Tile edge 16 is a multiple of the 8x8 sample blocks (block lists hold exactly
    the owned pixels); 12 is not (per-pixel ownership test in the kernels)."""
    import subprocess
    emu_dir = os.path.join(ROOT, "tests", "emu")
    subprocess.check_call(["make", "-s", "-C", emu_dir, "SAN=", "BUILD=build_ns", "build_ns/librtmi_emu.so"])
    lib_path = os.path.join(emu_dir, "build_ns", "librtmi_emu.so")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29100 + (os.getpid() % 500) + 7 * world
    procs = [ctx.Process(target=_worker_abi, args=(r, world, port, q, lib_path, tile_size)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    ok, n = q.get(timeout=5)
    assert ok and n == 5
