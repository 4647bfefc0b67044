"""Whole frames of the BASELINE configurations on the GPU against the oracle's digests.

tests/golden/fullframe_hashes.json holds, per frame, the SHA-256 of the oracle's RGBA8 bitmap and of its per-sample
hit-id array (plus a CRC-32 per band of 64 pixel rows and the ray counters), computed on the CPU by
tests/golden/make_fullframe_hashes.py.  Here the same frames are rendered through the C ABI and digested the same
way: every sample of the 4K/16-spp frame is compared, not a window of it.  Bar: identical digests (hit ids are
bit-exact; RGBA8 has a stated tolerance of 1 LSB elsewhere, but a digest can only be equal, and it is).
"""
import hashlib
import json
import os
import zlib

import numpy as np
import pytest
import torch  # before librtmi.so is loaded: one process, one HIP runtime

from raytracer_amd import _abi, scenes

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = json.load(open(os.path.join(HERE, "golden", "fullframe_hashes.json")))
COUNTERS = ("samples", "rays_reflection", "rays_shadow", "sphere_tests", "primary_hits")


def scene_for(rec, rt):
    data = scenes.mesh_scene(rec["mesh_level"], rec["form"], rec["lights"], bounding_sphere=rt.bounding_sphere)
    if "triangle_material" in rec:
        data.triangles["material"] = rec["triangle_material"]
    h = hashlib.sha256()
    for a in (data.lights, data.materials, data.objects, data.triangles):
        h.update(np.ascontiguousarray(a).tobytes())
    assert h.hexdigest() == rec["scene_sha256"], "the generated scene differs from the one the digests were made for"
    return data


def camera_for(rec):
    return scenes.make_camera(**(scenes.MESH_CAMERA if rec["camera"] == "mesh" else scenes.orbit_camera(int(rec["camera"]))))


def check_digest(name, rec, rgba_dev, hit_dev, cnt):
    rgba = rgba_dev.cpu().numpy()
    hit = hit_dev.cpu().numpy()
    band = rec["band_rows"]
    if hashlib.sha256(hit.tobytes()).hexdigest() != rec["hit_sha256"]:
        bad = [i for i, y in enumerate(range(0, hit.shape[0], band))
               if zlib.crc32(np.ascontiguousarray(hit[y:y + band]).tobytes()) != rec["hit_band_crc32"][i]]
        raise AssertionError(f"{name}: hit ids differ from the oracle's in {len(bad)} bands of {band} rows, first {bad[:5]}")
    if hashlib.sha256(rgba.tobytes()).hexdigest() != rec["rgba_sha256"]:
        bad = [i for i, y in enumerate(range(0, rgba.shape[0], band))
               if zlib.crc32(np.ascontiguousarray(rgba[y:y + band]).tobytes()) != rec["rgba_band_crc32"][i]]
        raise AssertionError(f"{name}: RGBA8 differs from the oracle's in {len(bad)} bands of {band} rows, first {bad[:5]}")
    for k in COUNTERS:
        assert cnt[k] == rec["counters"][k], (name, k)


def buffers(rec, rt):
    W, H = rec["width"], rec["height"]
    spp = rt.samples_per_pixel(rec["subdivision"])
    rgba = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda:0")
    hit = torch.full((H, W, spp), -1, dtype=torch.int32, device="cuda:0")
    return rgba, hit


@pytest.mark.parametrize("name", [n for n in ("C2", "C3", "C4", "T_640x360", "C2_extensions") if n in GOLDEN])
def test_whole_frame_equals_the_oracles(rt, name):
    rec = GOLDEN[name]
    sc = rt.Scene(scene_for(rec, rt))
    cam = camera_for(rec)
    for search in (_abi.RTMI_SEARCH_CONE_TREE, _abi.RTMI_SEARCH_LIST_SCAN):
        # (C2_extensions: the secondary-ray extension flags the digest was made with)
        p = scenes.make_params(rec["width"], rec["height"], rec["subdivision"], search=search, flags=rec.get("flags", 0))
        rgba, hit = buffers(rec, rt)
        torch.cuda.synchronize()
        cnt = sc.render_device(cam, p, rgba.data_ptr(), hit.data_ptr())
        check_digest(f"{name} search {search}", rec, rgba, hit, cnt)
        del rgba, hit
    sc.close()


@pytest.mark.parametrize("name", [n for n in ("C2", "C3", "C4", "C5_step05", "C5_step16", "C5_step48") if n in GOLDEN])
def test_whole_frame_pixels_only_equals_the_oracles(rt, name):
    """The same frames asked for with RTMI_FLAG_NO_RAY_STATISTICS -- what bench.py times.  These scenes (one bounding
    sphere without a material) then take the path without a ray-generation launch: FP32 directions, decisions by FP32
    margins or in FP64, FP32 lighting with an error bound and an FP64 second pass for the pixels it cannot settle.  Same
    bar: the oracle's digests of bitmap and hit ids.  Once more with each of the path's two fallbacks made the rule
    (RTMI_FAST_FORCE is read when the library first renders, so those run in child processes)."""
    rec = GOLDEN[name]
    sc = rt.Scene(scene_for(rec, rt))
    cam = camera_for(rec)
    p = scenes.make_params(rec["width"], rec["height"], rec["subdivision"], flags=_abi.RTMI_FLAG_NO_RAY_STATISTICS)
    for rep in range(2):  # the second frame reuses the camera's preparation
        rgba, hit = buffers(rec, rt)
        torch.cuda.synchronize()
        cnt = sc.render_device(cam, p, rgba.data_ptr(), hit.data_ptr())
        check_digest(f"{name} pixels only", dict(rec, counters={k: cnt[k] for k in COUNTERS}), rgba, hit, cnt)
        assert cnt["primary_hits"] == rec["counters"]["primary_hits"] and cnt["samples"] == rec["counters"]["samples"]
        assert cnt["rays_reflection"] == 0 and cnt["rays_shadow"] == 0
        # the path is the one that ran: nearly every hit decided by the FP32 margins, a small share of the pixels shaded twice
        assert cnt["hits_decided_fp32"] > 0.95 * cnt["primary_hits"], cnt
        assert 0 < cnt["pixels_fixed_fp64"] < 0.2 * cnt["primary_hits"] / rt.samples_per_pixel(rec["subdivision"]), cnt
        del rgba, hit
    sc.close()


@pytest.mark.parametrize("force", ["1", "2", "3", "walks start at the root"])
def test_pixels_only_fallbacks_made_the_rule(force):
    """C2 with every pair decided in FP64 (1), every pixel shaded again in FP64 (2), both (3), or with the tree walks
    started at the root instead of the records found per block: the same digests."""
    import subprocess
    import sys
    code = (
        "import json, hashlib, sys, torch\n"
        "sys.path.insert(0, %r)\n"
        "import raytracer_amd as rt\n"
        "from raytracer_amd import _abi, scenes\n"
        "rec = json.load(open(%r))['C2']\n"
        "data = scenes.mesh_scene(rec['mesh_level'], rec['form'], rec['lights'], bounding_sphere=rt.bounding_sphere)\n"
        "sc = rt.Scene(data)\n"
        "W, H = rec['width'], rec['height']\n"
        "rgba = torch.zeros((H, W, 4), dtype=torch.uint8, device='cuda:0')\n"
        "hit = torch.full((H, W, 4), -1, dtype=torch.int32, device='cuda:0')\n"
        "torch.cuda.synchronize()\n"
        "p = scenes.make_params(W, H, rec['subdivision'], flags=_abi.RTMI_FLAG_NO_RAY_STATISTICS)\n"
        "cnt = sc.render_device(scenes.make_camera(**scenes.MESH_CAMERA), p, rgba.data_ptr(), hit.data_ptr())\n"
        "print(json.dumps({'rgba': hashlib.sha256(rgba.cpu().numpy().tobytes()).hexdigest(), 'hit': hashlib.sha256(hit.cpu().numpy().tobytes()).hexdigest(), 'cnt': cnt}))\n"
    ) % (os.path.dirname(HERE), os.path.join(HERE, "golden", "fullframe_hashes.json"))
    env = dict(os.environ, RTMI_BLOCK_ENTRIES="0") if not force.isdigit() else dict(os.environ, RTMI_FAST_FORCE=force)
    force = force if force.isdigit() else "0"
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    rec = GOLDEN["C2"]
    assert out["rgba"] == rec["rgba_sha256"] and out["hit"] == rec["hit_sha256"]
    cnt = out["cnt"]
    if int(force) & 1:
        assert cnt["hits_decided_fp32"] == 0 and cnt["tri_exact_tests"] >= cnt["primary_hits"]
    if int(force) & 2:
        assert cnt["pixels_fixed_fp64"] >= cnt["primary_hits"] / 4


def test_orbit_frames_through_begin_end_with_two_scene_objects(rt):
    """Config 5 the way bench.py drives it: consecutive camera positions, frames alternating between two scene
    objects (each with its own stream and work buffers), frame i enqueued before frame i-1 is waited for."""
    names = sorted(n for n in GOLDEN if n.startswith("C5_step"))
    if len(names) < 4:
        pytest.skip("fewer than four orbit digests in fullframe_hashes.json")
    rec0 = GOLDEN[names[0]]
    data = scene_for(rec0, rt)
    scs = [rt.Scene(data), rt.Scene(data)]
    p = scenes.make_params(rec0["width"], rec0["height"], rec0["subdivision"])
    bufs = [buffers(rec0, rt), buffers(rec0, rt)]
    torch.cuda.synchronize()
    pending = None
    for i, name in enumerate(names + [None]):
        if name is not None:
            rgba, hit = bufs[i & 1]
            hit.fill_(-1)
            rgba.zero_()
            ev = torch.cuda.Event()
            ev.record()
            scs[i & 1].wait_event(ev.cuda_event)  # the frame starts after the fills, without a host synchronisation
            scs[i & 1].render_begin(camera_for(GOLDEN[name]), p, rgba.data_ptr(), hit.data_ptr())
        if pending is not None:
            j, pname = pending
            cnt = scs[j & 1].render_end()
            check_digest(pname, GOLDEN[pname], bufs[j & 1][0], bufs[j & 1][1], cnt)
        pending = (i, name) if name is not None else None
    for s in scs:
        s.close()
