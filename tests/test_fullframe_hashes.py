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
