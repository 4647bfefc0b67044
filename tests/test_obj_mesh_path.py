"""The path a real mesh takes: OBJ/MTL file -> the library's loader (readObjFile, readMtlFile, createTriangleList:
ObjLoader.swift:160-201, :95-158, ViewController.swift:107-120) -> loadObjFile's wrapper (ViewController.swift:101-104,
rtmi_scene_from_obj) -> frame.  The Crane pig cannot be fetched here; tools/write_mesh_obj.py writes the stand-in mesh as
such a file instead, and the file must render to the stand-in's frames: the whole-frame digests of
tests/golden/fullframe_hashes.json, through rtmi_scene_from_obj and through bench.py's --obj / RTMI_PIG_OBJ branch."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "fullframe_hashes.json")))


def test_written_obj_loads_to_the_generated_triangles(tmp_path):
    """Host code only: the file's triangles and material are the generated mesh's, bit for bit."""
    import raytracer_amd as rt
    import write_mesh_obj
    from raytracer_amd import scenes
    for level in (0, 3):
        obj, mtl, n = write_mesh_obj.write_mesh(str(tmp_path / f"mesh{level}.obj"), level)
        tris, mats, names, nv = rt.load_obj(obj, mtl)
        ref = scenes.mesh_scene(level, "T", 1)
        assert n == len(tris) == 20 * 4 ** level and names == ["pig"]
        assert tris.tobytes() == ref.triangles.tobytes() and mats.tobytes() == ref.materials.tobytes()


@pytest.mark.gpu
def test_obj_file_through_scene_from_obj_gives_the_stand_ins_digest(rt, tmp_path):
    import torch
    import write_mesh_obj
    from raytracer_amd import scenes
    rec = GOLDEN["C2"]
    obj, mtl, n = write_mesh_obj.write_mesh(str(tmp_path / "pig.obj"), rec["mesh_level"])
    lights = np.concatenate([scenes.light(**L) for L in (scenes.L1, scenes.L2, scenes.L3)[:rec["lights"]]])
    sc = rt.Scene.from_obj(obj, mtl, lights)
    W, H = rec["width"], rec["height"]
    spp = rt.samples_per_pixel(rec["subdivision"])
    rgba = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda:0")
    hit = torch.full((H, W, spp), -1, dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    cnt = sc.render_device(scenes.make_camera(**scenes.MESH_CAMERA), scenes.make_params(W, H, rec["subdivision"]), rgba.data_ptr(), hit.data_ptr())
    sc.close()
    assert hashlib.sha256(hit.cpu().numpy().tobytes()).hexdigest() == rec["hit_sha256"]
    assert hashlib.sha256(rgba.cpu().numpy().tobytes()).hexdigest() == rec["rgba_sha256"]
    assert cnt["primary_hits"] == rec["counters"]["primary_hits"] and cnt["rays_shadow"] == rec["counters"]["rays_shadow"]


@pytest.mark.gpu
def test_bench_obj_branch_renders_the_file(tmp_path):
    """bench.py with RTMI_PIG_OBJ set (the branch a real pig mesh would take): the line names the file and its last frame
    has the digest of the stand-in's C2 frame."""
    import write_mesh_obj
    rec = GOLDEN["C2"]
    obj, _, n = write_mesh_obj.write_mesh(str(tmp_path / "pig.obj"), rec["mesh_level"])
    env = dict(os.environ, RTMI_PIG_OBJ=obj)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "C2", "--static-camera", "--steps", "2", "--warmup", "1",
                        "--no-aux", "--no-cpu"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert "pig.obj" in d["config"]["mesh"] and f"{n} triangles" in d["config"]["mesh"]
    assert d["last_frame_sha256"] == rec["rgba_sha256"]


@pytest.mark.gpu
def test_large_obj_file_renders_like_the_generated_mesh(rt, oracle_mod, tmp_path):
    """327,680 triangles through the file path: the frame equals the hand-built scene's, and a window of it the oracle's."""
    import torch
    import write_mesh_obj
    from raytracer_amd import scenes
    level, W, H, sub = 7, 640, 360, 0.5
    obj, mtl, n = write_mesh_obj.write_mesh(str(tmp_path / "big.obj"), level)
    assert n == 327680
    data = scenes.mesh_scene(level, "B", 2, bounding_sphere=rt.bounding_sphere)
    cam = scenes.make_camera(**scenes.MESH_CAMERA)
    p = scenes.make_params(W, H, sub)
    frames = []
    for sc in (rt.Scene.from_obj(obj, mtl, data.lights), rt.Scene(data)):
        rgba = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda:0")
        hit = torch.full((H, W, 4), -1, dtype=torch.int32, device="cuda:0")
        torch.cuda.synchronize()
        sc.render_device(cam, p, rgba.data_ptr(), hit.data_ptr())
        frames.append((rgba.cpu().numpy(), hit.cpu().numpy()))
        sc.close()
    assert (frames[0][0] == frames[1][0]).all() and (frames[0][1] == frames[1][1]).all()
    assert (frames[0][1] >= 0).sum() > 100000
    win = (300, 170, 332, 186)
    o_rgba, o_hit, _ = oracle_mod.render(data, cam, scenes.make_params(W, H, sub, window=win), threads=8)
    x0, y0, x1, y1 = win
    assert (frames[0][1][y0:y1, x0:x1] == o_hit[y0:y1, x0:x1]).all() and (frames[0][0][y0:y1, x0:x1] == o_rgba[y0:y1, x0:x1]).all()
