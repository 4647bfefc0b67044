"""OBJ / MTL ingest (rtmi_obj_load): the reference's parsing rules (ObjLoader.swift:61-63, :95-201;
ViewController.swift:107-120) checked on small files written for this test.  Host only, no GPU.

No OBJ file ships with the reference, so these expectations are derived from its source, not from
fixtures of its own: parity unpinned by reference artefacts."""
import os

import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd import _abi

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def test_cube_triangles_and_materials():
    tris, mats, names, nverts = rt.load_obj(os.path.join(DATA, "cube.obj"), os.path.join(DATA, "cube.mtl"))
    assert nverts == 8, "an indented 'v' line does not start with 'v ' and is ignored"
    assert len(tris) == 11
    assert names == ["front", "red", "front2"]
    # 1-based -> 0-based, explicit vertex copies (ViewController.swift:110-116)
    assert tris["v"][0].tolist() == [[-0.5, -0.5, 0.5], [0.5, -0.5, 0.5], [0.5, 0.5, 0.5]]
    # a quad keeps its first three indices (ObjLoader.swift:193-195)
    assert tris["v"][6].tolist() == [[-0.5, -0.5, -0.5], [-0.5, -0.5, 0.5], [-0.5, 0.5, 0.5]]
    # usemtl takes the LAST token of the line (ObjLoader.swift:173); unknown names map to -1
    assert tris["material"].tolist() == [0, 0, 1, 1, -1, -1, 0, 0, 0, 0, 0]
    assert mats["Ns"].tolist() == [8.0, 32.5, 1.0] and mats["illum"].tolist() == [2, 1, 0]
    assert mats["Ks"][0].tolist() == [1.0, 1.0, 0.0], "colours are clamped to [0,1] like RGB.init (Utilities.swift:89-92)"
    assert mats["Ka"][1].tolist() == [0.2, 0.0, 0.0] and mats["d"][1] == 0.5


def test_obj_without_mtl_gives_nil_materials():
    tris, mats, names, _ = rt.load_obj(os.path.join(DATA, "cube.obj"))
    assert len(mats) == 0 and names == [] and (tris["material"] == -1).all()


@pytest.mark.parametrize("body,why", [
    ("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1/1/1 2/2/2 3/3/3\n", "slash tokens are not integers (the reference traps)"),
    ("v 0 0 0\r\nv 1 0 0\r\nv 0 1 0\r\nf 1 2 3\r\n", "CRLF leaves '\\r' on the last token (the reference traps)"),
    ("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 4\n", "vertex index out of range"),
    ("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 0 1 2\n", "index 0 is before the first vertex"),
    ("v 0 0\n", "vertex with two numbers"),
    ("v 0 0 zero\n", "unparsable number"),
    ("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2\n", "face with two indices"),
])
def test_malformed_obj_is_an_error_not_a_crash(tmp_path, body, why):
    p = tmp_path / "bad.obj"
    p.write_bytes(body.encode())
    with pytest.raises(rt.RtmiError) as ei:
        rt.load_obj(str(p))
    assert ei.value.status == _abi.RTMI_ERR_INVALID_ARGUMENT, why


def test_malformed_mtl(tmp_path):
    obj = tmp_path / "a.obj"
    obj.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    mtl = tmp_path / "a.mtl"
    mtl.write_text("newmtl m\nNs 1\nd 1\nillum 2\nKd 1 1 1\nKa 1 1 1\n")  # Ks missing: the reference traps
    with pytest.raises(rt.RtmiError):
        rt.load_obj(str(obj), str(mtl))
    mtl.write_text("newmtl m\nNs 1\nd 1\nillum 2.5\nKd 1 1 1\nKa 1 1 1\nKs 1 1 1\n")  # illum is an Int
    with pytest.raises(rt.RtmiError):
        rt.load_obj(str(obj), str(mtl))
    mtl.write_text("newmtl m\nNs 1\nd 1\nillum 2\nKd 1 1\nKa 1 1 1\nKs 1 1 1\n")  # RGB needs three numbers
    with pytest.raises(rt.RtmiError):
        rt.load_obj(str(obj), str(mtl))
    with pytest.raises(rt.RtmiError):
        rt.load_obj(str(tmp_path / "missing.obj"))


def test_repeated_material_name_replaces_the_earlier_one(tmp_path):
    obj = tmp_path / "a.obj"
    obj.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl m\nf 1 2 3\n")
    mtl = tmp_path / "a.mtl"
    block = "newmtl m\nNs {ns}\nd 1\nillum 2\nKd 1 1 1\nKa 1 1 1\nKs 1 1 1\n"
    mtl.write_text(block.format(ns=1) + block.format(ns=7))
    tris, mats, names, _ = rt.load_obj(str(obj), str(mtl))
    assert names == ["m"] and mats["Ns"].tolist() == [7.0] and tris["material"].tolist() == [0]


@pytest.mark.gpu
def test_loaded_cube_renders_like_the_oracle(oracle_mod):
    """ObjLoader output feeds the renderer unchanged: load, wrap in Sphere(boundingObjects:), render."""
    from raytracer_amd import scenes
    tris, mats, names, _ = rt.load_obj(os.path.join(DATA, "cube.obj"), os.path.join(DATA, "cube.mtl"))
    c, r = rt.bounding_sphere(tris)
    data = scenes.SceneData(lights=np.concatenate([scenes.light(**scenes.L1), scenes.light(**scenes.L3)]), materials=mats,
                            objects=scenes.sphere_object(c, r * 2.0, -1, bounding=True, first=0, count=len(tris)), triangles=tris)
    cam = scenes.make_camera((1.2, 0.9, 3.0), (-0.35, -0.25, -1.0), 8.0)
    params = scenes.make_params(96, 96, 0.5)
    o_rgba, o_hit, _ = oracle_mod.render(data, cam, params)
    sc = rt.Scene(data)
    g_rgba, g_hit, _ = sc.render(cam, params)
    sc.close()
    assert (g_hit == o_hit).all() and (g_rgba == o_rgba).all()
    assert len(set(np.unique(o_hit)) - {-1}) >= 3, "several cube faces must be visible"


def test_scene_from_obj_rejects_bad_input_without_a_device(tmp_path):
    # argument and parse errors are reported before any device is touched
    with pytest.raises(rt.RtmiError) as ei:
        rt.Scene.from_obj(str(tmp_path / "missing.obj"))
    assert ei.value.status == _abi.RTMI_ERR_INVALID_ARGUMENT
    bad = tmp_path / "bad.obj"
    bad.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1/1 2/2 3/3\n")
    with pytest.raises(rt.RtmiError):
        rt.Scene.from_obj(str(bad))


@pytest.mark.gpu
def test_scene_from_obj_equals_the_hand_built_scene(oracle_mod):
    """rtmi_scene_from_obj = loadObjFile: load + Sphere(boundingObjects:) + upload; same frame as the
    scene assembled by hand, and as the oracle's."""
    from raytracer_amd import scenes
    lights = np.concatenate([scenes.light(**scenes.L1), scenes.light(**scenes.L3)])
    tris, mats, names, _ = rt.load_obj(os.path.join(DATA, "cube.obj"), os.path.join(DATA, "cube.mtl"))
    c, r = rt.bounding_sphere(tris)
    data = scenes.SceneData(lights=lights, materials=mats,
                            objects=scenes.sphere_object(c, r, -1, bounding=True, first=0, count=len(tris)), triangles=tris)
    cam = scenes.make_camera((1.2, 0.9, 3.0), (-0.35, -0.25, -1.0), 8.0)
    params = scenes.make_params(96, 96, 0.5)
    o_rgba, o_hit, _ = oracle_mod.render(data, cam, params)
    sc = rt.Scene.from_obj(os.path.join(DATA, "cube.obj"), os.path.join(DATA, "cube.mtl"), lights)
    g_rgba, g_hit, _ = sc.render(cam, params)
    sc.close()
    assert (g_hit == o_hit).all() and (g_rgba == o_rgba).all()


@pytest.mark.gpu
def test_command_line_harness_writes_the_rendered_frame(tmp_path):
    from raytracer_amd import render
    from test_image_writer import decode_png
    out = tmp_path / "cube.png"
    cnt = render.main([os.path.join(DATA, "cube.obj"), str(out), "--mtl", os.path.join(DATA, "cube.mtl"), "--size", "120", "90",
                       "--camera", "1.2", "0.9", "3.0", "--direction", "-0.35", "-0.25", "-1.0", "--focal-length", "8",
                       "--subdivision", "0.5"])
    img = decode_png(out)
    assert img.shape == (90, 120, 4) and cnt["primary_hits"] > 0
    assert (img[:89, :119, 3] == 255).all() and img[:, :, :3].max() > 0
