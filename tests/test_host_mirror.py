"""The C++ mirror of the reference's Swift API (include/RayTraceHost.hpp) driven the way startRT
drives raytraceWorld (ViewController.swift:137-180); its bitmaps must equal the oracle's."""
import os
import re
import subprocess

import numpy as np
import pytest

from raytracer_amd import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "tests", "cpp", "host_mirror_demo")
DATA = os.path.join(ROOT, "tests", "data")


def test_demo_is_built_and_links_the_product_library():
    if not os.path.exists(DEMO):  # normally built by __graft_entry__.build(); g++ only, no GPU needed
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")])
    assert os.path.exists(DEMO)
    needed = subprocess.check_output(["readelf", "-d", DEMO]).decode()
    assert "librtmi.so" in needed and "liboracle" not in needed


def check_demo_outputs(oracle_mod, load_obj, tmp_path, out):
    assert out.returncode == 0, out.stderr
    # scene 1: the hard-coded startRT scene at 320x240
    got = np.fromfile(tmp_path / "two_sphere.rgba", np.uint8).reshape(240, 320, 4)
    want, _, _ = oracle_mod.render(scenes.two_sphere_scene(), scenes.make_camera(**scenes.VC_CAMERA), scenes.make_params(320, 240))
    assert (got == want).all()
    # scene 2: loadObjFile path + a top-level triangle + a light listed as an object + a sphere
    tris, mats, names, _ = load_obj(os.path.join(DATA, "cube.obj"), os.path.join(DATA, "cube.mtl"))
    c, r = oracle_mod.bounding_sphere(tris)
    m = re.search(r"centre (\S+) (\S+) (\S+) radius (\S+)", out.stdout)
    assert tuple(float(x) for x in m.groups()) == (*c, r), "Sphere(boundingObjects:) arithmetic"
    floor = np.zeros(1, scenes.TRI_DTYPE)
    floor["v"][0] = [[-3, -0.8, -3], [3, -0.8, 3], [3, -0.8, -3]]
    # material table as the mirror builds it: first use order -> cube materials 'front', 'red', then the VC material
    used = [0, 1]
    vc = scenes.material(**scenes.VC_MATERIAL)
    materials = np.concatenate([mats[used], vc])
    floor["material"] = len(used)
    alltris = np.concatenate([tris, floor])
    objs = np.concatenate([scenes.sphere_object(c, r, -1, bounding=True, first=0, count=len(tris)),
                           scenes.triangles_object(len(tris), 1), scenes.light_object(1),
                           scenes.sphere_object((-25.0, 0.0, 0.0), 25.0, len(used))])
    data = scenes.SceneData(lights=np.concatenate([scenes.light(**scenes.L1), scenes.light(**scenes.L2)]),
                            materials=materials, objects=objs, triangles=alltris)
    cam = scenes.make_camera((1.2, 0.9, 3.0), (-0.35, -0.25, -1.0), 8.0)
    want2, _, _ = oracle_mod.render(data, cam, scenes.make_params(128, 128, 0.5))
    got2 = np.fromfile(tmp_path / "cube.rgba", np.uint8).reshape(128, 128, 4)
    assert (got2 == want2).all()
    assert (got2[..., :3] > 0).any()


@pytest.mark.gpu
def test_cpp_call_site_matches_the_oracle(oracle_mod, rt, tmp_path):
    out = subprocess.run([DEMO, str(tmp_path), os.path.join(DATA, "cube.obj"), os.path.join(DATA, "cube.mtl")],
                         capture_output=True, text=True, timeout=120)
    check_demo_outputs(oracle_mod, rt.load_obj, tmp_path, out)


def test_cpp_call_site_matches_the_oracle_on_the_emulated_library(oracle_mod, tmp_path):
    """The same program linked against the CPU-emulated build of the library (tests/emu): the C++ mirror, the host
    side of the ABI and the kernels, without a GPU."""
    import raytracer_amd
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "SAN=", "BUILD=build_ns", "build_ns/librtmi_emu.so"])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp"), "host_mirror_demo_emu"])
    out = subprocess.run([DEMO + "_emu", str(tmp_path), os.path.join(DATA, "cube.obj"), os.path.join(DATA, "cube.mtl")],
                         capture_output=True, text=True, timeout=600)
    check_demo_outputs(oracle_mod, raytracer_amd.load_obj, tmp_path, out)
