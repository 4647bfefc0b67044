"""The C-ABI library loads without a GPU and exports everything include/rtmi.h declares."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd import _abi, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rtmi.h")


def test_header_declares_exactly_the_bound_symbols():
    text = open(HEADER).read()
    declared = set(re.findall(r"\b(rtmi_[a-z_]+)\s*\(", text))
    assert declared == set(_abi.EXPORTS)


def test_library_exports_every_declared_symbol():
    L = rt.lib()
    for name in _abi.EXPORTS:
        assert hasattr(L, name), name
    assert L.rtmi_abi_version() == _abi.RTMI_ABI_VERSION


def test_struct_layouts_match_the_c_compiler():
    names = ["rtmi_camera", "rtmi_light", "rtmi_material", "rtmi_triangle", "rtmi_object",
             "rtmi_scene_desc", "rtmi_params", "rtmi_counters"]
    py = [_abi.Camera, _abi.Light, _abi.Material, _abi.Triangle, _abi.Object, _abi.SceneDesc,
          _abi.Params, _abi.Counters]
    src = '#include <stdio.h>\n#include "rtmi.h"\nint main(void){' + "".join(
        f'printf("%zu\\n", sizeof({n}));' for n in names) + "return 0;}"
    with tempfile.TemporaryDirectory() as td:
        cfile, exe = os.path.join(td, "s.c"), os.path.join(td, "s")
        open(cfile, "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), cfile, "-o", exe])
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    assert sizes == [C.sizeof(t) for t in py]


def test_header_is_plain_c():
    with tempfile.TemporaryDirectory() as td:
        cfile = os.path.join(td, "h.c")
        open(cfile, "w").write('#include "rtmi.h"\nint x;\n')
        subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I",
                               os.path.join(ROOT, "include"), "-c", cfile, "-o", os.path.join(td, "h.o")])


def test_status_strings():
    L = rt.lib()
    assert L.rtmi_status_string(0) == b"ok"
    for code in (-1, -2, -3, -4, -5, -99):
        assert L.rtmi_status_string(code)


def test_samples_per_pixel_matches_stride_semantics(oracle_mod):
    for s in (1.0, 0.5, 0.25, 0.3, 0.125, 0.7):
        assert rt.samples_per_pixel(s) == len(oracle_mod.pixel_samples(0, 0, s))
    for bad in (0.0, -1.0, 1.5):
        with pytest.raises(rt.RtmiError):
            rt.samples_per_pixel(bad)


def test_bounding_sphere_host_arithmetic_matches_oracle(oracle_mod):
    # rtmi_bounding_sphere is pure host arithmetic (Renderables.swift:215-246): no GPU needed
    rng = np.random.default_rng(7)
    for n in (1, 2, 17, 500):
        tris = np.zeros(n, scenes.TRI_DTYPE)
        tris["v"] = rng.normal(size=(n, 3, 3)) * rng.uniform(0.1, 30.0)
        assert rt.bounding_sphere(tris) == oracle_mod.bounding_sphere(tris)
    v, f = scenes.displaced_icosphere(3)
    tris = scenes.triangles_from_vertices(v, f)
    assert rt.bounding_sphere(tris) == oracle_mod.bounding_sphere(tris)


def test_invalid_arguments_are_rejected_before_touching_a_device():
    L = rt.lib()
    h = C.c_void_p()
    assert L.rtmi_scene_create(None, 0, C.byref(h)) == _abi.RTMI_ERR_INVALID_ARGUMENT
    data = scenes.two_sphere_scene()
    data.objects["material"][0] = 5  # out of range
    desc = data.desc()
    assert L.rtmi_scene_create(C.byref(desc), 0, C.byref(h)) == _abi.RTMI_ERR_INVALID_ARGUMENT
    data = scenes.mesh_scene(1, "T")
    data.objects["count"][0] = 10 ** 6  # run past the triangle array
    desc = data.desc()
    assert L.rtmi_scene_create(C.byref(desc), 0, C.byref(h)) == _abi.RTMI_ERR_INVALID_ARGUMENT
    assert not h.value


def test_no_silent_cpu_fallback():
    """Without a device, scene creation fails with RTMI_ERR_NO_DEVICE rather than computing on the host."""
    L = rt.lib()
    if L.rtmi_device_count() >= 1:
        pytest.skip("a GPU is present")
    with pytest.raises(rt.RtmiError) as ei:
        rt.Scene(scenes.two_sphere_scene())
    assert ei.value.status == _abi.RTMI_ERR_NO_DEVICE


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "raytracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp", ".hpp")) or fn == "Makefile":
                text = open(os.path.join(dirpath, fn)).read()
                assert "import oracle" not in text and "from oracle" not in text, fn
                assert "rt_oracle" not in text and "liboracle" not in text, fn
