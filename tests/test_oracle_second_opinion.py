"""The C oracle (oracle/rt_oracle.c) against a second restatement of the reference written separately, object by
object, in plain Python (oracle/swift_walkthrough.py): hit ids, RGBA8 bytes and ray counts must agree exactly on
small frames of every kind of scene the tests use.  Two transcriptions of the same Swift sources that agree bit for
bit leave little room for a slip in either; what neither can rule out is a shared misreading, which is what the
hand-derived known-answer tests (tests/test_oracle_kat.py) are for."""
import numpy as np
import pytest

from golden_util import load_golden
from oracle import swift_walkthrough as sw
from raytracer_amd import scenes
from scene_cases import light_hit_scene, shadow_extension_cases


def agree(oracle_mod, data, cam, params):
    o_rgba, o_hit, o_cnt = oracle_mod.render(data, cam, params)
    rgba, hit, cnt = sw.render(data, cam, params)
    W, H = params.width, params.height
    for y in range(H):
        for x in range(W):
            if x < W - 1 and y < H - 1:
                assert hit[y][x] == list(o_hit[y, x]), (x, y, hit[y][x], o_hit[y, x])
                assert rgba[y][x] == list(o_rgba[y, x]), (x, y, rgba[y][x], o_rgba[y, x])
            else:
                assert rgba[y][x] is None  # the reference renders (W-1) x (H-1) pixels
    for k in ("samples", "rays_reflection", "rays_shadow", "primary_hits"):
        assert cnt[k] == o_cnt[k], k
    return o_cnt


@pytest.mark.parametrize("name,size", [("single_sphere_64x48", (24, 18, 1.0)), ("two_sphere_64x48", (26, 20, 0.5)),
                                       ("two_tri_list_order_32x32", (16, 16, 0.5)), ("mixed_3L_64x48", (22, 16, 0.5)),
                                       ("ico2_bounded_s025_48x40", (12, 10, 0.25)), ("ico2_toplevel_3L_40x40", (12, 12, 1.0))])
def test_second_restatement_agrees_on_fixture_scenes(oracle_mod, name, size):
    data, cam, params, _, _, _ = load_golden(name)
    p = scenes.make_params(size[0], size[1], size[2], tuple(params.ambient))
    cnt = agree(oracle_mod, data, cam, p)
    assert cnt["primary_hits"] > 0


def test_second_restatement_agrees_on_odd_subdivisions_and_cameras(oracle_mod):
    data = scenes.mesh_scene(level=1, form="B", n_lights=3, bounding_sphere=oracle_mod.bounding_sphere)
    data.objects = np.concatenate([data.objects, scenes.sphere_object((0.9, 0.6, 0.8), 0.35, 0)])
    for sub, cam in ((0.4, scenes.make_camera((0, 0, 4), (0, 0, -1), 10.0)),
                     (0.3, scenes.make_camera((1.5, 0.4, 3.5), (-0.9, -0.2, -2.1), 6.0)),     # direction not normalised
                     (0.7, scenes.make_camera((0.05, 0.02, 0.1), (0.1, 0.3, -1.0), 2.0))):   # inside the mesh
        agree(oracle_mod, data, cam, scenes.make_params(13, 11, sub, (0.3, 0.1, 0.25)))


def test_second_restatement_agrees_on_point_light_hits(oracle_mod):
    data, cam, params, pixels = light_hit_scene(oracle_mod)
    agree(oracle_mod, data, cam, params)
    _, hit, _ = sw.render(data, cam, params)
    assert len({hit[j][i][0] for i, j in pixels} & {-2, -3, -5, -6}) >= 3


def test_second_restatement_builds_the_same_bounding_sphere(oracle_mod):
    """Sphere(boundingObjects:) (Renderables.swift:215-246), once from each restatement."""
    rng = np.random.default_rng(3)
    tris = np.zeros(50, scenes.TRI_DTYPE)
    tris["v"] = rng.normal(size=(50, 3, 3)) * [2.0, 0.7, 1.3] + [0.5, -1.0, 2.0]
    c, r = oracle_mod.bounding_sphere(tris)
    shapes = [sw.Triangle([sw.V3(*t["v"][k]) for k in range(3)], None, i) for i, t in enumerate(tris)]
    s = sw.Sphere.bounding_objects(shapes, None, -2)
    assert (s.location.x, s.location.y, s.location.z) == tuple(c) and s.radius == r


def test_second_restatement_agrees_on_random_soup(oracle_mod):
    rng = np.random.default_rng(17)
    tris = np.zeros(60, scenes.TRI_DTYPE)
    tris["v"] = rng.uniform(-1.2, 1.2, size=(60, 1, 3)) + rng.normal(size=(60, 3, 3)) * 0.5
    tris["material"] = rng.integers(-1, 2, size=60)
    mats = np.concatenate([scenes.material(**scenes.VC_MATERIAL),
                           scenes.material(Ns=3.5, d=1.0, illum=2, Kd=(0.8, 0.3, 0.2), Ka=(0.2, 0.5, 0.9), Ks=(0.4, 0.9, 0.1))])
    lights = np.concatenate([scenes.light(**scenes.L1), scenes.light(**scenes.L2), scenes.light(**scenes.L3)])
    c, r = oracle_mod.bounding_sphere(tris[:35])
    objects = np.concatenate([scenes.sphere_object(c, r, 1, bounding=True, first=0, count=35),
                              scenes.sphere_object((0.3, -0.8, 0.2), 0.5, 0),
                              scenes.triangles_object(35, 25), scenes.light_object(2)])
    data = scenes.SceneData(lights=lights, materials=mats, objects=objects, triangles=tris)
    agree(oracle_mod, data, scenes.make_camera((0.2, 0.3, 4), (0, -0.05, -1), 7.0), scenes.make_params(18, 14, 0.5))


def test_second_restatement_agrees_on_generated_scenes(oracle_mod):
    """Scenes of the emulated-kernel fuzzer (degenerate and non-finite triangles, empty groups, spheres that do not
    bound, cameras inside the geometry), whole small frames."""
    import emu_fuzz
    done = 0
    for i in range(40):
        data, cam, params = emu_fuzz.random_case(21, i)
        if len(data.triangles) > 150:
            continue
        p = scenes.make_params(min(params.width, 10), min(params.height, 8), params.pixel_subdivision, tuple(params.ambient))
        agree(oracle_mod, data, cam, p)
        done += 1
    assert done >= 20


def test_second_restatement_agrees_on_the_nearest_hit_extension(oracle_mod):
    """RTMI_HIT_NEAREST in the C oracle against `min(intersections, key=parameter)` in the Python restatement, on
    scenes where the two policies differ: a non-convex bounded mesh, overlapping spheres, top-level triangles."""
    from raytracer_amd import _abi
    data, cam, params, _, _, _ = load_golden("mixed_3L_64x48")
    differs = 0
    for d, c, size in ((data, cam, (22, 16, 0.5)),
                       (scenes.mesh_scene(1, "B", 2, bounding_sphere=oracle_mod.bounding_sphere, shuffle_seed=5),
                        scenes.make_camera(**scenes.MESH_CAMERA), (14, 12, 0.5)),
                       (scenes.mesh_scene(1, "T", 3, shuffle_seed=6), scenes.make_camera((0.6, 0.4, 3.5), (-0.15, -0.1, -1.0), 8.0), (12, 12, 1.0)),
                       (scenes.two_sphere_scene(), scenes.make_camera((60.0, 10.0, 40.0), (-1.0, -0.2, -0.6), 2.0), (20, 16, 1.0))):
        p = scenes.make_params(size[0], size[1], size[2], hit_policy=_abi.RTMI_HIT_NEAREST)
        agree(oracle_mod, d, c, p)
        a = oracle_mod.render(d, c, p)[1]
        b = oracle_mod.render(d, c, scenes.make_params(size[0], size[1], size[2]))[1]
        differs += int((a != b).sum())
    assert differs > 0  # the extension does change hits on these scenes


@pytest.mark.parametrize("flags", [2, 4, 6, 8, 14])
def test_second_restatement_agrees_on_the_shadow_extensions(oracle_mod, flags):
    """RTMI_FLAG_SHADOW_MAX_DISTANCE (2), RTMI_FLAG_SHADOWS_ENTER_BOUNDED (4) and RTMI_FLAG_REFLECTIONS_ENTER_BOUNDED (8) in
    the C oracle against their statement in the Python restatement (shadow_blockers; the reflection branch of
    calculate_lighting), with both hit policies; each flag must change pixels."""
    changed = 0
    for data, cam in shadow_extension_cases(oracle_mod.bounding_sphere):
        for policy in (0, 1):
            p = scenes.make_params(16, 14, 0.5, hit_policy=policy, flags=flags)
            agree(oracle_mod, data, cam, p)
            a = oracle_mod.render(data, cam, p)[0]
            b = oracle_mod.render(data, cam, scenes.make_params(16, 14, 0.5, hit_policy=policy))[0]
            changed += int((a != b).any(axis=-1).sum())
    assert changed > 0
