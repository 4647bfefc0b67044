"""Known-answer tests for the CPU oracle, derived by hand from the reference source.

The reference has no tests; each case cites the lines its expected value follows from.
RTF = RayTracingFunctions.swift, REN = Renderables.swift, WCS = WorldCoordinateSequence.swift,
UTL = Utilities.swift, VC = ViewController.swift (all under /root/reference/RayTraceUI/).
"""
import math

import numpy as np
import pytest

from raytracer_amd import scenes

TRI = [[0, 0, 0], [1, 0, 0], [0, 1, 0]]


def test_plane_vectors_default_direction(oracle_mod):
    # RTF:49-54 with dir (0,0,-1): n=(0,0,1), up=(0,1,-1)/sqrt2, u=normalize(up x n)=(1,0,0), v=n x u=(0,1,0)
    u, v = oracle_mod.plane_vectors([0, 0, -1])
    assert u == [1.0, 0.0, 0.0]
    assert [abs(x) for x in v] == [0.0, 1.0, 0.0] and v[1] == 1.0


def test_sphere_two_roots(oracle_mod):
    # REN:251-281: a=5, delta=1 -> t=4 at (0,0,1) n=(0,0,1); t=6 at (0,0,-1) n=(0,0,-1)
    hits = oracle_mod.sphere_intersections([0, 0, 0], 1, [0, 0, 5], [0, 0, -1])
    assert hits == [(4.0, [0.0, 0.0, 1.0], [0.0, 0.0, 1.0]), (6.0, [0.0, 0.0, -1.0], [0.0, 0.0, -1.0])]


def test_sphere_from_inside_keeps_positive_root_only(oracle_mod):
    # roots -1, +1; REN:274 keeps >= 1e-7
    hits = oracle_mod.sphere_intersections([0, 0, 0], 1, [0, 0, 0], [0, 0, -1])
    assert len(hits) == 1 and hits[0][0] == 1.0


def test_sphere_miss_and_tangent(oracle_mod):
    assert oracle_mod.sphere_intersections([0, 0, 0], 1, [2, 0, 5], [0, 0, -1]) == []
    # tangent: delta == 0 -> single root a (REN:262-263)
    hits = oracle_mod.sphere_intersections([0, 0, 0], 1, [1, 0, 5], [0, 0, -1])
    assert len(hits) == 1 and hits[0][0] == 5.0


def test_sphere_epsilon_filter(oracle_mod):
    # origin on the surface looking outward: roots 0 (filtered, < 1e-7) and -2
    assert oracle_mod.sphere_intersections([0, 0, 0], 1, [0, 0, 1], [0, 0, 1]) == []


def test_triangle_front_hit(oracle_mod):
    # REN:142-165: n=(0,0,1), t=1
    t, P, n = oracle_mod.triangle_intersection(TRI, [0.25, 0.25, 1], [0, 0, -1])
    assert (t, P, n) == (1.0, [0.25, 0.25, 0.0], [0.0, 0.0, 1.0])


def test_triangle_back_face_culled(oracle_mod):
    # REN:146: nddot = -1 <= 0
    assert oracle_mod.triangle_intersection(TRI, [0.25, 0.25, -1], [0, 0, 1]) is None


def test_triangle_edge_inclusive(oracle_mod):
    # REN:161: dp(a, n) == 0 passes `>= 0`
    assert oracle_mod.triangle_intersection(TRI, [0.5, 0, 1], [0, 0, -1]) is not None
    # just outside the same edge
    assert oracle_mod.triangle_intersection(TRI, [0.5, -1e-9, 1], [0, 0, -1]) is None


def test_triangle_behind_origin_and_parallel(oracle_mod):
    # REN:152 t > 0 without epsilon: plane behind the origin
    assert oracle_mod.triangle_intersection(TRI, [0.25, 0.25, -1], [0, 0, -1]) is None
    # origin exactly on the plane: t == 0 is not > 0
    assert oracle_mod.triangle_intersection(TRI, [0.25, 0.25, 0], [0, 0, -1]) is None
    # parallel ray: nddot == 0
    assert oracle_mod.triangle_intersection(TRI, [0.25, 0.25, 1], [1, 0, 0]) is None


def test_triangle_degenerate_never_hit(oracle_mod):
    # zero-area triangle: normalize(0) = NaN, every comparison false
    assert oracle_mod.triangle_intersection([[0, 0, 0], [1, 0, 0], [2, 0, 0]], [0.5, 0, 1], [0, 0, -1]) is None


def test_light_as_object(oracle_mod):
    # REN:84-90 + UTL:155-157: along an axis two quotients are 0/0 = NaN -> no hit
    assert oracle_mod.light_intersection([0, 0, -3], [0, 0, 0], [0, 0, -1]) is None
    # all three quotients equal and finite -> hit at t
    d = [1 / math.sqrt(3)] * 3
    hit = oracle_mod.light_intersection([2, 2, 2], [0, 0, 0], d)
    assert hit is not None and hit[0] == 2 / d[0]
    # behind the origin
    assert oracle_mod.light_intersection([-2, -2, -2], [0, 0, 0], d) is None


def test_under_bounding_sphere(oracle_mod):
    # REN:235-241: centre = AABB midpoint, radius = max extent / 2 -> vertex (0,0,0) is outside
    tris = scenes.triangles_from_vertices(np.array(TRI, float), [[0, 1, 2]])
    c, r = oracle_mod.bounding_sphere(tris)
    assert c == (0.5, 0.5, 0.0) and r == 0.5
    assert math.dist(c, (0, 0, 0)) > r


def test_pixel_samples_order(oracle_mod):
    # WCS:74-78: x outer, y inner; sub 1 -> one sample at the pixel corner
    assert oracle_mod.pixel_samples(3, 4, 1.0) == [(3.0, 4.0)]
    assert oracle_mod.pixel_samples(3, 4, 0.5) == [(3.0, 4.0), (3.0, 4.5), (3.5, 4.0), (3.5, 4.5)]
    s = oracle_mod.pixel_samples(0, 0, 0.25)
    assert len(s) == 16 and s[1] == (0.0, 0.25) and s[4] == (0.25, 0.0) and s[15] == (0.75, 0.75)
    # a subdivision that does not divide 1: ceil(1/0.3) = 4 values per axis
    assert len(oracle_mod.pixel_samples(0, 0, 0.3)) == 16


def test_pixel_to_world_scaling(oracle_mod):
    # WCS:58-65 with W = H = 1000, camera (0,0,40) dir (0,0,-1) f=1: ul = (-5, 5, 39),
    # pixels per world unit = 999/10, pixel (i, j) -> ul + u*i/99.9 - v*j/99.9
    cam = scenes.make_camera(**scenes.VC_CAMERA)
    ul, u, v, ph, pv = oracle_mod.frame(cam, 1000, 1000)
    assert ul == [-5.0, 5.0, 39.0] and ph == 999 / 10 and pv == 999 / 10
    p = oracle_mod.pixel_to_world(ul, u, v, ph, pv, 999, 999)
    assert p == pytest.approx([5.0, -5.0, 39.0], abs=1e-12)
    p = oracle_mod.pixel_to_world(ul, u, v, ph, pv, 10, 20)
    assert p[0] == -5.0 + (1.0 * 10) / 99.9 and p[1] == 5.0 - (1.0 * 20) / 99.9


def test_iteration_covers_w_minus_1_by_h_minus_1(oracle_mod):
    # WCS:69-71,82-87 with RTF:170: last row and last column keep their initial bytes
    data = scenes.single_sphere_scene()
    cam = scenes.make_camera((0, 0, 30), (0, 0, -1), 10.0)  # sphere fills the frame
    rgba, hit, cnt = oracle_mod.render(data, cam, scenes.make_params(4, 4))
    assert cnt["samples"] == 9
    assert (rgba[3, :, :] == 0).all() and (rgba[:, 3, :] == 0).all()
    assert (rgba[:3, :3, 3] == 255).all()
    assert (hit[3, :, :] == -1).all() and (hit[:3, :3, 0] == -2).all()


def test_quantise_truncates(oracle_mod):
    # RTF:108-111: UInt8(255 * c) truncates; a sphere with Ka = 0.999/0.2-scaled ambient and no lights
    def render_ambient(ka, amb):
        d = scenes.single_sphere_scene()
        d.lights = d.lights[:0]
        d.materials["Ka"] = ka
        cam = scenes.make_camera((0, 0, 30), (0, 0, -1), 10.0)
        rgba, _, _ = oracle_mod.render(d, cam, scenes.make_params(3, 3, ambient=(amb,) * 3))
        return rgba[0, 0]

    # no lights and the reflection ray leaves the only sphere -> I = Ka * ambient exactly
    assert list(render_ambient(0.999, 1.0)) == [254, 254, 254, 255]
    assert list(render_ambient(1.0, 1.0)) == [255, 255, 255, 255]
    assert list(render_ambient(0.5, 1.0)) == [127, 127, 127, 255]


def test_list_order_not_nearest(oracle_mod):
    # RTF:83/93: where both triangles are hit, the far one (listed first) wins over the nearer one
    from golden_util import load_golden
    data, cam, params, rgba, hit, _ = load_golden("two_tri_list_order_32x32")
    _, h, _ = oracle_mod.render(data, cam, params)
    ul, u, v, ph, pv = oracle_mod.frame(cam, params.width, params.height)
    both = 0
    for y in range(params.height - 1):
        for x in range(params.width - 1):
            p = np.array(oracle_mod.pixel_to_world(ul, u, v, ph, pv, x, y))
            d = p - np.array(list(cam.pos))
            d = d / math.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])
            far = oracle_mod.triangle_intersection(data.triangles["v"][0].tolist(), list(cam.pos), d.tolist())
            near = oracle_mod.triangle_intersection(data.triangles["v"][1].tolist(), list(cam.pos), d.tolist())
            if far and near:
                both += 1
                assert near[0] < far[0] and h[y, x, 0] == 0
            elif near:
                assert h[y, x, 0] == 1
    assert both > 20


def test_screenshot_extents(oracle_mod):
    # RayTraceExample.png: camera (0,0,40), dir (0,0,-1), f=1, ambient .2, two spheres of VC:155-167.
    # The discs span about 0.41 of the half-width and 0.16 of the half-height and meet at the centre.
    data = scenes.two_sphere_scene()
    cam = scenes.make_camera(**scenes.VC_CAMERA)
    _, hit, cnt = oracle_mod.render(data, cam, scenes.make_params(1000, 1000))
    assert cnt["samples"] == 998001  # the UI's 1,000,000 label counts W*H; (W-1)(H-1) are rendered
    ys, xs = np.nonzero(hit[:, :, 0] != -1)
    half = 499.5
    assert (xs.max() - half) / half == pytest.approx(0.41, abs=0.01)
    assert (half - xs.min()) / half == pytest.approx(0.41, abs=0.01)
    assert (ys.max() - half) / half == pytest.approx(0.16, abs=0.01)
    # both spheres are visible, left one is object 1 (-25,0,0), right one object 0
    assert set(np.unique(hit)) == {-1, -2, -3}
    assert (hit[500, 600, 0], hit[500, 400, 0]) == (-2, -3)


def test_oracle_threads_are_deterministic(oracle_mod):
    data = scenes.mesh_scene(2, "B", 3, bounding_sphere=oracle_mod.bounding_sphere)
    cam = scenes.make_camera(**scenes.MESH_CAMERA)
    p = scenes.make_params(64, 48, 0.5)
    a = oracle_mod.render(data, cam, p, threads=1)
    b = oracle_mod.render(data, cam, p, threads=4)
    assert (a[0] == b[0]).all() and (a[1] == b[1]).all() and a[2] == b[2]
