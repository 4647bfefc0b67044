"""The error bounds behind the pixels-only path of bounding-sphere scenes (kernels_fast.h), checked on the CPU.

That path computes in FP32 and keeps the reference's bytes only because (1) every FP32 quantity carries a bound on its
distance from the reference's FP64 value and (2) whatever the bounds cannot settle is evaluated again in FP64.  The two
bounds everything else is built on:

* the FP32 lighting's running error bound (rtmi_shade32.h): tests/cpp/shade32_check.cpp draws 1.2 * 10^8 random hits and
  compares with the FP64 lighting;
* the bound on the FP32 sample directions (rtmi_direction_error_bound): here the kernel's evaluation (fast_dir:
  float32 tables, two additions, a normalisation by reciprocal square root) is restated in numpy and compared with the
  FP64 directions of WorldCoordinateSequence.swift:62-66 / RayTracingFunctions.swift:73 for random cameras and frames;
* the FP32 "certainly intersects the bounding sphere" test against the FP64 Sphere.intersections for the same rays.

This validates the mathematics and the margins, not the device code itself (the emulated and GPU parity tests do that).
"""
import os
import subprocess

import numpy as np

import raytracer_amd as rt
from raytracer_amd import scenes

HERE = os.path.dirname(os.path.abspath(__file__))
F = np.float32
U = 2.0 ** -24


def test_fp32_lighting_error_bound_holds_on_random_hits():
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "cpp"), "shade32_check"])
    r = subprocess.run([os.path.join(HERE, "cpp", "shade32_check"), "120", "2027"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "120000000 hits" in r.stdout and " 0 errors above their bound" in r.stdout and " 0 of " in r.stdout, r.stdout
    flagged = float(r.stdout.split("flagged (")[1].split(" %")[0])
    assert flagged < 5.0, r.stdout  # the generator's hits include grazing ones and lights a hundredth of a unit away


def normalize64(v):
    return v / np.sqrt((v * v).sum(axis=-1, keepdims=True))


def frame(cam_pos, cam_dir, focal, W, H):
    """raytraceWorld's camera frame in FP64 (RayTracingFunctions.swift:46-57, :156-170)."""
    cam_pos, cam_dir = np.asarray(cam_pos, np.float64), np.asarray(cam_dir, np.float64)
    center = cam_pos + focal * cam_dir
    n = normalize64(-cam_dir)
    up = normalize64(np.array([0.0, 1.0, -1.0]))
    u = normalize64(np.cross(up, n))
    v = np.cross(n, u)
    ul = center - u * 5.0 + v * 5.0
    return ul, u, v, (W - 1) / 10.0, (H - 1) / 10.0


def directions(cam_pos, ul, u, v, ppw_h, ppw_v, xs, ys):
    """(FP64 reference directions, the kernel's FP32 directions) for sample coordinates xs, ys."""
    uc = (u[None, :] * xs[:, None]) / ppw_h   # (u * i) / pixelsPerWorldHorizontal
    vc = (v[None, :] * ys[:, None]) / ppw_v
    p = (ul[None, :] + uc) - vc
    d64 = normalize64(p - np.asarray(cam_pos, np.float64)[None, :])
    base_f = (ul - np.asarray(cam_pos, np.float64)).astype(F)
    f = (base_f[None, :] + uc.astype(F)) - vc.astype(F)  # float32 additions
    dd = (f[:, 0] * f[:, 0] + f[:, 1] * f[:, 1]) + f[:, 2] * f[:, 2]
    inv = (1.0 / np.sqrt(dd.astype(np.float64))).astype(F)
    # v_rsq_f32 is good to one ulp: take the worse neighbour as well
    d32 = [f * np.nextafter(inv, F(k))[:, None] for k in (0.0, np.inf)] + [f * inv[:, None]]
    return d64, d32


def test_fp32_sample_direction_bound_exceeds_observed_errors():
    rng = np.random.default_rng(5)
    worst = 0.0
    for case in range(400):
        scale = float(rng.choice([0.1, 1.0, 4.0, 40.0, 1000.0]))
        pos = rng.normal(size=3) * scale
        look = rng.normal(size=3) * float(rng.choice([0.1, 1.0, 3.0]))
        focal = float(rng.choice([0.2, 1.0, 2.0, 10.0, 50.0]))
        W, H = int(rng.integers(2, 5000)), int(rng.integers(2, 3000))
        cam = scenes.make_camera(tuple(pos), tuple(look), focal)
        bound = rt.direction_error_bound(cam, W, H)
        if not np.isfinite(bound):
            continue
        ul, u, v, ph, pv = frame(pos, look, focal, W, H)
        xs = rng.uniform(0, W - 1, 4000)
        ys = rng.uniform(0, H - 1, 4000)
        xs[:4], ys[:4] = [0, W - 1, 0, W - 1], [0, 0, H - 1, H - 1]
        d64, d32s = directions(pos, ul, u, v, ph, pv, xs, ys)
        for d32 in d32s:
            err = np.sqrt(((d32.astype(np.float64) - d64) ** 2).sum(axis=1)).max()
            assert err <= bound, (case, err, bound)
            worst = max(worst, err / bound)
    assert 0.02 < worst <= 1.0, worst  # the bound is within a factor of fifty of what happens


def test_direction_bound_is_infinite_without_a_focal_distance():
    assert not np.isfinite(rt.direction_error_bound(scenes.make_camera((0, 0, 4), (0, 0, -1), 0.0), 640, 480))
    assert not np.isfinite(rt.direction_error_bound(scenes.make_camera((0, 0, 4), (0, 0, 0), 1.0), 640, 480))


def test_fp32_sphere_test_is_certain_only_when_the_fp64_test_agrees():
    """k_shade_candidates_fast accepts 'the ray intersects the bounding sphere' from FP32 values when a = -d.cte exceeds
    1e-6 + 2 ea and delta = a^2 - k exceeds twice its evaluation error; Sphere.intersections (Renderables.swift:248-282) in FP64
    must then report an intersection.  Rays are drawn around the sphere's silhouette, where the two can differ."""
    rng = np.random.default_rng(11)
    certain = 0
    for case in range(300):
        scale = float(rng.choice([0.5, 4.0, 100.0]))
        cam = rng.normal(size=3) * scale
        centre = cam + rng.normal(size=3) * scale
        dist = np.linalg.norm(cam - centre)
        r = float(rng.uniform(0.05, 0.98)) * dist if rng.random() < 0.8 else float(rng.uniform(1.02, 3.0)) * dist
        dir_err = float(rng.choice([2e-7, 2e-6, 2e-5]))
        cte = cam - centre
        k = float((cte * cte).sum() - r * r)
        ea = F(abs((dir_err + 8 * U) * np.abs(cte).sum()) * 1.0000002)
        kabs = F(abs(k) * 1.0000002)
        # unit directions around the tangent cone of the sphere (or anywhere, for a camera inside it)
        towards = -cte / dist
        sin_t = min(r / dist, 1.0)
        perp = normalize64(np.cross(np.broadcast_to(towards, (20000, 3)), rng.normal(size=(20000, 3))))
        ang = np.arcsin(sin_t) + rng.normal(size=20000) * float(rng.choice([1e-7, 1e-5, 1e-3, 0.3]))
        d64 = normalize64(np.cos(ang)[:, None] * towards[None, :] + np.sin(ang)[:, None] * perp)
        wob = normalize64(rng.normal(size=(20000, 3))) * (0.9 * dir_err * rng.random(20000))[:, None]
        d32 = (d64 + wob).astype(F)
        c32 = cte.astype(F)
        a = -((d32[:, 0] * c32[0] + d32[:, 1] * c32[1]) + d32[:, 2] * c32[2])
        delta = a * a - F(k)
        ed = F(2.0) * np.abs(a) * ea + ea * ea + F(8 * U) * (a * a + kabs)
        sure = (a > F(1.0e-6) + F(2.0) * ea) & (delta > F(2.0) * ed)
        # Sphere.intersections in FP64
        a64 = -(d64 * cte[None, :]).sum(axis=1)
        delta64 = a64 * a64 - ((cte * cte).sum() - r * r)
        hit64 = (delta64 >= 0) & ((a64 + np.sqrt(np.maximum(delta64, 0.0)) >= 1e-7))
        assert not (sure & ~hit64).any(), case
        certain += int(sure.sum())
    assert certain > 100000
