"""The oracle and the GPU path against the one output the reference itself holds: the rendered view inside its
screenshot RayTraceExample.png (fixture: tests/golden/reference_screenshot_view.npz, cut out by
tests/golden/make_screenshot_fixture.py).

The screenshot shows `startRT`'s hard-coded scene (two spheres of radius 25 at x = +-25, material Ns 8 / Kd .5 /
Ka .7 / Ks 1, ViewController.swift:155-167) under the default lights (:59-64), camera (0, 0, 40) looking down -z,
focal length 1, ambient 0.2 -- all visible in the UI -- as a 1000x1000 bitmap stretched into a 1340x1454-pixel view.
A frame of the same scene is stretched the same way (bilinear) and compared pixel by pixel.

What this pins: the camera model (10x10 window, un-normalised direction, (W-1)x(H-1) scaling), the sphere test, the
whole of calculateLighting for sphere hits (ambient, diffuse, Phong specular with Ns = 8, the reflected image of each
sphere in the other, the shadow each casts on the other), list order between the two spheres, clamping and the
truncating x255 quantisation: a frame that is ONE grey level brighter wherever it is lit already fails the bounds
below (exact matches on smooth areas drop from 85 % to 41 %, the mean difference rises from 0.17 to 1.2).
What it cannot pin: anything below 1/255 (e.g. whether `normalize` divides or multiplies by a reciprocal square root),
triangles, bounding spheres, supersampling, lights used as objects.
Tolerances absorb the unknown resampling filter of the window server along edges (silhouettes, highlight rims).
"""
import os

import numpy as np
import pytest
from PIL import Image

from raytracer_amd import scenes

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_screenshot_view.npz")


def compare_with_screenshot(rgba):
    """rgba: [1000, 1000, 4] uint8 frame of the two-sphere scene.  Asserts the bounds described above."""
    f = np.load(FIXTURE)
    crop = f["crop"].astype(int)
    cx0, cy0 = (int(v) for v in f["crop_origin"])
    vw, vh = (int(v) for v in f["view_size"])
    assert tuple(f["bitmap_size"]) == rgba.shape[1::-1] == (1000, 1000)
    assert (rgba[..., 3] == 255)[:999, :999].all()
    up = np.array(Image.fromarray(np.ascontiguousarray(rgba[..., :3])).resize((vw, vh), Image.BILINEAR)).astype(int)
    h, w = crop.shape[:2]
    inside = up[cy0:cy0 + h, cx0:cx0 + w]
    outside = up.copy()
    outside[cy0:cy0 + h, cx0:cx0 + w] = 0
    assert outside.sum() == 0, "the frame is lit where the screenshot's view is black"

    # silhouettes: the lit bounding box agrees to a pixel of the 1340x1454 view
    for ours, theirs in zip(np.nonzero(inside.sum(-1) > 0), np.nonzero(crop.sum(-1) > 0)):
        assert abs(int(ours.min()) - int(theirs.min())) <= 1 and abs(int(ours.max()) - int(theirs.max())) <= 1

    d = np.abs(inside - crop).max(-1)
    assert d.mean() <= 1.0
    # smooth areas (no grey-level step above 2 in the 3x3 neighbourhood): the grey levels themselves
    g0 = inside[..., 0]
    grad = np.maximum(np.abs(np.diff(g0, axis=0, prepend=g0[:1])), np.abs(np.diff(g0, axis=1, prepend=g0[:, :1])))
    near = np.maximum.reduce([np.roll(np.roll(grad, i, 0), j, 1) for i in (-1, 0, 1) for j in (-1, 0, 1)])
    smooth = near <= 2
    assert smooth.mean() > 0.8
    assert d[smooth].max() <= 3
    assert d[smooth].mean() <= 0.25
    assert (d[smooth] == 0).mean() >= 0.80

    # the two saturated specular blobs and the bright mirror images next to the contact point
    for lo, hi in ((0, w // 2), (w // 2, w)):
        a = np.argwhere(inside[:, lo:hi, 0] >= 250)
        b = np.argwhere(crop[:, lo:hi, 0] >= 250)
        assert len(a) > 200 and abs(len(a) - len(b)) <= 0.05 * len(b)
        assert np.abs(a.mean(0) - b.mean(0)).max() <= 1.0
    # grey levels along the centre row of the view, away from steps
    row = vh // 2 - cy0
    ok = smooth[row]
    assert ok.sum() > 300 and np.abs(inside[row, ok, 0] - crop[row, ok, 0]).max() <= 2


def two_sphere_frame(render):
    return render(scenes.two_sphere_scene(), scenes.make_camera(**scenes.VC_CAMERA), scenes.make_params(1000, 1000))


def test_oracle_frame_matches_the_reference_screenshot(oracle_mod):
    compare_with_screenshot(two_sphere_frame(lambda d, c, p: oracle_mod.render(d, c, p, want_hit_ids=False, threads=8)[0]))


def test_comparison_rejects_a_frame_one_grey_level_off(oracle_mod):
    rgba = two_sphere_frame(lambda d, c, p: oracle_mod.render(d, c, p, want_hit_ids=False, threads=8)[0]).copy()
    lit = rgba[..., :3] > 0
    rgba[..., :3] = np.where(lit, np.minimum(rgba[..., :3].astype(int) + 1, 255), 0).astype(np.uint8)
    with pytest.raises(AssertionError):
        compare_with_screenshot(rgba)


@pytest.mark.gpu
def test_gpu_frame_matches_the_reference_screenshot(rt):
    def render(d, c, p):
        sc = rt.Scene(d)
        rgba, _, _ = sc.render(c, p, want_hit_ids=False)
        sc.close()
        return rgba
    compare_with_screenshot(two_sphere_frame(render))
