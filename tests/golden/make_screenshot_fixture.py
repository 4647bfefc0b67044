"""Cuts the rendered view out of the reference's own screenshot, RayTraceExample.png, and keeps it as a fixture.

The screenshot is the only output of the reference that exists: the application window after `startRT`
(ViewController.swift:137-195) with every input visible in the UI -- camera (0, 0, 40), direction (0, 0, -1), focal
length 1, ambient 0.2/0.2/0.2 -- showing the hard-coded two-sphere scene of ViewController.swift:155-167 under the
default lights (:59-64).  The 1000x1000 bitmap is drawn stretched into the window's image view (a layer's default
contents gravity), which measures 1340 x 1454 device pixels in the screenshot.

The fixture holds the pixels of that view (cropped to the box that contains everything that is not black, with the
box's position) as plain arrays; tests/test_reference_screenshot.py compares the oracle's and the GPU's frame of the
same scene with it.  Needs /root/reference (build container only):
    python tests/golden/make_screenshot_fixture.py
"""
import os

import numpy as np
from PIL import Image

SRC = "/root/reference/RayTraceExample.png"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_screenshot_view.npz")
UI_GREY = 34  # the window background around the image view


def main():
    im = np.array(Image.open(SRC))
    assert im.shape == (1616, 2044, 4) and (im[..., 3] == 255).all()
    rgb = im[..., :3]
    # the image view: the black rectangle bounded by the window background on all four sides
    x0, x1, y0, y1 = 666, 2006, 80, 1534
    def is_ui(a):
        return (np.abs(a.astype(int) - UI_GREY) <= 2).all()

    assert is_ui(rgb[400, x0 - 5:x0]) and is_ui(rgb[400, x1:x1 + 5])
    assert is_ui(rgb[y0 - 5:y0, 1000]) and (rgb[y1:y1 + 2, 1000] != 0).all()
    assert (rgb[400, x0:x0 + 5] == 0).all() and (rgb[400, x1 - 5:x1] == 0).all()
    assert (rgb[y0:y0 + 5, 1000] == 0).all() and (rgb[y1 - 5:y1, 1000] == 0).all()
    view = rgb[y0:y1, x0:x1]
    ys, xs = np.nonzero(view.sum(-1) > 0)
    pad = 16
    cy0, cy1 = ys.min() - pad, ys.max() + 1 + pad
    cx0, cx1 = xs.min() - pad, xs.max() + 1 + pad
    crop = view[cy0:cy1, cx0:cx1].copy()
    np.savez_compressed(OUT, crop=crop, crop_origin=np.array([cx0, cy0]), view_size=np.array([x1 - x0, y1 - y0]),
                        view_origin_in_png=np.array([x0, y0]), bitmap_size=np.array([1000, 1000]))
    print("view", view.shape, "crop", crop.shape, "at", (cx0, cy0), os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
