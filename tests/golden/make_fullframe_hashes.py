"""Full-frame digests of the BASELINE configurations, computed with the CPU oracle.

The small .npz fixtures in this directory hold whole frames; the headline configurations are far too
large for that (a 4K/16-spp frame has 132.6 M samples), so what is kept here is, per frame, the SHA-256
of the oracle's RGBA8 bitmap and of its per-sample hit-id array, a CRC-32 per band of 64 pixel rows
(to localise a difference), and the oracle's ray counters.  tests/test_fullframe_hashes.py renders the
same frames on the GPU and compares digests, so every sample of every frame is checked.

Scenes are regenerated from raytracer_amd.scenes (deterministic numpy arithmetic); the digest of the
scene's own arrays is stored too, so a libm that perturbs the inputs shows up as exactly that.

Run from the repo root (hours of CPU time for all frames; finished frames are kept):
    python tests/golden/make_fullframe_hashes.py [--threads 6] [--only C2,C3,...]
"""
import argparse
import hashlib
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from raytracer_amd import scenes  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fullframe_hashes.json")
BAND = 64

# name -> (width, height, subdivision, mesh level, lights, form, camera: "mesh" or orbit step)
FRAMES = {
    "C2": (1024, 1024, 0.5, 4, 1, "B", "mesh"),
    "C3": (3840, 2160, 0.25, 4, 3, "B", "mesh"),
    "C4": (1920, 1080, 0.5, 8, 1, "B", "mesh"),
}
# config 5: the orbiting camera at the C3 frame size -- eight consecutive steps and three more round the circle
for _step in (0, 1, 2, 3, 4, 5, 6, 7, 16, 32, 48):
    FRAMES[f"C5_step{_step:02d}"] = (3840, 2160, 0.25, 4, 3, "B", _step)
# form T (triangles listed at top level, so reflection and shadow rays do triangle work) at a size the
# oracle finishes: every secondary ray scans the list
FRAMES["T_640x360"] = (640, 360, 0.5, 3, 3, "T", "mesh")
# the secondary-ray extensions (not the reference's behaviour): the C2 frame size with three lights, a material on the
# mesh's triangles (so that its shadows and mirror images show) and all three flags
EXTENSION_FRAMES = {"C2_extensions": dict(flags=14, triangle_material=0)}
FRAMES["C2_extensions"] = (1024, 1024, 0.5, 4, 3, "B", "mesh")


def camera_kw(cam):
    return scenes.MESH_CAMERA if cam == "mesh" else scenes.orbit_camera(int(cam))


def digest(rgba, hit):
    H = rgba.shape[0]
    return {
        "rgba_sha256": hashlib.sha256(np.ascontiguousarray(rgba).tobytes()).hexdigest(),
        "hit_sha256": hashlib.sha256(np.ascontiguousarray(hit).tobytes()).hexdigest(),
        "rgba_band_crc32": [zlib.crc32(np.ascontiguousarray(rgba[y:y + BAND]).tobytes()) for y in range(0, H, BAND)],
        "hit_band_crc32": [zlib.crc32(np.ascontiguousarray(hit[y:y + BAND]).tobytes()) for y in range(0, H, BAND)],
    }


def scene_digest(data):
    h = hashlib.sha256()
    for a in (data.lights, data.materials, data.objects, data.triangles):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=6)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    done = json.load(open(OUT)) if os.path.exists(OUT) else {}
    names = [n for n in FRAMES if not args.only or n in args.only.split(",")]
    for name in names:
        if name in done:
            continue
        W, H, sub, level, n_lights, form, cam = FRAMES[name]
        data = scenes.mesh_scene(level, form, n_lights, bounding_sphere=oracle.bounding_sphere)
        ext = EXTENSION_FRAMES.get(name, {})
        if "triangle_material" in ext:
            data.triangles["material"] = ext["triangle_material"]
        params = scenes.make_params(W, H, sub, flags=ext.get("flags", 0))
        t0 = time.time()
        rgba, hit, cnt = oracle.render(data, scenes.make_camera(**camera_kw(cam)), params, threads=args.threads)
        rec = {"width": W, "height": H, "subdivision": sub, "mesh_level": level, "lights": n_lights, "form": form,
               "camera": cam, "band_rows": BAND, "scene_sha256": scene_digest(data), **ext,
               "counters": {k: int(cnt[k]) for k in ("samples", "rays_reflection", "rays_shadow", "tri_tests",
                                                     "sphere_tests", "primary_hits")},
               "oracle_seconds": round(time.time() - t0, 1), "oracle_threads": args.threads}
        rec.update(digest(rgba, hit))
        done[name] = rec
        with open(OUT + ".tmp", "w") as f:
            json.dump(done, f, indent=1, sort_keys=True)
        os.replace(OUT + ".tmp", OUT)
        print(name, rec["oracle_seconds"], "s", rec["counters"], flush=True)


if __name__ == "__main__":
    main()
