"""Loading of the fixtures under tests/golden/ (plain numeric arrays only)."""
import glob
import os

import numpy as np

from raytracer_amd import scenes

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
COUNTER_KEYS = ("samples", "rays_reflection", "rays_shadow", "tri_tests", "sphere_tests", "primary_hits")


# fixtures of another shape: the view cut out of the reference's screenshot (tests/test_reference_screenshot.py)
NOT_ORACLE_FRAMES = ("reference_screenshot_view",)


def golden_names():
    names = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    return [n for n in names if n not in NOT_ORACLE_FRAMES]


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    data = scenes.SceneData.from_arrays({k: z[k] for k in ("lights", "materials", "objects", "triangles")}, name)
    c = z["cam"]
    cam = scenes.make_camera(c[0:3], c[3:6], c[6])
    p = z["params"]
    policy = int(z["hit_policy"][0]) if "hit_policy" in z.files else 0  # 1: the nearest-hit extension
    flags = int(z["flags"][0]) if "flags" in z.files else 0  # the shadow-ray extensions (RTMI_FLAG_SHADOW_*)
    params = scenes.make_params(int(p[0]), int(p[1]), float(p[2]), tuple(p[3:6]), hit_policy=policy, flags=flags)
    counters = dict(zip(COUNTER_KEYS, (int(x) for x in z["counters"])))
    return data, cam, params, z["rgba"], z["hit"], counters
