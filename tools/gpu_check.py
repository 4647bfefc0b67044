"""Quick GPU-vs-oracle parity sweep (development aid; the graded tests live in tests/)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
import raytracer_amd as rt  # noqa: E402
from raytracer_amd import scenes  # noqa: E402


def compare(name, data, cam, params, threads=16):
    t0 = time.time()
    o_rgba, o_hit, o_cnt = oracle.render(data, cam, params, threads=threads)
    t1 = time.time()
    sc = rt.Scene(data)
    g_rgba, g_hit, g_cnt = sc.render(cam, params)
    t2 = time.time()
    g_rgba2, g_hit2, g_cnt2 = sc.render(cam, params)
    sc.close()
    hit_mis = int((o_hit != g_hit).sum())
    rgba_diff = np.abs(o_rgba.astype(int) - g_rgba.astype(int))
    rep = int((g_rgba != g_rgba2).sum()) + int((g_hit != g_hit2).sum())
    print(f"{name}: samples {o_cnt['samples']} hits {o_cnt['primary_hits']} | hit-id mismatches {hit_mis} | "
          f"rgba max diff {rgba_diff.max()} n_diff {int((rgba_diff > 0).sum())} | repeat diff {rep} | "
          f"oracle {t1 - t0:.2f}s gpu {t2 - t1:.2f}s | gpu ms_total {g_cnt2['ms_total']:.2f} search {g_cnt2['ms_tri_search']:.2f} "
          f"tests {g_cnt2['tri_tests_primary_kernel']} exact {g_cnt2['tri_exact_tests']}", flush=True)
    for key in ("samples", "rays_reflection", "rays_shadow", "primary_hits", "sphere_tests"):
        if o_cnt[key] != g_cnt[key]:
            print(f"   counter {key}: oracle {o_cnt[key]} gpu {g_cnt[key]}")
    return hit_mis, int(rgba_diff.max())


def main():
    bad = 0
    cam = scenes.make_camera(**scenes.VC_CAMERA)
    h, d = compare("two_sphere 400x300", scenes.two_sphere_scene(), cam, scenes.make_params(400, 300))
    bad += h + d
    h, d = compare("single_sphere 640x480", scenes.single_sphere_scene(), cam, scenes.make_params(640, 480))
    bad += h + d
    mcam = scenes.make_camera(**scenes.MESH_CAMERA)
    for level, form, nl, sub, wh in [(2, "B", 1, 1.0, (200, 160)), (3, "B", 3, 0.5, (160, 160)),
                                     (2, "T", 2, 1.0, (96, 96)), (4, "B", 3, 0.25, (128, 128)),
                                     (4, "B", 1, 0.5, (512, 512))]:
        data = scenes.mesh_scene(level, form, nl, bounding_sphere=oracle.bounding_sphere)
        h, d = compare(f"icosphere L{level} form {form} {nl}L sub {sub} {wh}", data, mcam,
                       scenes.make_params(wh[0], wh[1], sub))
        bad += h + d
    # shuffled order: list-order semantics differ from nearest
    data = scenes.mesh_scene(3, "B", 1, shuffle_seed=0, bounding_sphere=oracle.bounding_sphere)
    h, d = compare("icosphere L3 shuffled", data, mcam, scenes.make_params(200, 200, 0.5))
    bad += h + d
    # off-axis camera
    cam2 = scenes.make_camera((1.5, 0.7, 3.0), (-0.4, -0.2, -1.0), 6.0)
    h, d = compare("icosphere L3 off-axis", scenes.mesh_scene(3, "B", 2, bounding_sphere=oracle.bounding_sphere),
                   cam2, scenes.make_params(240, 200, 0.5))
    bad += h + d
    print("TOTAL BAD", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
