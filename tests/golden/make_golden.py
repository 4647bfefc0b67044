"""Generates the golden fixtures in this directory from the CPU oracle.

The reference ships no golden vectors and cannot be run here, so these pin the ORACLE's
output (scene arrays, camera, parameters -> RGBA8 bitmap, per-sample hit ids, counters) at
the commit that produced them; tests/test_golden.py replays them against the oracle (CPU)
and against librtmi.so (GPU).  Scene arrays are stored in the fixture so that a different
libm cannot perturb the inputs.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from raytracer_amd import scenes  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


ONLY = sys.argv[1:]  # fixture names to (re)generate; none = all


def save(name, data, cam_kw, W, H, sub, ambient=(0.2, 0.2, 0.2), hit_policy=0, flags=0):
    if ONLY and name not in ONLY:
        return
    cam = scenes.make_camera(**cam_kw)
    params = scenes.make_params(W, H, sub, ambient, hit_policy=hit_policy, flags=flags)
    rgba, hit, cnt = oracle.render(data, cam, params)
    arrs = data.to_arrays()
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        cam=np.array(list(cam_kw["pos"]) + list(cam_kw["direction"]) + [cam_kw["focal_length"]], np.float64),
        params=np.array([W, H, sub] + list(ambient), np.float64),
        rgba=rgba, hit=hit, hit_policy=np.array([hit_policy], np.int64), flags=np.array([flags], np.int64),
        counters=np.array([cnt[k] for k in ("samples", "rays_reflection", "rays_shadow", "tri_tests",
                                            "sphere_tests", "primary_hits")], np.int64),
        **arrs)
    print(name, rgba.shape, "hits", cnt["primary_hits"], "of", cnt["samples"])


def two_triangle_list_order():
    """Two parallel front-facing triangles, the far one listed first: list order wins."""
    far = [[-1.0, -1.0, -1.0], [1.0, -1.0, -1.0], [0.0, 1.0, -1.0]]
    near = [[-1.0, -1.0, 0.0], [1.0, -1.0, 0.0], [0.0, 1.0, 0.0]]
    tris = np.zeros(2, scenes.TRI_DTYPE)
    tris["v"][0], tris["v"][1] = far, near
    tris["material"] = [0, 1]
    mats = np.concatenate([scenes.material(**scenes.VC_MATERIAL),
                           scenes.material(Ns=2.0, d=1.0, illum=2, Kd=(0.9, 0.1, 0.1), Ka=(0.2, 0.9, 0.2), Ks=(0.3, 0.3, 0.9))])
    return scenes.SceneData(lights=scenes.light(**scenes.L1), materials=mats,
                            objects=scenes.triangles_object(0, 2), triangles=tris, name="two_tri")


def mixed_scene():
    """Sphere, a light placed in the object list, a bounded mesh and top-level triangles
    interleaved; three lights; exercises reflection and shadow rays against every kind."""
    v, f = scenes.displaced_icosphere(1)
    mesh = scenes.triangles_from_vertices(v * 0.8, f, 0)
    floor = np.zeros(2, scenes.TRI_DTYPE)
    floor["v"][0] = [[-3.0, -1.2, -3.0], [3.0, -1.2, 3.0], [3.0, -1.2, -3.0]]
    floor["v"][1] = [[-3.0, -1.2, -3.0], [-3.0, -1.2, 3.0], [3.0, -1.2, 3.0]]
    floor["material"] = 1
    tris = np.concatenate([mesh, floor])
    c, r = oracle.bounding_sphere(mesh)
    objs = np.concatenate([
        scenes.sphere_object((1.6, 0.2, 0.0), 0.6, 1),
        scenes.light_object(2),
        scenes.sphere_object(c, r, -1, bounding=True, first=0, count=len(mesh)),
        scenes.triangles_object(len(mesh), 2),
        scenes.sphere_object((-1.7, -0.3, 0.5), 0.5, 0),
    ])
    mats = np.concatenate([scenes.material(**scenes.VC_MATERIAL),
                           scenes.material(Ns=20.0, d=1.0, illum=2, Kd=(0.2, 0.6, 0.8), Ka=(0.3, 0.3, 0.5), Ks=(0.9, 0.9, 0.9))])
    lights = np.concatenate([scenes.light(**scenes.L1), scenes.light(**scenes.L2),
                             scenes.light(loc=(0.0, 6.0, 4.0), specular=(0.7, 0.7, 0.7), diffuse=(0.6, 0.6, 0.6))])
    return scenes.SceneData(lights=lights, materials=mats, objects=objs, triangles=tris, name="mixed")


def main():
    save("single_sphere_64x48", scenes.single_sphere_scene(), scenes.VC_CAMERA, 64, 48, 1.0)
    save("two_sphere_64x48", scenes.two_sphere_scene(), scenes.VC_CAMERA, 64, 48, 1.0)
    save("two_tri_list_order_32x32", two_triangle_list_order(),
         dict(pos=(0.0, 0.0, 4.0), direction=(0.0, 0.0, -1.0), focal_length=10.0), 32, 32, 1.0)
    for sub, tag in ((1.0, "s1"), (0.5, "s05"), (0.25, "s025")):
        save(f"ico2_bounded_{tag}_48x40", scenes.mesh_scene(2, "B", 1, bounding_sphere=oracle.bounding_sphere),
             scenes.MESH_CAMERA, 48, 40, sub)
    save("ico2_toplevel_3L_40x40", scenes.mesh_scene(2, "T", 3), scenes.MESH_CAMERA, 40, 40, 0.5)
    save("mixed_3L_64x48", mixed_scene(), dict(pos=(0.5, 0.8, 5.0), direction=(-0.05, -0.12, -1.0), focal_length=8.0),
         64, 48, 0.5, ambient=(0.25, 0.2, 0.3))
    # the nearest-hit extension (RTMI_HIT_NEAREST), on scenes where it differs from the reference's list order:
    # a non-convex mesh whose triangles are listed in shuffled order, and the mixed scene
    save("ico2_bounded_nearest_48x40", scenes.mesh_scene(2, "B", 2, shuffle_seed=11, bounding_sphere=oracle.bounding_sphere),
         scenes.MESH_CAMERA, 48, 40, 0.5, hit_policy=1)
    save("mixed_nearest_3L_64x48", mixed_scene(), dict(pos=(0.5, 0.8, 5.0), direction=(-0.05, -0.12, -1.0), focal_length=8.0),
         64, 48, 0.5, ambient=(0.25, 0.2, 0.3), hit_policy=1)
    # the shadow-ray extensions: RTMI_FLAG_SHADOWS_ENTER_BOUNDED (4) on a non-convex bounded mesh whose triangles carry
    # a material (its folds then shadow one another), RTMI_FLAG_REFLECTIONS_ENTER_BOUNDED (8) on the same (its folds then
    # mirror one another), and both shadow flags (6) on the mixed scene
    folded = scenes.mesh_scene(2, "B", 3, bounding_sphere=oracle.bounding_sphere)
    folded.triangles["material"] = 0
    save("ico2_bounded_shadows_enter_48x40", folded, scenes.MESH_CAMERA, 48, 40, 0.5, flags=4)
    save("ico2_bounded_reflections_enter_48x40", folded, scenes.MESH_CAMERA, 48, 40, 0.5, flags=8)
    save("mixed_shadow_extensions_3L_64x48", mixed_scene(), dict(pos=(0.5, 0.8, 5.0), direction=(-0.05, -0.12, -1.0), focal_length=8.0),
         64, 48, 0.5, ambient=(0.25, 0.2, 0.3), flags=6)


if __name__ == "__main__":
    main()
