"""The scene fuzzer's cases (tests/emu_fuzz.py) rendered by the product library on the GPU, each compared with the
oracle: test infrastructure, run by hand on a GPU box.

    python tests/gpu_fuzz.py --cases 300 --seed 1 [--scale 4]

One process, one scene at a time.  --scale multiplies the frame size of every case (the generator draws frames of at
most 40x40 pixels for the emulator's sake)."""
import argparse
import copy
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from emu_fuzz import bounding_spheres_only, random_case  # noqa: E402
from raytracer_amd import _abi, scenes  # noqa: E402

COUNTERS = ("samples", "rays_reflection", "rays_shadow", "sphere_tests", "primary_hits")


def mesh_case(seed, index):
    """Closed meshes listed at top level (their bounding-sphere tree gets usable cones of normals, which random
    triangle soups never do): one to three displaced icospheres, some turned inside out, some overlapping, with
    a sphere, a light and a bounded soup from the random case of the same index around them."""
    data, cam, params = random_case(seed, index)
    rng = np.random.default_rng([seed, index, 7])
    tris, objects = [], []
    n = 0
    for _ in range(int(rng.integers(1, 4))):
        v, f = scenes.icosphere(int(rng.integers(0, 4)))
        theta = np.arccos(np.clip(v[:, 2], -1.0, 1.0))
        phi = np.arctan2(v[:, 1], v[:, 0])
        v = v * (1.0 + rng.uniform(0, 0.4) * np.sin(rng.integers(1, 5) * theta) * np.cos(rng.integers(1, 4) * phi))[:, None]
        rot, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        v = v @ rot.T * rng.uniform(0.2, 1.2) + rng.normal(size=3) * 0.6
        if rng.random() < 0.3:
            f = f[:, ::-1]  # inside out
        if rng.random() < 0.3:
            f = f[rng.permutation(len(f))[: max(1, int(len(f) * rng.uniform(0.3, 1.0)))]]  # an open shell, list order shuffled
        t = scenes.triangles_from_vertices(v, f, 0)
        t["material"] = rng.integers(-1, len(data.materials), size=len(t))
        tris.append(t)
        objects.append(scenes.triangles_object(n, len(t)))
        n += len(t)
        if rng.random() < 0.4:
            objects.append(scenes.sphere_object(tuple(rng.normal(size=3)), float(rng.uniform(0.1, 0.6)), int(rng.integers(-1, len(data.materials)))))
        if rng.random() < 0.3 and len(data.lights):
            objects.append(scenes.light_object(int(rng.integers(0, len(data.lights)))))
    # the bounded groups of the random case, behind the meshes in the list
    for ob in data.objects:
        if int(ob["kind"]) == _abi.RTMI_BOUNDING_SPHERE and int(ob["count"]) > 0 and rng.random() < 0.5:
            o2 = ob.copy()
            tris.append(data.triangles[int(ob["first"]): int(ob["first"]) + int(ob["count"])])
            o2["first"] = n
            objects.append(np.array([o2], dtype=data.objects.dtype))
            n += int(ob["count"])
    order = rng.permutation(len(objects))
    data = scenes.SceneData(lights=data.lights, materials=data.materials, objects=np.concatenate([objects[k] for k in order]),
                            triangles=np.concatenate(tris))
    return data, cam, params


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--scale", type=int, default=1)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--only", default=None, help="comma-separated case indices")
    ap.add_argument("--meshes", action="store_true", help="closed meshes at top level instead of the emulator fuzzer's soups")
    ap.add_argument("--extras", action="store_true", help="also: a second camera per scene, the pixels-only flag, up to 23 lights")
    ap.add_argument("--fast", action="store_true",
                    help="every scene reduced to its bounding spheres (no materials on them, whole-number exponents) and rendered with "
                         "RTMI_FLAG_NO_RAY_STATISTICS: the path without a ray-generation launch, three frames per scene (two cameras)")
    a = ap.parse_args(argv)
    import oracle
    import raytracer_amd as rt
    oracle.build()
    failed, skipped = [], []
    t0 = time.time()
    todo = [int(x) for x in a.only.split(",")] if a.only else list(range(a.cases))
    for i in todo:
        data, cam, params = mesh_case(a.seed, i) if a.meshes else random_case(a.seed, i)
        if a.scale > 1:
            w = (params.x0, params.y0, params.x1, params.y1)
            window = tuple(v * a.scale for v in w) if params.x1 > params.x0 else None
            tile = dict(tile_size=params.tile_size, tile_rank=params.tile_rank, tile_world=params.tile_world) if params.tile_world > 1 else {}
            params = scenes.make_params(params.width * a.scale, params.height * a.scale, params.pixel_subdivision, tuple(params.ambient),
                                        window=window, **tile)
        if a.fast:
            reduced = bounding_spheres_only(data, a.seed, i)
            if reduced is None:
                skipped.append(i)
                continue
            _, cam2, _ = random_case(a.seed + 1000, i)
            want = [oracle.render(reduced, c, params, threads=a.threads) for c in (cam, cam2, cam)]
            p = copy.copy(params)
            p.flags = _abi.RTMI_FLAG_NO_RAY_STATISTICS
            sc = rt.Scene(reduced)
            try:
                for k, (c, (o_rgba, o_hit, o_cnt)) in enumerate(zip((cam, cam2, cam), want)):
                    g_rgba, g_hit, g_cnt = sc.render(c, p)
                    if not ((g_hit == o_hit).all() and (g_rgba == o_rgba).all() and g_cnt["primary_hits"] == o_cnt["primary_hits"]):
                        failed.append(i)
                        print(f"case {i} frame {k} FAILED: {int((g_hit != o_hit).sum())} hit ids, {int((g_rgba != o_rgba).sum())} bytes differ", flush=True)
            finally:
                sc.close()
            continue
        nearest = i % 3 == 2
        if nearest:
            params.hit_policy = _abi.RTMI_HIT_NEAREST
        if i % 4 == 1:  # every fourth case with the secondary-ray extensions: distance limit, shadow / reflection rays entering bounding spheres
            params.flags = [2, 4, 6, 8, 12, 14][(i // 4) % 6]
        if a.extras and i % 4 == 3:  # many lights (more than 16: the shading kernel that traces its own rays)
            rng = np.random.default_rng([a.seed, i, 99])
            n = int(rng.choice([5, 9, 16, 17, 20]))
            extra = np.concatenate([scenes.light(loc=tuple(rng.normal(size=3) * rng.choice([2.0, 6.0])), specular=tuple(rng.uniform(0, 0.3, 3)),
                                                 diffuse=tuple(rng.uniform(0, 0.3, 3))) for _ in range(n)])
            data = scenes.SceneData(lights=np.concatenate([data.lights, extra]), materials=data.materials, objects=data.objects,
                                    triangles=data.triangles)
        o_rgba, o_hit, o_cnt = oracle.render(data, cam, params, threads=a.threads)
        sc = rt.Scene(data)
        try:
            if a.extras:
                # another camera on the same scene object first (the operands of this one are then packed afresh), and
                # the pixels-only flag: same bitmap and hit ids, whatever it does to the ray counters
                _, cam2, _ = random_case(a.seed + 1000, i)
                p2 = copy.copy(params)
                o2_rgba, o2_hit, _ = oracle.render(data, cam2, p2, threads=a.threads)
                p3 = copy.copy(params)
                p3.flags |= _abi.RTMI_FLAG_NO_RAY_STATISTICS
                try:
                    g2_rgba, g2_hit, _ = sc.render(cam2, p2)
                    g3_rgba, g3_hit, _ = sc.render(cam, p3)
                except rt.RtmiError as e:
                    if not (nearest and "RTMI_HIT_NEAREST has no list-scan fallback" in str(e)):
                        raise
                    skipped.append(i)
                else:
                    if not ((g2_hit == o2_hit).all() and (g2_rgba == o2_rgba).all()):
                        failed.append(i)
                        print(f"case {i} second camera FAILED", flush=True)
                    if not ((g3_hit == o_hit).all() and (g3_rgba == o_rgba).all()):
                        failed.append(i)
                        print(f"case {i} pixels only FAILED", flush=True)
            for search in (_abi.RTMI_SEARCH_CONE_TREE, _abi.RTMI_SEARCH_LIST_SCAN):
                if nearest and search == _abi.RTMI_SEARCH_LIST_SCAN:
                    continue
                p = copy.copy(params)
                p.search = search
                for rep in range(2):  # the second frame reuses the camera's preparation
                    try:
                        g_rgba, g_hit, g_cnt = sc.render(cam, p)
                    except rt.RtmiError as e:
                        # the documented limit of the extension: no list-scan fallback when the candidate lists overflow
                        if nearest and "RTMI_HIT_NEAREST has no list-scan fallback" in str(e):
                            skipped.append(i)
                            break
                        raise
                    bad = []
                    if not (g_hit == o_hit).all():
                        bad.append(f"{int((g_hit != o_hit).sum())} hit ids")
                    if not (g_rgba == o_rgba).all():
                        bad.append(f"{int((g_rgba != o_rgba).any(axis=-1).sum())} pixels")
                    bad += [f"{k}: {g_cnt[k]} != {o_cnt[k]}" for k in COUNTERS if g_cnt[k] != o_cnt[k]]
                    if bad:
                        failed.append(i)
                        print(f"case {i} search {search} frame {rep} FAILED: {'; '.join(bad)}", flush=True)
        finally:
            sc.close()
        if i % 25 == 0:
            print(f"case {i} done, {time.time() - t0:.0f} s", flush=True)
    failed = sorted(set(failed))
    print(f"{len(todo) - len(failed)} of {len(todo)} cases passed ({len(set(skipped))} of them refused as documented); failed: {failed}")
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
