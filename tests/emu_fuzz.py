"""Random scenes through the CPU-emulated HIP kernels (tests/emu), each compared with the oracle: test infrastructure.

    python tests/emu_fuzz.py --cases 200 --seed 1 --jobs 6

Every case is reproducible from (seed, index): `--only INDEX` reruns one.  The generator aims at the corners of
the device code rather than at pretty pictures: group sizes around the tree's fan-out, empty groups, degenerate
and non-finite triangles, cameras inside the geometry, bounding spheres that do not bound, odd frame sizes,
windows, tiles, all sample counts."""
import argparse
import copy
import os
import sys
import tempfile
import traceback
from concurrent.futures import ThreadPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import emu_util  # noqa: E402
from raytracer_amd import _abi, scenes  # noqa: E402

GROUP_SIZES = (0, 1, 2, 7, 8, 9, 15, 16, 17, 63, 64, 65, 100, 260, 520)
SUBDIVISIONS = (1.0, 1.0, 0.5, 0.5, 0.25, 0.4, 0.3, 0.7, 0.15)


def random_triangles(rng, n, centre, spread, scale):
    t = np.zeros(n, scenes.TRI_DTYPE)
    if n == 0:
        return t
    c = np.asarray(centre) + rng.uniform(-spread, spread, size=(n, 1, 3))
    t["v"] = c + rng.normal(size=(n, 3, 3)) * scale
    style = rng.integers(0, 12, size=n)
    for i in np.flatnonzero(style == 0):  # degenerate: two equal vertices, or all three on a line
        t["v"][i, 1] = t["v"][i, 0] if rng.random() < 0.5 else 2 * t["v"][i, 2] - t["v"][i, 0]
    for i in np.flatnonzero(style == 1):  # very large
        t["v"][i] *= 40.0
    for i in np.flatnonzero(style == 2):  # very small
        t["v"][i] = t["v"][i, 0] + (t["v"][i] - t["v"][i, 0]) * 1e-6
    if rng.random() < 0.25:  # a few non-finite coordinates
        for i in rng.integers(0, n, size=max(1, n // 40)):
            t["v"][i, rng.integers(0, 3), rng.integers(0, 3)] = rng.choice([np.nan, np.inf, -np.inf])
    return t


def random_case(seed, index):
    rng = np.random.default_rng([seed, index])
    n_mats = int(rng.integers(1, 4))
    mats = np.concatenate([scenes.material(Ns=float(rng.choice([0.0, 1.0, 3.5, 8.0, 40.0])), d=1.0, illum=2,
                                           Kd=tuple(rng.uniform(0, 1, 3)), Ka=tuple(rng.uniform(0, 1, 3)), Ks=tuple(rng.uniform(0, 1, 3)))
                           for _ in range(n_mats)])
    n_lights = int(rng.choice([0, 1, 1, 2, 3, 3]))
    lights = [scenes.light(loc=tuple(rng.normal(size=3) * rng.choice([1.0, 5.0, 50.0])), specular=tuple(rng.uniform(0, 1, 3)),
                           diffuse=tuple(rng.uniform(0, 1, 3))) for _ in range(n_lights)]
    lights = np.concatenate(lights) if lights else np.zeros(0, scenes.LIGHT_DTYPE)

    tris, objects = [], []
    n_tris = 0
    for _ in range(int(rng.integers(1, 7))):
        kind = rng.choice(["sphere", "bounding", "bounding", "run", "run", "light"])
        if kind == "sphere":
            objects.append(scenes.sphere_object(tuple(rng.normal(size=3) * 0.8), float(rng.uniform(0.05, 0.9)), int(rng.integers(-1, n_mats))))
        elif kind == "light":
            if n_lights:
                objects.append(scenes.light_object(int(rng.integers(0, n_lights))))
        else:
            n = int(rng.choice(GROUP_SIZES))
            centre = rng.normal(size=3) * 0.7
            t = random_triangles(rng, n, centre, float(rng.uniform(0.1, 1.0)), float(rng.choice([0.02, 0.1, 0.3, 1.0])))
            t["material"] = rng.integers(-1, n_mats, size=n)
            tris.append(t)
            if kind == "run":
                objects.append(scenes.triangles_object(n_tris, n))
            else:
                # a sphere that may or may not bound its members (the reference's own one does not always)
                objects.append(scenes.sphere_object(tuple(centre + rng.normal(size=3) * 0.1), float(rng.uniform(0.2, 2.5)),
                                                    int(rng.integers(-1, n_mats)), bounding=True, first=n_tris, count=n))
            n_tris += n
    tris = np.concatenate(tris) if tris else np.zeros(0, scenes.TRI_DTYPE)
    if not objects:
        objects.append(scenes.sphere_object((0, 0, 0), 0.5, 0))
    data = scenes.SceneData(lights=lights, materials=mats, objects=np.concatenate(objects), triangles=tris)

    pos = rng.normal(size=3)
    pos = pos / np.linalg.norm(pos) * rng.choice([0.05, 0.6, 2.0, 4.0, 6.0])  # inside, near, outside
    look = -pos + rng.normal(size=3) * 0.15 * np.linalg.norm(pos)
    if rng.random() < 0.15:
        look = rng.normal(size=3)
    look = look * rng.choice([0.1, 1.0, 3.0])  # the direction is used as typed, not normalised
    cam = scenes.make_camera(tuple(pos), tuple(look), float(rng.choice([1.0, 2.0, 5.0, 10.0])))

    W, H = int(rng.integers(2, 41)), int(rng.integers(2, 41))
    sub = float(rng.choice(SUBDIVISIONS))
    window = None
    if rng.random() < 0.3 and W > 3 and H > 3:
        x0, y0 = int(rng.integers(0, W - 2)), int(rng.integers(0, H - 2))
        window = (x0, y0, int(rng.integers(x0 + 1, W)), int(rng.integers(y0 + 1, H)))
    tile = dict()
    if rng.random() < 0.3:
        world = int(rng.integers(2, 5))
        tile = dict(tile_size=int(rng.choice([4, 8, 32])), tile_rank=int(rng.integers(0, world)), tile_world=world)
    params = scenes.make_params(W, H, sub, tuple(rng.uniform(0, 0.5, 3)), window=window, **tile)
    return data, cam, params


def bounding_spheres_only(data, seed, index):
    """The scene reduced to what the path without a ray-generation launch serves: only its bounding spheres stay in
    the list (without materials, up to eight), every specular exponent becomes a small whole number.  None if the scene
    has no bounding sphere."""
    rng = np.random.default_rng([seed, index, 7])
    objs = data.objects[data.objects["kind"] == _abi.RTMI_BOUNDING_SPHERE][:8].copy()
    if not len(objs):
        return None
    objs["material"] = -1
    mats = data.materials.copy()
    mats["Ns"] = rng.choice([1.0, 2.0, 8.0, 8.0, 40.0, 64.0], size=len(mats))
    return scenes.SceneData(lights=data.lights, materials=mats, objects=objs, triangles=data.triangles)


def run_one(seed, index, oracle, keep_dir=None, tsan=False):
    from test_emulated_kernels import SEARCHES, check_frames
    data, cam, params = random_case(seed, index)
    # every third case under the nearest-hit extension (cone tree only, no list-scan fallback after an overflow)
    nearest = index % 3 == 2 and index % 5 != 4
    if nearest:
        params.hit_policy = _abi.RTMI_HIT_NEAREST
    if index % 4 == 1:  # every fourth case with the secondary-ray extensions: distance limit, shadow / reflection rays entering bounding spheres
        params.flags = [2, 4, 6, 8, 12, 14][(index // 4) % 6]
    # every sixth case asks for pixels only on a scene of bounding spheres: the path without a ray-generation launch (same
    # bitmap and hit ids as the reference's; the ray counters differ, so they are not compared)
    pixels_only = index % 6 == 0 and not params.flags and not nearest
    if pixels_only:
        reduced = bounding_spheres_only(data, seed, index)
        pixels_only = reduced is not None
        if pixels_only:
            data = reduced
    expected = [oracle.render(data, cam, params, threads=1)] * 2
    env = {"EMU_SEED": str(index % 4), "EMU_FILL": str([0xA5, 0, 0xFF, 0x7F][index % 4]), "EMU_CUS": str(1 + index % 3)}
    if index % 5 == 4:  # candidate lists that overflow, frames cut into batches
        env["RTMI_CAND_CAPACITY"] = str([64, 640, 6400][(index // 5) % 3])
    if index % 7 == 6:
        env["RTMI_MAX_BATCH_SAMPLES"] = "1024"
    if index % 9 == 6:  # room for two live 8x8 blocks to start with: frames that need more are rendered again
        env["RTMI_LIVE_CAPACITY"] = "2"
    if index % 11 == 3:  # the list of candidates of rays that have several overflows: the decisions scan every candidate
        env["RTMI_MULTI_CAPACITY"] = "64"
    if pixels_only:
        params.flags = _abi.RTMI_FLAG_NO_RAY_STATISTICS
        expected = [e[:2] + (None,) for e in expected]
        if index % 12 == 0:
            env["RTMI_FAST_FORCE"] = str(1 + (index // 12) % 3)
        # start records that name triangles although the mesh is fine for the frame (small frames always are), or never
        env["RTMI_LEAF_MASKS"] = "2" if index % 2 == 0 else ("0" if index % 8 == 1 else "1")
        if index % 2 == 0:
            env["RTMI_SEARCH_R"] = "4"  # (the walk with four rays per lane is the one that reads them)
    for name, search in SEARCHES.items():
        if nearest and name != "tree":
            continue
        p = copy.copy(params)
        p.search = search
        with tempfile.TemporaryDirectory(dir=keep_dir) as d:
            if tsan:
                e = dict(env, TSAN_OPTIONS="halt_on_error=0 exitcode=0 report_signal_unsafe=0")
                frames, stats, _, stderr = emu_util.run_emulated(data, [cam, cam], p, d, env=e, timeout=1800, binary=emu_util.TSAN_BIN,
                                                                 with_stderr=True)
                races = emu_util.race_reports(stderr)
                assert not races, races[:3]
            else:
                frames, stats, _ = emu_util.run_emulated(data, [cam, cam], p, d, env=env, timeout=600)
            check_frames(frames, stats, expected, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=50)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--jobs", type=int, default=4)
    ap.add_argument("--only", type=int, default=None)
    ap.add_argument("--tsan", action="store_true", help="run the ThreadSanitizer build and fail on any race report (make -C tests/emu tsan)")
    a = ap.parse_args()
    import oracle
    oracle.build()
    emu_util.build()
    if a.tsan:
        import subprocess
        subprocess.check_call(["make", "-s", "-C", emu_util.EMU_DIR, "tsan"])
    todo = [a.only] if a.only is not None else list(range(a.cases))
    failed = []

    def job(i):
        try:
            run_one(a.seed, i, oracle, tsan=a.tsan)
            return i, None
        except BaseException:  # noqa: BLE001
            return i, traceback.format_exc()

    with ThreadPoolExecutor(a.jobs) as pool:
        for i, err in pool.map(job, todo):
            if err:
                failed.append(i)
                print(f"case {i} FAILED\n{err[-3000:]}", flush=True)
            elif i % 10 == 0:
                print(f"case {i} ok", flush=True)
    print(f"{len(todo) - len(failed)} of {len(todo)} cases passed; failed: {failed}")
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
