"""The HIP kernels, compiled unchanged as host C++ against a CPU stand-in for the HIP runtime (tests/emu) and
run under AddressSanitizer + UBSan: every work-item is a fiber, every "device" buffer a malloc block filled
with garbage, wave-level operations and barriers are checked for agreement.  Results are held to the same
bar as on the GPU: hit ids and RGBA8 identical to the oracle's.  This is a CPU-side check of the DEVICE code
(indexing, barriers, uninitialised reads); it says nothing about speed and does not replace the GPU tests."""
import copy
import subprocess
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import emu_util
from golden_util import golden_names, load_golden
from scene_cases import light_hit_scene, shadow_extension_cases, toplevel_mesh_with_company
from raytracer_amd import _abi, scenes, sharding

SEARCHES = {"tree": _abi.RTMI_SEARCH_CONE_TREE, "scan": _abi.RTMI_SEARCH_LIST_SCAN}
PARITY_COUNTERS = ("samples", "rays_reflection", "rays_shadow", "sphere_tests", "primary_hits")
# kernels whose wave-level votes are meant to be reached by part of a wave (the secondary-ray tree walk is
# called from divergent shading code); everywhere else a vote that misses live lanes is a defect
DIVERGENT_OK = ("k_shade<true, ", "k_shade_candidates<true, ", "k_trace_top<", "k_shade_replay<", "k_shade_ext<")


@pytest.fixture(scope="module")
def emu():
    emu_util.build()
    subprocess.check_call(["make", "-s", "-C", emu_util.EMU_DIR, "build/emu_selftest"])
    return emu_util


def rendered_mask(p):
    W, H = p.width, p.height
    m = np.zeros((H, W), bool)
    x0, y0, x1, y1 = (p.x0, p.y0, p.x1, p.y1) if (p.x0 or p.y0 or p.x1 or p.y1) else (0, 0, W - 1, H - 1)
    m[y0:y1, x0:x1] = True
    if p.tile_size > 0 and p.tile_world > 1:
        m &= sharding.owned_pixel_mask(W, H, p.tile_rank, p.tile_world, p.tile_size)
    return m


def check_frames(frames, stats, expected, params):
    mask = rendered_mask(params)
    for k, (rgba, hit, cnt) in enumerate(frames):
        o_rgba, o_hit, o_cnt = expected[k]
        assert ((hit[..., 0] != -77) == mask).all(), f"frame {k}: the set of written samples is not the rendered set"
        assert (rgba[~mask] == 0x5A).all(), f"frame {k}: bytes outside the rendered pixels were written"
        assert (hit[mask] == o_hit[mask]).all(), f"frame {k}: {int((hit[mask] != o_hit[mask]).sum())} hit-id mismatches"
        assert (rgba[mask] == o_rgba[mask]).all(), f"frame {k}: {int((rgba[mask] != o_rgba[mask]).sum())} RGBA8 byte mismatches"
        if o_cnt is not None:
            g = cnt.as_dict()
            for key in PARITY_COUNTERS:
                assert g[key] == o_cnt[key], (k, key)
    for name, st in stats.items():
        assert st["shfl_from_absent"] == 0, name
        if not any(d in name for d in DIVERGENT_OK):
            assert st["partial_wave_ops"] == 0, f"{name}: a wave-level vote or barrier was reached by part of a wave"


def run_case(emu, oracle_mod, tmp_path, data, cams, params, searches=("tree", "scan"), env=None, frames_per_scene=0):
    expected = [oracle_mod.render(data, c, params, threads=8) for c in cams]
    ps = []
    for s in searches:
        ps.append(copy.copy(params))
        ps[-1].search = SEARCHES[s]
    with ThreadPoolExecutor(len(ps)) as pool:  # the emulated renders are child processes
        jobs = [pool.submit(emu.run_emulated, data, cams, p, tmp_path / s, env=env, frames_per_scene=frames_per_scene)
                for s, p in zip(searches, ps)]
        rendered = {}
        for s, job, p in zip(searches, jobs, ps):
            frames, stats, _ = job.result()
            check_frames(frames, stats, expected, p)
            rendered[s] = frames
    return rendered


def test_emulator_selftest(emu):
    exe = emu.EMU_DIR + "/build/emu_selftest"
    r = subprocess.run([exe, "ok"], capture_output=True, text=True)
    assert r.returncode == 0 and "selftest ok" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([exe, "ok"], capture_output=True, text=True, env={"EMU_SEED": "7"})
    assert r.returncode == 0 and "selftest ok" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([exe, "oob"], capture_output=True, text=True)
    assert r.returncode != 0 and "heap-buffer-overflow" in r.stderr and "k_oob" in r.stderr
    r = subprocess.run([exe, "lds"], capture_output=True, text=True)
    assert r.returncode != 0 and ("global-buffer-overflow" in r.stderr or "out of bounds" in r.stderr)
    r = subprocess.run([exe, "barrier"], capture_output=True, text=True)
    assert r.returncode != 0 and "different __syncthreads" in r.stderr


def test_emulated_api_lifecycle_and_allocation_failures(emu, oracle_mod):
    """tests/emu/emu_api.cpp: one scene object across growing and shrinking frames, the begin/end protocol,
    destruction with a frame in flight, every device allocation failing once; leak detection on."""
    subprocess.check_call(["make", "-s", "-C", emu.EMU_DIR, "build/emu_api"])
    r = subprocess.run([emu.EMU_DIR + "/build/emu_api"], capture_output=True, text=True, timeout=900,
                       env={"ASAN_OPTIONS": "detect_leaks=1", "UBSAN_OPTIONS": "print_stacktrace=1"})
    assert r.returncode == 0 and "emu_api ok" in r.stdout, r.stdout[-3000:] + r.stderr[-6000:]


@pytest.mark.parametrize("name", golden_names())
def test_emulated_kernels_reproduce_golden(emu, tmp_path, name):
    data, cam, params, rgba, hit, counters = load_golden(name)
    ps = []
    for search in SEARCHES.values():
        if params.hit_policy == _abi.RTMI_HIT_NEAREST and search == _abi.RTMI_SEARCH_LIST_SCAN:
            continue  # the extension exists for the cone tree only
        ps.append(copy.copy(params))
        ps[-1].search = search
    with ThreadPoolExecutor(len(ps)) as pool:
        # the same camera twice: the second frame takes the cached-preparation path
        jobs = [pool.submit(emu.run_emulated, data, [cam, cam], p, tmp_path / str(p.search)) for p in ps]
        for job, p in zip(jobs, ps):
            frames, stats, _ = job.result()
            check_frames(frames, stats, [(rgba, hit, counters)] * 2, p)


def test_emulated_deeper_tree_moving_camera(emu, oracle_mod, tmp_path):
    # 1280 triangles: three levels of the cone tree; three cameras on one scene object (re-packed per position)
    data = scenes.mesh_scene(level=3, form="B", n_lights=3, bounding_sphere=oracle_mod.bounding_sphere)
    cams = [scenes.make_camera(**scenes.orbit_camera(f, 64)) for f in (0, 5, 5)]
    run_case(emu, oracle_mod, tmp_path, data, cams, scenes.make_params(40, 36, 0.5))


def test_emulated_tiles_and_window(emu, oracle_mod, tmp_path):
    data = scenes.mesh_scene(level=2, form="B", n_lights=2, bounding_sphere=oracle_mod.bounding_sphere)
    cam = scenes.make_camera((0, 0, 4), (0, 0, -1), 10.0)
    run_case(emu, oracle_mod, tmp_path / "tiles", data, [cam], scenes.make_params(70, 50, 0.5, tile_size=32, tile_rank=1, tile_world=3))
    run_case(emu, oracle_mod, tmp_path / "window", data, [cam], scenes.make_params(70, 50, 0.25, window=(13, 7, 58, 41)), searches=("tree",))


def test_emulated_mixed_groups_with_bad_vertices(emu, oracle_mod, tmp_path):
    """Two bounded groups, a top-level run, spheres and a light in the object list; some vertices are not finite."""
    rng = np.random.default_rng(11)

    def soup(n, centre, scale):
        t = np.zeros(n, scenes.TRI_DTYPE)
        c = np.asarray(centre) + rng.uniform(-0.6, 0.6, size=(n, 1, 3))
        t["v"] = c + rng.normal(size=(n, 3, 3)) * scale
        t["material"] = rng.integers(0, 2, size=n)
        return t

    a, b, top = soup(150, (-0.8, 0, 0), 0.25), soup(90, (0.9, 0.2, -0.5), 0.3), soup(40, (0, -0.6, 0.5), 0.35)
    a["v"][7, 1, 2] = np.nan
    b["v"][3, 0, 0] = np.inf
    top["v"][5, 2, 1] = -np.inf
    tris = np.concatenate([a, b, top])
    mats = np.concatenate([scenes.material(**scenes.VC_MATERIAL),
                           scenes.material(Ns=3.5, d=1.0, illum=2, Kd=(0.8, 0.3, 0.2), Ka=(0.2, 0.5, 0.9), Ks=(0.4, 0.9, 0.1))])
    lights = np.concatenate([scenes.light(**scenes.L1), scenes.light(**scenes.L3)])
    ca, ra = oracle_mod.bounding_sphere(a[np.isfinite(a["v"]).all(axis=(1, 2))])
    cb, rb = oracle_mod.bounding_sphere(b[np.isfinite(b["v"]).all(axis=(1, 2))])
    objects = np.concatenate([
        scenes.sphere_object(ca, ra, -1, bounding=True, first=0, count=len(a)),
        scenes.sphere_object((0.2, 0.9, -1.0), 0.4, 1),
        scenes.triangles_object(len(a) + len(b), len(top)),
        scenes.light_object(1),
        scenes.sphere_object(cb, rb, 0, bounding=True, first=len(a), count=len(b)),
    ])
    data = scenes.SceneData(lights=lights, materials=mats, objects=objects, triangles=tris)
    cam = scenes.make_camera((0.1, 0.2, 4), (0, -0.05, -1), 8.0)
    run_case(emu, oracle_mod, tmp_path, data, [cam, cam], scenes.make_params(44, 38, 0.5))


@pytest.mark.parametrize("env", [{"EMU_SEED": "3"}, {"EMU_SEED": "12", "EMU_FILL": "255"}, {"EMU_FILL": "0", "EMU_CUS": "5"},
                                 {"RTMI_MAX_BATCH_SAMPLES": "2048", "EMU_SEED": "9"}])
def test_emulated_schedules_fill_patterns_batches(emu, oracle_mod, tmp_path, env):
    """Other interleavings of the waves of a workgroup, other contents of fresh device memory, another grid
    size, a frame cut into several batches: none of them may change a byte."""
    data = scenes.mesh_scene(level=2, form="T", n_lights=3)
    data_b = scenes.mesh_scene(level=2, form="B", n_lights=3, bounding_sphere=oracle_mod.bounding_sphere)
    cam = scenes.make_camera((0, 0, 4), (0, 0, -1), 10.0)
    run_case(emu, oracle_mod, tmp_path / "T", data, [cam], scenes.make_params(36, 30, 0.5), searches=("tree",), env=env)
    run_case(emu, oracle_mod, tmp_path / "B", data_b, [cam, cam], scenes.make_params(36, 30, 0.5), env=env)


@pytest.mark.parametrize("capacity", ["64", "1500"])
def test_emulated_candidate_list_overflow(emu, oracle_mod, tmp_path, capacity):
    """Candidate lists far too small for the frame (RTMI_CAND_CAPACITY): the search must notice, every kernel
    after it must stay within bounds on the incomplete lists, and the frame comes out right through the list scan."""
    cam = scenes.make_camera((0, 0, 4), (0, 0, -1), 10.0)
    env = {"RTMI_CAND_CAPACITY": capacity}
    data_b = scenes.mesh_scene(level=2, form="B", n_lights=3, bounding_sphere=oracle_mod.bounding_sphere)
    out_b = run_case(emu, oracle_mod, tmp_path / "B", data_b, [cam, cam], scenes.make_params(36, 30, 0.5), searches=("tree",), env=env)
    data_t = scenes.mesh_scene(level=2, form="T", n_lights=2)
    out_t = run_case(emu, oracle_mod, tmp_path / "T", data_t, [cam, cam], scenes.make_params(30, 26, 0.5), searches=("tree",), env=env)
    for out in (out_b, out_t):
        for _, _, cnt in out["tree"]:  # the counters are those of the second rendering: no tree nodes, a scan instead
            assert cnt.node_tests == 0 and cnt.tri_tests_primary_kernel > 0


def test_emulated_candidate_list_overflow_under_the_nearest_hit_policy(emu, oracle_mod, tmp_path):
    """The nearest-hit policy has no list scan to fall back on: the lists are made longer and the frame is rendered
    again with the tree until they hold it (two frames, so the second starts from the grown lists)."""
    cam = scenes.make_camera((0, 0, 4), (0, 0, -1), 10.0)
    data_b = scenes.mesh_scene(level=2, form="B", n_lights=3, bounding_sphere=oracle_mod.bounding_sphere)
    p = scenes.make_params(36, 30, 0.5, hit_policy=_abi.RTMI_HIT_NEAREST)
    out = run_case(emu, oracle_mod, tmp_path / "B", data_b, [cam, cam], p, searches=("tree",), env={"RTMI_CAND_CAPACITY": "64"})
    for _, _, cnt in out["tree"]:
        assert cnt.node_tests > 0 and cnt.tri_candidates > 64


def test_emulated_random_scenes(emu, oracle_mod):
    """A few cases of the scene fuzzer (tests/emu_fuzz.py); long runs of it are done by hand."""
    import emu_fuzz
    with ThreadPoolExecutor(4) as pool:
        list(pool.map(lambda i: emu_fuzz.run_one(5, i, oracle_mod), range(8)))


@pytest.mark.parametrize("form,level,size", [("B", 4, (40, 36, 0.5)), ("T", 4, (28, 24, 1.0)), ("B", 7, (12, 10, 1.0))])
def test_emulated_many_triangles(emu, oracle_mod, tmp_path, form, level, size):
    """5,120 triangles: the cone tree spans several workgroups of the packing kernel and its upper levels are merged
    by the workgroup that finishes last.  327,680 triangles: six levels, and the list scan's compaction scans more
    block counts than it has threads."""
    data = scenes.mesh_scene(level=level, form=form, n_lights=3, bounding_sphere=oracle_mod.bounding_sphere)
    cam = scenes.make_camera((0, 0, 4), (0, 0, -1), 10.0)
    run_case(emu, oracle_mod, tmp_path, data, [cam, cam], scenes.make_params(*size))


@pytest.mark.parametrize("rays_per_lane", ["4", "8"])
def test_emulated_other_rays_per_lane(emu, oracle_mod, tmp_path, rays_per_lane):
    """RTMI_SEARCH_R selects the rays-per-lane instantiation of both search kernels (small frames take 2 in the tree
    search and 4 in the list scan by themselves)."""
    env = {"RTMI_SEARCH_R": rays_per_lane}
    cam = scenes.make_camera((0, 0, 4), (0, 0, -1), 10.0)
    data_b = scenes.mesh_scene(level=3, form="B", n_lights=2, bounding_sphere=oracle_mod.bounding_sphere)
    run_case(emu, oracle_mod, tmp_path / "B", data_b, [cam], scenes.make_params(37, 29, 0.5), env=env)
    data_t = scenes.mesh_scene(level=2, form="T", n_lights=2)
    run_case(emu, oracle_mod, tmp_path / "T", data_t, [cam], scenes.make_params(30, 22, 0.5), env=env)


def test_emulated_primary_rays_hit_point_lights(emu, oracle_mod, tmp_path):
    """PointLight as a member of `objects` (Renderables.swift:80-91): a hit without a normal, shaded with the ambient
    term only (RayTracingFunctions.swift:199-202)."""
    data, cam, params, pixels = light_hit_scene(oracle_mod)
    _, o_hit, _ = oracle_mod.render(data, cam, params)
    ids = {int(o_hit[j, i, 0]) for i, j in pixels}
    assert ids <= {-2, -3, -5, -6} and len(ids) >= 3, ids  # the lights are hit (one may lie behind the sphere in list order)
    run_case(emu, oracle_mod, tmp_path, data, [cam], params, searches=("tree",))


DEGENERATE_CAMERAS = {
    "zero_dir": ((0, 0, 4), (0, 0, 0), 10.0),              # normalize(-dir) is NaN: every direction is NaN, nothing is hit
    "dir_parallel_to_up": ((0, -4, 4), (0, 1, -1), 10.0),  # up x n = 0 (RayTracingFunctions.swift:50-53)
    "focal_zero": ((0, 0, 4), (0, 0, -1), 0.0),            # the image plane passes through the camera
    "focal_negative": ((0, 0, -4), (0, 0, -1), -10.0),
    "focal_huge": ((0, 0, 4), (0, 0, -1), 1e300),
    "pos_nan": ((float("nan"), 0, 4), (0, 0, -1), 10.0),
    "pos_inf": ((float("inf"), 0, 4), (0, 0, -1), 10.0),
    "dir_huge": ((0, 0, 4), (0, 0, -1e200), 10.0),
    "dir_tiny": ((0, 0, 4), (0, 0, -1e-200), 10.0),
    "pos_far": ((0, 0, 4e15), (0, 0, -1), 4e15),           # FP32 filters see the scene as a point
}


@pytest.mark.parametrize("form", ["B", "T"])
def test_emulated_degenerate_cameras(emu, oracle_mod, tmp_path, form):
    """Cameras the reference would accept from its text fields and then render garbage (or trap) with: NaN and
    infinite directions must not become addresses, and what is rendered must still equal the oracle's frame."""
    data = scenes.mesh_scene(level=2, form=form, n_lights=2, bounding_sphere=oracle_mod.bounding_sphere)
    if form == "B":
        data.objects = np.concatenate([data.objects, scenes.sphere_object((0.5, 0.5, 1.0), 0.3, 0)])
    cams = [scenes.make_camera(*c) for c in DEGENERATE_CAMERAS.values()]
    run_case(emu, oracle_mod, tmp_path, data, cams, scenes.make_params(20, 16, 0.5))


RACE_CASES = [
    ("B", 2, (36, 30, 0.5), "tree", {}),
    ("B", 2, (36, 30, 0.5), "scan", {"EMU_SEED": "4"}),
    ("T", 2, (30, 24, 0.5), "tree", {"EMU_SEED": "2"}),
    ("T", 2, (30, 24, 0.5), "scan", {}),
    ("B", 2, (36, 30, 0.5), "tree", {"RTMI_CAND_CAPACITY": "64", "EMU_SEED": "6"}),      # overflowing candidate lists
    ("B", 4, (30, 24, 0.5), "tree", {"RTMI_MAX_BATCH_SAMPLES": "2048", "EMU_SEED": "3"}),  # batches; tree packed by 10 workgroups
    ("B", 2, (24, 20, 0.5), "scan", {"CAMERA": "inside"}),  # triangles that subtend wide angles: flagged batches of the scan
    ("T", 2, (24, 20, 0.5), "tree", {"CAMERA": "inside"}),
]


def test_emulated_kernels_have_no_data_races(emu, oracle_mod, tmp_path):
    """The same kernels under ThreadSanitizer (make -C tests/emu tsan): every work-item is a sanitizer fiber, and
    only wave-level operations, __syncthreads(), atomics and the end of a workgroup order them.  Two work-items of
    a workgroup that touch one LDS or global location, one of them writing, with none of those in between, are
    reported -- the kind of defect that shows on a GPU as a hang or a fault once in a while."""
    subprocess.check_call(["make", "-s", "-C", emu.EMU_DIR, "tsan"])
    r = subprocess.run([emu.EMU_DIR + "/build_tsan/emu_selftest", "race"], capture_output=True, text=True)
    assert "ThreadSanitizer: data race" in r.stderr and "k_race" in r.stderr  # the detector detects
    r = subprocess.run([emu.EMU_DIR + "/build_tsan/emu_selftest", "ok"], capture_output=True, text=True)
    assert "selftest ok" in r.stdout and "ThreadSanitizer" not in r.stderr

    # the host side: two host threads, a scene object each (tests/emu/emu_api.cpp, last part)
    r = subprocess.run([emu.EMU_DIR + "/build_tsan/emu_api", "threads"], capture_output=True, text=True, timeout=900,
                       env={"TSAN_OPTIONS": "halt_on_error=0 exitcode=0"})
    assert "emu_api ok" in r.stdout and not emu.race_reports(r.stderr), r.stdout[-500:] + r.stderr[-3000:]

    def one(case):
        k, (form, level, size, search, env) = case
        data = scenes.mesh_scene(level=level, form=form, n_lights=3, bounding_sphere=oracle_mod.bounding_sphere)
        env = dict(env)
        inside = env.pop("CAMERA", "") == "inside"
        cam = scenes.make_camera((0.05, 0.02, 0.1), (0.1, 0.3, -1.0), 2.0) if inside else scenes.make_camera((0, 0, 4), (0, 0, -1), 10.0)
        p = scenes.make_params(*size)
        p.search = SEARCHES[search]
        expected = [oracle_mod.render(data, cam, p, threads=2)] * 2
        e = dict(env, TSAN_OPTIONS="halt_on_error=0 exitcode=0 report_signal_unsafe=0")
        frames, stats, _, stderr = emu.run_emulated(data, [cam, cam], p, tmp_path / str(k), env=e, binary=emu.TSAN_BIN,
                                                    with_stderr=True, timeout=1800)
        check_frames(frames, stats, expected, p)
        return emu.race_reports(stderr)

    with ThreadPoolExecutor(8) as pool:
        for case, reports in zip(RACE_CASES, pool.map(one, enumerate(RACE_CASES))):
            assert not reports, (case, reports[:3])


def test_emulated_lean_shading_hands_over_to_the_full_kernel(emu, oracle_mod, tmp_path):
    """k_shade_candidates shades with the lean instantiation and queues what it cannot finish: a material with a
    fractional specular exponent (no sphere in the scene, so nothing else fills the shade queue), and a reflected
    ray that finds a sphere with a material."""
    data = scenes.mesh_scene(1, "B", 2, bounding_sphere=oracle_mod.bounding_sphere)
    data.materials = scenes.material(Ns=3.5, d=1.0, illum=2, Kd=(0.8, 0.3, 0.2), Ka=(0.2, 0.5, 0.9), Ks=(0.4, 0.9, 0.1))
    cam = scenes.make_camera(**scenes.MESH_CAMERA)
    run_case(emu, oracle_mod, tmp_path / "ns", data, [cam], scenes.make_params(40, 32, 0.5), searches=("tree",))
    data = scenes.mesh_scene(1, "B", 2, bounding_sphere=oracle_mod.bounding_sphere)
    data.objects = np.concatenate([data.objects, scenes.sphere_object((1.9, 0.3, 0.4), 0.7, 0)])
    run_case(emu, oracle_mod, tmp_path / "sphere", data, [cam], scenes.make_params(48, 32, 0.5), searches=("tree",))


@pytest.mark.parametrize("flags", [_abi.RTMI_FLAG_SHADOW_MAX_DISTANCE, _abi.RTMI_FLAG_SHADOWS_ENTER_BOUNDED,
                                   _abi.RTMI_FLAG_SHADOW_MAX_DISTANCE | _abi.RTMI_FLAG_SHADOWS_ENTER_BOUNDED,
                                   _abi.RTMI_FLAG_REFLECTIONS_ENTER_BOUNDED, 14])
def test_emulated_shadow_extensions(emu, oracle_mod, tmp_path, flags):
    """The extensions of the secondary rays (a distance limit for shadow rays; shadow rays and reflection rays that
    enter bounding spheres): every hit goes through the shade queue to the recording and replaying kernels (to
    k_shade_ext under the nearest-hit policy), which walk the top-level runs' tree and the tree over the bounded
    children.  Both searches, both hit policies (the nearest-hit policy with the cone tree only), two frames on one
    scene object (the bounded tree is built by the first)."""
    changed = 0
    for k, (data, cam) in enumerate(shadow_extension_cases(oracle_mod.bounding_sphere)):
        out = run_case(emu, oracle_mod, tmp_path / f"lo{k}", data, [cam, cam], scenes.make_params(40, 32, 0.5, flags=flags))
        plain = oracle_mod.render(data, cam, scenes.make_params(40, 32, 0.5), threads=4)[0]
        changed += int((out["tree"][0][0] != plain).any(axis=-1).sum())
        run_case(emu, oracle_mod, tmp_path / f"nr{k}", data, [cam],
                 scenes.make_params(40, 32, 0.5, flags=flags, hit_policy=_abi.RTMI_HIT_NEAREST), searches=("tree",))
    assert changed > 0  # the flags do change these frames


def test_emulated_pixels_only_flag_gives_the_same_frame(emu, oracle_mod, tmp_path):
    """RTMI_FLAG_NO_RAY_STATISTICS on a scene of bounding spheres without material (what loadObjFile builds): the
    shading instantiation without secondary rays runs; bitmap and hit ids equal the oracle's, no secondary ray
    is counted.  On a scene with a sphere that has a material the flag changes nothing."""
    cam = scenes.make_camera(**scenes.MESH_CAMERA)
    data = scenes.mesh_scene(1, "B", 3, bounding_sphere=oracle_mod.bounding_sphere)
    p = scenes.make_params(40, 32, 0.5, flags=_abi.RTMI_FLAG_NO_RAY_STATISTICS)
    o_rgba, o_hit, o_cnt = oracle_mod.render(data, cam, scenes.make_params(40, 32, 0.5), threads=4)
    for env, kernel in (({"RTMI_FAST": "0"}, "k_shade_candidates<false, true>"), (None, "k_shade_candidates_fast")):
        # the general path's instantiation without secondary rays, then the path without a ray-generation launch
        frames, stats, _ = emu.run_emulated(data, [cam, cam], p, tmp_path / ("inert" + kernel[-4:]), env=env)
        for rgba, hit, cnt in frames:
            assert (hit[:31, :39] == o_hit[:31, :39]).all() and (rgba[:31, :39] == o_rgba[:31, :39]).all()
            assert cnt.primary_hits == o_cnt["primary_hits"] > 0
            assert cnt.rays_reflection == 0 and cnt.rays_shadow == 0 and cnt.shadow_rays_elided == 0
        assert any(kernel in k for k in stats), sorted(stats)
        assert ("k_gen" in " ".join(stats)) == (env is not None)
    data.objects = np.concatenate([data.objects, scenes.sphere_object((1.9, 0.3, 0.4), 0.7, 0)])
    o_rgba, o_hit, o_cnt = oracle_mod.render(data, cam, scenes.make_params(40, 32, 0.5), threads=4)
    frames, stats, _ = emu.run_emulated(data, [cam], p, tmp_path / "not_inert")
    rgba, hit, cnt = frames[0]
    assert (hit[:31, :39] == o_hit[:31, :39]).all() and (rgba[:31, :39] == o_rgba[:31, :39]).all()
    assert cnt.rays_reflection == o_cnt["rays_reflection"] and cnt.rays_shadow == o_cnt["rays_shadow"]
    assert not any("k_shade_candidates<false, true>" in k for k in stats)


@pytest.mark.parametrize("env", [None, {"RTMI_TRACE_THEN_SHADE": "0"}, {"RTMI_NORMAL_CONES": "0"}])
def test_emulated_secondary_rays_of_toplevel_triangles(emu, oracle_mod, tmp_path, env):
    """Secondary rays that do find things -- a non-convex top-level mesh next to a sphere, a light in the object
    list -- traced in one kernel and shaded in another (the default), traced by the shading kernel itself, and
    without the cones of normals in the bounding-sphere tree: always the oracle's frame and ray counts."""
    data = toplevel_mesh_with_company(3)
    cams = [scenes.make_camera((0.9, 0.5, 4.0), (-0.12, -0.1, -1.0), 8.0)]
    run_case(emu, oracle_mod, tmp_path, data, cams, scenes.make_params(46, 34, 0.5), searches=("tree",), env=env)


def test_emulated_more_lights_than_the_trace_record_holds(emu, oracle_mod, tmp_path):
    """Seventeen lights: one more than the recorded shadow-ray outcomes of a sample have room for, so the frame
    goes through the shading kernel that traces its own rays; sixteen lights still take the two-kernel path."""
    cams = [scenes.make_camera((0.9, 0.5, 4.0), (-0.12, -0.1, -1.0), 8.0)]
    params = scenes.make_params(30, 24, 1.0)
    for n in (17, 16):
        data = toplevel_mesh_with_company(n, level=1)
        expected = [oracle_mod.render(data, c, params, threads=8) for c in cams]
        frames, stats, _ = emu.run_emulated(data, cams, params, tmp_path / str(n))
        check_frames(frames, stats, expected, params)
        recorded = any("k_shade_replay<" in name for name in stats)
        assert recorded == (n == 16), sorted(stats)
        assert any("k_shade<true, " in name for name in stats) == (n == 17), sorted(stats)


def bounded_spheres_scene(oracle_mod, n_groups=1, n_lights=3, extra_materials=False, level=2):
    """Nothing but bounding spheres without a material in the list (what loadObjFile builds, one or several times):
    the scenes of the path without a ray-generation launch.  With extra_materials the triangles use three materials
    (exponents 8, 1 and 33) and every seventh triangle has none."""
    parts, objs, first = [], [], 0
    for g in range(n_groups):
        v, f = scenes.displaced_icosphere(level if g == 0 else 1)
        tr = scenes.triangles_from_vertices(v * (1.0 - 0.35 * g) + np.array([0.9 * g, -0.3 * g, 0.6 * g]), f, 0)
        c, r = oracle_mod.bounding_sphere(tr)
        objs.append(scenes.sphere_object(c, r, -1, bounding=True, first=first, count=len(tr)))
        parts.append(tr)
        first += len(tr)
    tris = np.concatenate(parts)
    mats = scenes.material(**scenes.VC_MATERIAL)
    if extra_materials:
        mats = np.concatenate([mats, scenes.material(1.0, 1.0, 2, (0.9, 0.2, 0.1), (0.3, 0.6, 0.1), (0.2, 0.2, 0.9)),
                               scenes.material(33.0, 1.0, 2, (0.1, 0.4, 0.7), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))])
        tris["material"] = np.arange(len(tris)) % 3
        tris["material"][::7] = -1
    lights = np.concatenate([scenes.light(**L) for L in (scenes.L1, scenes.L2, scenes.L3)[:n_lights]])
    return scenes.SceneData(lights=lights, materials=mats, objects=np.concatenate(objs), triangles=tris)


FAST_CASES = [
    # groups, materials, (W, H, sub), params extras, env
    (1, False, (40, 32, 0.25), {}, None),
    (1, True, (37, 29, 1.0), {}, None),
    (1, True, (33, 27, 1.0 / 3.0), {}, None),                                 # 9 samples per pixel: groups of 64 samples straddle blocks
    (1, False, (40, 32, 0.5), {"window": (5, 3, 31, 22)}, None),
    (1, True, (48, 40, 0.5), {"tile_size": 8, "tile_rank": 1, "tile_world": 3}, None),
    (1, False, (48, 40, 0.5), {"tile_size": 12, "tile_rank": 0, "tile_world": 2}, None),   # tiles that cut 8x8 blocks
    (3, True, (44, 36, 0.5), {}, None),                                       # later groups are always decided in FP64
    (2, False, (40, 32, 0.5), {}, {"RTMI_MAX_BATCH_SAMPLES": "2048"}),        # several batches
    (1, True, (40, 32, 0.5), {}, {"RTMI_CAND_CAPACITY": "64"}),               # overflow: rendered again by the list scan
    (1, True, (40, 32, 0.5), {}, {"RTMI_FAST_FORCE": "1"}),                   # every pair decided in FP64
    (1, True, (40, 32, 0.5), {}, {"RTMI_FAST_FORCE": "2", "EMU_SEED": "3"}),  # every pixel shaded again in FP64
    (2, True, (36, 30, 0.25), {}, {"RTMI_FAST_FORCE": "3", "RTMI_SEARCH_R": "4"}),
    (1, True, (40, 32, 0.5), {}, {"RTMI_MULTI_CAPACITY": "64"}),              # the list of candidates of rays with several overflows
    (1, True, (72, 56, 0.5), {}, {"RTMI_LIVE_CAPACITY": "3"}),                # more live blocks than the arrays hold: rendered again with more
    (2, False, (64, 48, 0.25), {}, {"RTMI_LIVE_CAPACITY": "2", "RTMI_MAX_BATCH_SAMPLES": "8192", "EMU_CUS": "1"}),
    (1, True, (40, 32, 0.5), {}, {"RTMI_BLOCK_ENTRIES": "0"}),                # every walk starts at the root
    (2, True, (40, 32, 0.5), {}, {"RTMI_UPLOAD_KERNEL": "0"}),                # the frame's small uploads as separate calls
    (1, True, (56, 40, 0.25), {"_level": 3}, None),                           # 1280 triangles: a tree of four record levels
    (1, True, (40, 32, 0.5), {}, {"RTMI_LEAF_MASKS": "2", "RTMI_SEARCH_R": "4"}),   # start records name the triangles to look at
    (1, True, (56, 40, 0.25), {"_level": 3}, {"RTMI_LEAF_MASKS": "2", "RTMI_SEARCH_R": "4", "EMU_SEED": "5"}),
    (3, False, (44, 36, 0.5), {}, {"RTMI_LEAF_MASKS": "2", "RTMI_SEARCH_R": "4"}),
]


@pytest.mark.parametrize("case", range(len(FAST_CASES)))
def test_emulated_frames_without_a_ray_generation_launch(emu, oracle_mod, tmp_path, case):
    """Pixels only on scenes of bounding spheres: FP32 directions from the sample tables, candidates decided by FP32
    margins or in FP64, FP32 lighting with an error bound, the pixels it cannot settle shaded again in FP64.  Hit ids and
    bytes must be the oracle's, for three camera positions in a row on one scene object (one of them inside the mesh)."""
    n_groups, mats, (W, H, sub), extra, env = FAST_CASES[case]
    extra = dict(extra)
    data = bounded_spheres_scene(oracle_mod, n_groups, extra_materials=mats, level=extra.pop("_level", 2))
    cams = [scenes.make_camera(**scenes.MESH_CAMERA), scenes.make_camera((0.7, 0.4, 3.6), (-0.15, -0.1, -1.0), 7.0),
            scenes.make_camera((0.05, 0.0, 0.3), (0.0, 0.2, -1.0), 3.0), scenes.make_camera(**scenes.MESH_CAMERA)]
    p = scenes.make_params(W, H, sub, flags=_abi.RTMI_FLAG_NO_RAY_STATISTICS, **extra)
    expected = [oracle_mod.render(data, c, p, threads=8)[:2] + (None,) for c in cams]
    frames, stats, _ = emu.run_emulated(data, cams, p, tmp_path, env=env)
    check_frames(frames, stats, expected, p)
    names = " ".join(stats)
    overflowed = bool(env and "RTMI_CAND_CAPACITY" in env)  # such a frame is rendered again by the general path's list scan
    assert "k_shade_candidates_fast" in names and "k_resolve_fast" in names and ("k_gen" in names) == overflowed
    assert ("k_block_entries" in names) == (not (env and env.get("RTMI_BLOCK_ENTRIES") == "0"))
    hits = sum(int(o[2]["primary_hits"]) for o in [oracle_mod.render(data, c, p, threads=8) for c in cams[:1]])
    assert frames[0][2].primary_hits == hits > 0
    force = int((env or {}).get("RTMI_FAST_FORCE", "0"))
    if force & 1:
        assert all(f[2].hits_decided_fp32 == 0 for f in frames)
    elif n_groups == 1 and not overflowed:
        assert frames[0][2].hits_decided_fp32 > 0.9 * frames[0][2].primary_hits
    if force & 2:
        assert frames[0][2].pixels_fixed_fp64 >= (frames[0][1][..., 0] >= 0).sum()
