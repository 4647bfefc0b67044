#!/usr/bin/env python3
"""Benchmark of the render hot path on MI355X.

A step is one full frame of the headline configuration: the stand-in "pig" mesh (5,120-triangle
displaced icosphere wrapped in Sphere(boundingObjects:), i.e. form B, what the reference's
loadObjFile builds), 3840x2160, 16 samples per pixel (pixelSubdivision 0.25), 3 lights, shadow and
reflection rays, reference-exact list-order hit policy.  With N > 1 the frame's 32x32 tiles are
dealt cyclically over the ranks (strong scaling: total work per step is fixed), the scene is
broadcast once over RCCL before the timed region and every rank's own pixels are gathered on rank 0
inside it.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOPS_PER_TEST = 69            # reference triangle test with ray-independent terms hoisted
PEAK_FP32_TFLOPS = 157.3       # MI355X vector FP32 (256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz)
PEAK_HBM_GBPS = 8000.0         # MI355X HBM3E
SHADE_BYTES_PER_HIT = 64       # algorithmic HBM bytes of the shading kernels per shaded hit (see the roofline object)
PEAK_FP64_TFLOPS = 78.6        # MI355X vector FP64 (256 CU x 4 SIMD x 32 FLOP/clk x 2.4 GHz; v_fma_f64 measured at 73 TFLOP/s)


def shade_flops_per_hit(n_lights, n_sphere_like):
    """FP64 operations of calculateLighting (RayTracingFunctions.swift:182-271) for one triangle hit with a
    material in a scene whose other objects are `n_sphere_like` material-less bounding spheres (form B);
    +, -, *, /, sqrt and pow count 1 each, a normalize is 9 (dot 5, sqrt, 3 divisions):
    hit point 6, surfaceToCamera 12, reflection direction 12, the reflection ray's sphere tests
    22 each + 13 for the hit it finds, the unlit reflected hit's light loop 26 per light,
    and per light of the primary hit 57 (direction 12, n.l 5, diffuse 9, reflection of the light 16,
    r.v 5, pow 1, specular 9)."""
    return 6 + 12 + 12 + 22 * n_sphere_like + 13 + 26 * n_lights + 57 * n_lights

CONFIGS = {
    # name: (width, height, subdivision, mesh level, n_lights)
    "C2": (1024, 1024, 0.5, 4, 1),
    "C3": (3840, 2160, 0.25, 4, 3),
    "C4": (1920, 1080, 0.5, 8, 1),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C3", choices=sorted(CONFIGS))
    ap.add_argument("--form", default="B", choices=["B", "T"])
    ap.add_argument("--search", default="tree", choices=["tree", "scan"],
                    help="triangle search of the timed region: the cone tree (default) or the list scan")
    ap.add_argument("--no-scan-leg", action="store_true",
                    help="skip the extra list-scan frames that measure BASELINE's tests/s figure (1 GPU only)")
    ap.add_argument("--orbit", action="store_true", help="config 5: move the camera every step")
    ap.add_argument("--ray-stats", action="store_true",
                    help="keep the reference's ray statistics in the timed frames (every ray the reference would trace is "
                         "counted, which costs the tracing of rays that cannot change a pixel)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--verify", action="store_true",
                    help="after timing, rank 0 re-renders the last frame alone and compares it byte for byte")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="multi-rank logic check on a box with ONE GPU: all ranks render on device 0 and the "
                         "collectives run over gloo on host tensors (not a performance mode)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import raytracer_amd as rt
    from raytracer_amd import scenes, sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    if args.rehearse_gloo:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    comm_dev = torch.device("cpu") if args.rehearse_gloo else dev
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_gloo:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    W, H, sub, level, n_lights = CONFIGS[args.config]
    data = None
    mesh_note = "stand-in (no pig OBJ available offline)"
    if rank == 0:
        pig = os.environ.get("RTMI_PIG_OBJ")
        if pig and args.config != "C4":
            # a real mesh, if one is supplied on the box: triangulated, vertex-index-only OBJ (the
            # reference's parser accepts nothing else); same wrapper, lights and material as the stand-in
            tris, _, _, _ = rt.load_obj(pig)
            tris["material"] = 0
            data = scenes.mesh_scene(0, args.form, n_lights, bounding_sphere=rt.bounding_sphere)
            data.triangles = tris
            if args.form == "B":
                c, r = rt.bounding_sphere(tris)
                data.objects = scenes.sphere_object(c, r, -1, bounding=True, first=0, count=len(tris))
            else:
                data.objects = scenes.triangles_object(0, len(tris))
            mesh_note = f"real OBJ {os.path.basename(pig)} ({len(tris)} triangles)"
        else:
            data = scenes.mesh_scene(level, args.form, n_lights, bounding_sphere=rt.bounding_sphere)
    if world > 1:
        data = sharding.broadcast_scene(data, 0, device=comm_dev)  # RCCL byte broadcast, once
    scene = rt.Scene(data, device=local_rank)
    from raytracer_amd import _abi
    search = _abi.RTMI_SEARCH_CONE_TREE if args.search == "tree" else _abi.RTMI_SEARCH_LIST_SCAN
    # timed frames ask for pixels only (RTMI_FLAG_NO_RAY_STATISTICS): rays whose outcome cannot change a pixel are then
    # neither traced nor counted; --ray-stats keeps the reference's full ray statistics inside the timed region
    flags = 0 if args.ray_stats else _abi.RTMI_FLAG_NO_RAY_STATISTICS
    params = scenes.make_params(W, H, sub, search=search, flags=flags)
    my_params = sharding.tile_params(params, rank, world) if world > 1 else params
    composer = sharding.TileComposer(W, H, rank, world, comm_dev)
    staging = [torch.zeros((H, W, 4), dtype=torch.uint8, device=dev) for _ in range(2)] if args.rehearse_gloo else None
    torch.cuda.synchronize()

    def camera_for(step):
        if args.orbit:
            return scenes.make_camera(**scenes.orbit_camera(step % 64))
        return scenes.make_camera(**scenes.MESH_CAMERA)

    totals = {"rays": 0, "refl": 0, "shadow": 0, "elided": 0, "tests": 0, "tests_primary": 0, "ms_primary": 0.0, "launches": 0, "ms_total": 0.0,
              "exact": 0, "samples": 0, "hits": 0, "ms_shade": 0.0, "node_tests": 0, "candidates": 0}

    # With more than one rank a frame is short (1/N of the work) and the gaps between frames start to show: each
    # rank then keeps TWO frames in flight, on two scene objects with a stream and work buffers of their own
    # (measured on the 1/8 share of this workload: 0.524 -> 0.493 ms per frame).  Frame i is enqueued before
    # frame i-1 is waited for; the gather of frame i-1 then overlaps the rendering of frame i.
    scenes_in_flight = [scene] + ([rt.Scene(data, device=local_rank)] if world > 1 else [])
    depth = len(scenes_in_flight)
    in_flight = []  # (scene, slot, record) of frames enqueued and not yet waited for

    def account(cnt):
        totals["rays"] += cnt["samples"] + cnt["rays_reflection"] + cnt["rays_shadow"]
        totals["samples"] += cnt["samples"]
        totals["refl"] += cnt["rays_reflection"]
        totals["shadow"] += cnt["rays_shadow"]
        totals["elided"] += cnt["shadow_rays_elided"]
        totals["tests"] += cnt["tri_tests"]
        totals["tests_primary"] += cnt["tri_tests_primary_kernel"]
        totals["ms_primary"] += cnt["ms_primary_kernel"]
        totals["ms_total"] += cnt["ms_total"]
        totals["launches"] += cnt["tri_search_launches"]
        totals["exact"] += cnt["tri_exact_tests"]
        totals["hits"] += cnt["primary_hits"]
        totals["ms_shade"] += cnt["ms_shade"]
        totals["node_tests"] += cnt["node_tests"]
        totals["candidates"] += cnt["tri_candidates"]

    def retire_oldest():
        sc_, slot_, record_ = in_flight.pop(0)
        cnt = sc_.render_end()
        if args.rehearse_gloo:
            composer.frames[slot_].copy_(staging[slot_])
        composer.compose(slot_)  # travels to rank 0 while the next frame renders
        if record_:
            account(cnt)

    def step(i, record):
        # Frames alternate between two buffers.  Frame i reuses the buffer of frame i-2: that frame is waited
        # for and its gather enqueued (retire_oldest), and the gather's read of the buffer must have happened
        # (wait_slot), before frame i is enqueued.  Frame i-1 keeps the GPU busy meanwhile.
        slot = i & 1
        if len(in_flight) == depth:
            retire_oldest()
        composer.wait_slot(slot)
        sc_ = scenes_in_flight[i % depth]
        target = staging[slot] if args.rehearse_gloo else composer.frames[slot]
        sc_.render_begin(camera_for(i), my_params, target.data_ptr())   # enqueue frame i, do not wait
        in_flight.append((sc_, slot, record))

    def flush():
        while in_flight:
            retire_oldest()

    for i in range(args.warmup):
        step(i, False)
    flush()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i, True)
    flush()  # the last frame's gather belongs to the timed region
    fence()
    elapsed = time.perf_counter() - t0

    verified = None
    if args.verify and rank == 0:
        alone = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        scene.render_device(camera_for(args.warmup + args.steps - 1), params, alone.data_ptr())
        last = composer.frames[(args.warmup + args.steps - 1) & 1]
        verified = bool((alone.cpu() == last.cpu()).all())
        assert verified, "the composed multi-rank frame differs from the single-GPU frame"

    agg = torch.tensor([float(totals["rays"]), float(totals["tests"]), elapsed], dtype=torch.float64, device=comm_dev)
    if world > 1:
        sums = agg.clone()
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        mx = agg.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        total_rays, total_tests, elapsed_max = sums[0].item(), sums[1].item(), mx[2].item()
    else:
        total_rays, total_tests, elapsed_max = agg[0].item(), agg[1].item(), elapsed

    scan_leg = None
    if rank == 0 and world == 1 and args.search == "tree" and not args.no_scan_leg:
        scan_leg = list_scan_leg(scene, camera_for(args.warmup), params, composer.frames[0], min(3, max(1, args.steps)), args.config)

    if rank == 0:
        mrays = total_rays / elapsed_max / 1e6
        tests_per_s = total_tests / elapsed_max
        launches = max(totals["launches"], 1)
        if args.search == "scan":
            avg_ms = totals["ms_primary"] / launches
            roofline = scan_roofline(totals["tests_primary"] / launches, avg_ms, args.config)
        else:
            # the dominant kernel of the timed region is the shading of the decided hits (FP64)
            flops_hit = shade_flops_per_hit(n_lights, 1 if args.form == "B" else 0)
            ms_shade = totals["ms_shade"] / args.steps
            hits = totals["hits"] / args.steps
            achieved = hits * flops_hit / (ms_shade * 1e-3) / 1e12 if ms_shade > 0 else 0.0
            roofline = {
                "bound": "valu_fp64",
                "kernel": "k_shade_candidates" if args.form == "B" else "k_shade",
                "achieved": round(achieved, 3),
                "peak": PEAK_FP64_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_FP64_TFLOPS, 4),
                "traffic": traffic_from_profile(args.config, "k_shade_candidates"),
                "traffic_unit": "bytes of HBM traffic per launch (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes; profiles/)",
                "flops_per_hit_algorithmic": flops_hit,
                "hits_per_launch": hits,
                "avg_launch_ms": round(ms_shade, 4),
                # the same launch against the HBM roofline: per shaded hit the kernel reads its candidate entry (8 B),
                # the ray's best list position (4 B) and direction (24 B) and writes hit id (4 B) and colour (24 B);
                # triangles, planes, materials and lights are a few hundred KB and stay in cache
                "hbm_view": {"algorithmic_bytes_per_hit": SHADE_BYTES_PER_HIT,
                             "achieved": round(hits * SHADE_BYTES_PER_HIT / (ms_shade * 1e-3) / 1e9, 1) if ms_shade > 0 else 0.0,
                             "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                             "frac": round(hits * SHADE_BYTES_PER_HIT / (ms_shade * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4) if ms_shade > 0 else 0.0},
                "note": "achieved = shaded hits per launch x the FP64 operations of calculateLighting per hit (division, sqrt, "
                        "pow counted as 1) / launch time (HIP events on the library's stream).  An IEEE FP64 division costs "
                        "about 13 instructions on this ISA and a sqrt about 20, and the reference's arithmetic (3 divisions "
                        "per normalize) is kept operation for operation, so the kernel executes several instructions "
                        "per counted operation (about 2.6 on this kernel: 820 VALU instructions per 64 shaded hits, profiles/).",
            }
        out = {
            "metric": "Mrays/s + ray-tri tests/s (% FP32 peak), pig OBJ 4K 16spp, 1/2/4/8 GPU",
            "value": round(mrays, 3),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed_max / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32 filter + f64 decision",
            "data": "synthetic",
            "config": {
                "workload": f"{args.config}: pig mesh [{mesh_note}; default = displaced icosphere level {level}], "
                            f"{len(data.triangles)} triangles, {W}x{H}, {scene_spp(sub)} spp, {n_lights} lights, "
                            f"form {args.form} ({'bounded by Sphere(boundingObjects:)' if args.form == 'B' else 'top-level triangles'}), "
                            "list-order hit policy" + (", orbiting camera" if args.orbit else ""),
                "mesh": mesh_note,
                "search": "cone tree (per-frame cone hierarchy; every hit of a ray is found, the lowest list position wins)"
                          if args.search == "tree" else "list scan (every front-facing triangle in list order until the ray's first hit)",
                "tiles": "32x32 cyclic over ranks, own pixels gathered on rank 0" if world > 1 else "single GPU, whole frame",
            },
            "verified_against_single_gpu_frame": verified,
            "rays_per_step": int(total_rays / args.steps),
            "rays_per_step_rank0": {"primary": totals["samples"] // args.steps, "reflection": totals["refl"] // args.steps,
                                    "shadow": totals["shadow"] // args.steps,
                                    "shadow_elided": totals["elided"] // args.steps,
                                    "note": "ray counts follow the reference (the oracle's counters); 'shadow_elided' of the "
                                            "shadow rays start at hits on material-less objects and are not traced because "
                                            "their outcome cannot change the pixel"},
            "tests_per_s": tests_per_s,
            "tests_note": "(ray, triangle) pairs actually evaluated by a kernel lane per second over the timed region; pairs "
                          "skipped by the origin-side cull, the cone tree or a ray's early exit are not counted",
            "tests_per_s_frac_of_fp32_roofline": tests_per_s * FLOPS_PER_TEST / (PEAK_FP32_TFLOPS * 1e12 * world),
            "search_rank0_per_step": {"ms_tri_search": round(totals["ms_primary"] / args.steps, 4),
                                      "ms_shade": round(totals["ms_shade"] / args.steps, 4),
                                      "tri_tests": totals["tests_primary"] // args.steps,
                                      "node_tests": totals["node_tests"] // args.steps,
                                      "candidates": totals["candidates"] // args.steps,
                                      "fp64_decisions": totals["exact"] // args.steps,
                                      "primary_hits": totals["hits"] // args.steps},
            "device_ms_per_step_rank0": round(totals["ms_total"] / args.steps, 4),
            "roofline": roofline,
        }
        if scan_leg is not None:
            out["list_scan"] = scan_leg
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(data, params, args.cpu_threads)
        print(json.dumps(out), flush=True)

    for sc_ in scenes_in_flight:
        sc_.close()
    if world > 1:
        dist.destroy_process_group()


def traffic_from_profile(config, kernel):
    """HBM bytes per launch of `kernel` from the committed PMC profile of this configuration
    (bench.py itself cannot collect PMC counters); None if there is none for `config`."""
    name = {"k_search_primary": "r01_pmc_search_kernel.json", "k_shade_candidates": "r01_pmc_shade_kernel.json"}[kernel]
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
    except OSError:
        return None
    return d.get("hbm_bytes_per_launch") if d.get("config") == config else None


def scan_roofline(tests_per_launch, avg_ms, config):
    """BASELINE's figure for the list-scan kernel: evaluated pairs x 69 algorithmic flops / launch time."""
    achieved = tests_per_launch * FLOPS_PER_TEST / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
    return {
        "bound": "valu_fp32",
        "kernel": "k_search_primary",
        "achieved": round(achieved, 3),
        "peak": PEAK_FP32_TFLOPS,
        "unit": "TFLOP/s",
        "frac": round(achieved / PEAK_FP32_TFLOPS, 4),
        "traffic": traffic_from_profile(config, "k_search_primary"),
        "traffic_unit": "bytes of HBM traffic per launch (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, "
                        "separate passes; profiles/r01_pmc_search_kernel.json)",
        "flops_per_test_algorithmic": FLOPS_PER_TEST,
        "tests_per_launch": tests_per_launch,
        "avg_launch_ms": round(avg_ms, 4),
        "executed_fp32_flops_per_test": 7,
        "note": "achieved = evaluated (ray, triangle) pairs per launch x 69 algorithmic flops / launch time "
                "(HIP events on the library's stream).  The kernel decides most pairs with a conservative "
                "3-FMA cone test (about 7 executed FP32 flops per pair incl. the max), the rest with a 9-FMA "
                "edge test and an FP64 reference-order test, so frac exceeds 1 by construction; the VALU "
                "is 88% busy during the launch (SQ_ACTIVE_INST_VALU, profiles/).",
    }


def list_scan_leg(scene, cam, params, frame, steps, config):
    """The same frame with RTMI_SEARCH_LIST_SCAN, after the timed region: the brute-force figure BASELINE's
    metric names (ray-triangle tests/s against the FP32 peak) for the kernel that performs it."""
    import copy
    import torch
    from raytracer_amd import _abi
    p = copy.copy(params)
    p.search = _abi.RTMI_SEARCH_LIST_SCAN
    scene.render_device(cam, p, frame.data_ptr())  # warm-up: packs the list-scan operands
    tests = ms = ms_total = rays = 0.0
    launches = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        c = scene.render_device(cam, p, frame.data_ptr())
        tests += c["tri_tests_primary_kernel"]
        ms += c["ms_primary_kernel"]
        ms_total += c["ms_total"]
        launches += c["tri_search_launches"]
        rays += c["samples"] + c["rays_reflection"] + c["rays_shadow"]
    dt = time.perf_counter() - t0
    launches = max(launches, 1)
    r = scan_roofline(tests / launches, ms / launches, config)
    r.update({"steps": steps, "ms_per_step": round(dt / steps * 1e3, 4), "mrays_per_s": round(rays / dt / 1e6, 3),
              "tests_per_s": tests / dt,
              "tests_per_s_frac_of_fp32_roofline": tests / dt * FLOPS_PER_TEST / (PEAK_FP32_TFLOPS * 1e12)})
    return r


def scene_spp(sub):
    import raytracer_amd as rt
    return rt.samples_per_pixel(sub)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip() + f" ({os.cpu_count()} logical CPUs visible)"
    except OSError:
        pass
    return "unknown"


def cpu_baseline(data, params, threads):
    """The CPU oracle (FP64 restatement of the reference) on a bounded sample of the same workload:
    a centre window of the same frame, same scene, same spp."""
    import oracle
    from raytracer_amd import scenes
    W, H = params.width, params.height
    ww, wh = min(W - 1, 1024), min(H - 1, 512)
    x0, y0 = (W - 1 - ww) // 2, (H - 1 - wh) // 2
    win = (x0, y0, x0 + ww, y0 + wh)
    p = scenes.make_params(W, H, params.pixel_subdivision, tuple(params.ambient), window=win)
    cam = scenes.make_camera(**scenes.MESH_CAMERA)
    t0 = time.perf_counter()
    _, _, cnt = oracle.render(data, cam, p, want_hit_ids=False, threads=threads)
    dt = time.perf_counter() - t0
    rays = cnt["samples"] + cnt["rays_reflection"] + cnt["rays_shadow"]
    # the reference itself is single-threaded (ViewController.swift:154): one thread on a twenty-fifth of the window
    sw, sh = max(ww // 5, 1), max(wh // 5, 1)
    sx, sy = (W - 1 - sw) // 2, (H - 1 - sh) // 2
    p1 = scenes.make_params(W, H, params.pixel_subdivision, tuple(params.ambient), window=(sx, sy, sx + sw, sy + sh))
    t1 = time.perf_counter()
    _, _, c1 = oracle.render(data, cam, p1, want_hit_ids=False, threads=1)
    d1 = time.perf_counter() - t1
    rays1 = c1["samples"] + c1["rays_reflection"] + c1["rays_shadow"]
    return {
        "value": round(rays / dt / 1e6, 4),
        "unit": "Mrays/s",
        "cores": threads,
        "cpu_model": cpu_model(),
        "kind": "port",
        "sample": f"centre window {ww}x{wh} px of the same frame ({cnt['samples']} primary rays, "
                  f"{cnt['tri_tests']} ray-triangle tests, {dt:.1f} s); the window lies inside the mesh silhouette, "
                  "so its rays are costlier than the frame average",
        "tests_per_s": cnt["tri_tests"] / dt,
        "seconds": round(dt, 2),
        "single_thread": {"value": round(rays1 / d1 / 1e6, 4), "unit": "Mrays/s", "cores": 1,
                          "sample": f"centre window {sw}x{sh} px ({c1['samples']} primary rays, {d1:.1f} s)",
                          "tests_per_s": c1["tri_tests"] / d1},
    }


if __name__ == "__main__":
    main()
