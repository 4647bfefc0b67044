#!/usr/bin/env python3
"""Benchmark of the render hot path on MI355X.

A step is one full frame of the headline configuration: the stand-in "pig" mesh (5,120-triangle displaced icosphere
wrapped in Sphere(boundingObjects:), i.e. form B, what the reference's loadObjFile builds), 3840x2160, 16 samples per
pixel (pixelSubdivision 0.25), 3 lights, reference-exact list-order hit policy, the camera moving every step on the
orbit of config 5 (--static-camera holds it still).  Timed frames ask for pixels only (RTMI_FLAG_NO_RAY_STATISTICS):
rays that cannot change a pixel are neither traced nor counted.

  python bench.py --gpus N            C3: with N > 1 the frame's 32x32 tiles are dealt cyclically over the ranks (strong
                                      scaling: total work per step is fixed), the scene is broadcast once over RCCL before
                                      the timed region and every rank's own pixels are gathered on rank 0 inside it
  python bench.py --gpus N --config C5   a batch of orbit frames: one WHOLE frame per step, frame i on rank i mod N, two in
                                      flight per rank, no collective in the timed region (digests gathered after it)

Without a launcher around it (`WORLD_SIZE` unset) `--gpus N` starts its N ranks itself: torch.distributed.run as a
child process, before this process has touched a GPU.

`value` = rays actually traced per second (primary rays that were generated and searched + reflection and shadow rays
that were traced); the count of rays the REFERENCE traces for the same frames is reported beside it
(`reference_rays`), taken from an extra frame with full ray statistics after the timed region.

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

FLOPS_PER_TEST = 69            # the reference's triangle test with ray-independent terms hoisted (algorithmic count)
PEAK_FP32_TFLOPS = 157.3       # MI355X vector FP32, packed (256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz)
PEAK_FP64_TFLOPS = 78.6        # MI355X vector FP64 (256 CU x 4 SIMD x 32 FLOP/clk x 2.4 GHz)
PEAK_HBM_GBPS = 8000.0         # MI355X HBM3E
SIMDS = 1024                   # 256 CUs x 4
PMC_FILE = os.path.join(ROOT, "profiles", "r03_pmc_kernels.json")

CONFIGS = {
    # name: (width, height, subdivision, mesh level, n_lights)
    "C2": (1024, 1024, 0.5, 4, 1),
    "C3": (3840, 2160, 0.25, 4, 3),
    "C4": (1920, 1080, 0.5, 8, 1),
    "C5": (3840, 2160, 0.25, 4, 3),   # the frames of C3, camera on the 64-step orbit, one whole frame per step
}
TINY = (96, 64, 0.5, 2)  # --tiny: frame and mesh of the logic checks (tests), in place of the configuration's


def config_dims(args):
    W, H, sub, level, n_lights = CONFIGS[args.config]
    if args.tiny:
        W, H, sub, level = TINY
    return W, H, sub, level, n_lights


def shade_flops_per_hit(n_lights, inert, n_sphere_like=1):
    """FP64 operations of calculateLighting (RayTracingFunctions.swift:182-271) for one triangle hit with a material
    in a scene whose other objects are `n_sphere_like` material-less bounding spheres (form B); +, -, *, /, sqrt and
    pow count 1 each, a normalize is 9 (dot 5, sqrt, 3 divisions).
    Always: hit point 6, ambient 3, surfaceToCamera 12, and per light 57 (direction 12, n.l 5, diffuse 9, reflection of
    the light 16, r.v 5, pow 1, specular 9).
    Only when the secondary rays are traced (inert = False; the reference always does): reflection direction 12, the
    reflection ray's sphere tests 22 each + 13 for the hit it finds, the unlit reflected hit's light loop 26 per light."""
    always = 6 + 3 + 12 + 57 * n_lights
    secondary = 12 + 22 * n_sphere_like + 13 + 26 * n_lights
    return always if inert else always + secondary


def pmc_kernel(config, kernel):
    """Counters of `kernel` per launch from the committed PMC passes of this configuration (bench.py itself cannot
    collect PMC counters); None if there are none."""
    try:
        d = json.load(open(PMC_FILE))
    except OSError:
        return None
    return d.get(config, {}).get(kernel)


def pmc_fractions(k, live_ms=None):
    """What the hardware executed, from the counters of one launch (formulas spelled out in the result):
    executed FP64 (or FP32) flops per second against the peak, and the share of cycles the VALU issued.
    live_ms: this run's average duration of the same kernel; the result then says how far the profiled launch is from
    it (`launch_ms_drift_frac`) and calls itself stale above 10 %."""
    if not k:
        return None
    ms = k["avg_launch_ms"]
    out = {"replayed_from": os.path.relpath(PMC_FILE, ROOT), "profiled_commit": k.get("commit"), "source": k.get("source"),
           "avg_launch_ms_profiled": ms,
           "replay_note": "counter values of a separate rocprofv3 --pmc run of this configuration, committed with the code; NOT "
                          "measured in this run (only achieved/frac/avg_launch_ms are)"}
    f64 = 64 * (k.get("SQ_INSTS_VALU_ADD_F64", 0) + k.get("SQ_INSTS_VALU_MUL_F64", 0) + 2 * k.get("SQ_INSTS_VALU_FMA_F64", 0)
                + k.get("SQ_INSTS_VALU_TRANS_F64", 0))
    f32 = 64 * (k.get("SQ_INSTS_VALU_ADD_F32", 0) + k.get("SQ_INSTS_VALU_MUL_F32", 0) + 2 * k.get("SQ_INSTS_VALU_FMA_F32", 0))
    if f64:
        out["executed_fp64_tflops"] = round(f64 / (ms * 1e-3) / 1e12, 3)
        out["executed_fp64_frac_of_peak"] = round(f64 / (ms * 1e-3) / 1e12 / PEAK_FP64_TFLOPS, 4)
        out["executed_fp64_formula"] = "64 lanes x (ADD_F64 + MUL_F64 + 2 FMA_F64 + TRANS_F64 wave-instructions) / launch time / 78.6 TFLOP/s"
    if f32:
        out["executed_fp32_tflops_if_unpacked"] = round(f32 / (ms * 1e-3) / 1e12, 3)
        out["executed_fp32_frac_of_peak_if_unpacked"] = round(f32 / (ms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS, 4)
        out["executed_fp32_formula"] = ("64 lanes x (ADD_F32 + MUL_F32 + 2 FMA_F32 wave-instructions) / launch time / 157.3 TFLOP/s; the "
                                        "counters tally a packed v_pk_* instruction once, so a loop of packed instructions executes up "
                                        "to twice this")
    if k.get("SQ_ACTIVE_INST_VALU") and k.get("GRBM_GUI_ACTIVE"):
        out["valu_issue_frac"] = round(k["SQ_ACTIVE_INST_VALU"] * 4 / (k["GRBM_GUI_ACTIVE"] / 8 * SIMDS), 4)
        out["valu_issue_formula"] = "SQ_ACTIVE_INST_VALU x 4 cycles / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)"
    if k.get("hbm_bytes_per_launch"):
        out["hbm_bytes_per_launch"] = k["hbm_bytes_per_launch"]
    if live_ms and live_ms > 0:
        out["launch_ms_drift_frac"] = round(abs(live_ms - ms) / ms, 4)
        out["stale"] = out["launch_ms_drift_frac"] > 0.10
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 20; 64 for --config C5)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps before them (default 3; 4 for --config C5)")
    ap.add_argument("--config", default="C3", choices=sorted(CONFIGS),
                    help="C3 (default): one 4K frame per step, 32x32 tiles dealt over the ranks, own pixels gathered on rank 0.  "
                         "C5: a batch of orbit frames, one whole frame per step, frame i on rank i mod N, no per-frame collective.")
    ap.add_argument("--form", default="B", choices=["B", "T"])
    ap.add_argument("--search", default="tree", choices=["tree", "scan"],
                    help="triangle search of the timed region: the cone tree (default) or the list scan")
    ap.add_argument("--static-camera", action="store_true", help="hold the camera still (default: it moves every step)")
    ap.add_argument("--orbit", action="store_true", help="(default now) move the camera every step")
    ap.add_argument("--ray-stats", action="store_true",
                    help="keep the reference's ray statistics in the timed frames (every ray the reference would trace is "
                         "counted, which costs the tracing of rays that cannot change a pixel)")
    ap.add_argument("--no-aux", action="store_true",
                    help="skip the legs after the timed region (reference ray counts, static camera, list scan, form T)")
    ap.add_argument("--no-scan-leg", action="store_true", help="skip the list-scan leg only")
    ap.add_argument("--no-overlap-leg", action="store_true",
                    help="skip the two-frames-in-flight leg only (profiling: overlapping frames stretch each other's kernel durations)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--no-verify", action="store_true",
                    help="C3, N > 1: do not re-render the last frame on rank 0 alone to compare it byte for byte.  "
                         "C5: do not digest the frames after the timed region")
    ap.add_argument("--verify", action="store_true", help="(default)")
    ap.add_argument("--deliver", default="keep", choices=["keep", "rank0"],
                    help="C5: finished frames stay in the HBM of the rank that rendered them and only their digests travel "
                         "(keep, default), or every finished frame is sent whole to rank 0 (rank0)")
    ap.add_argument("--obj", default=os.environ.get("RTMI_PIG_OBJ"),
                    help="a triangulated, vertex-index-only OBJ file to render instead of the stand-in mesh (default: $RTMI_PIG_OBJ)")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="multi-rank logic check on a box with ONE GPU: all ranks render on device 0 and the "
                         "collectives run over gloo on host tensors (not a performance mode)")
    ap.add_argument("--tiny", action="store_true", help="logic checks: a 96x64 frame of a 320-triangle mesh instead of the configuration's")
    ap.add_argument("--emulated-cpu", action="store_true",
                    help="logic check without any GPU (tests): the CPU-emulated build of the kernels (tests/emu) stands in "
                         "for librtmi.so, buffers are host tensors, collectives run over gloo (not a performance mode)")
    args = ap.parse_args(argv)
    if args.steps is None:
        args.steps = 64 if args.config == "C5" else 20
    if args.warmup is None:
        args.warmup = 4 if args.config == "C5" else 3
    return args


def self_launch(argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks with torch.distributed.run as a child process and
    pass its output and exit status on.  This process has not touched a GPU (torch is not even imported here)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    n = parse_args(argv).gpus
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


class Runtime:
    """What the two step loops share: the ranks, the device (or the host, for the logic checks), fences."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.args = torch, dist, args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = 0 if (args.rehearse_gloo or args.emulated_cpu) else int(os.environ.get("LOCAL_RANK", "0"))
        self.on_gpu = not args.emulated_cpu
        if self.on_gpu:
            torch.cuda.set_device(self.local_rank)
            self.dev = torch.device("cuda", self.local_rank)
        else:
            self.dev = torch.device("cpu")
        host_comm = args.rehearse_gloo or args.emulated_cpu
        self.comm_dev = torch.device("cpu") if host_comm else self.dev
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if host_comm:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            else:
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.dev)

    def sync(self):
        if self.on_gpu:
            self.torch.cuda.synchronize()

    def fence(self):
        self.sync()
        if self.world > 1:
            self.dist.barrier()
        self.sync()

    def close(self):
        if self.world > 1:
            self.dist.destroy_process_group()


def build_scene_data(args, rt, scenes, level, n_lights):
    """The scene of the configuration on rank 0: the stand-in mesh, or the OBJ file given with --obj / RTMI_PIG_OBJ (read by
    the library's own loader: readObjFile + createTriangleList) under the same wrapper, lights and material."""
    if args.obj and args.config != "C4":
        tris, _, _, _ = rt.load_obj(args.obj)
        tris["material"] = 0
        data = scenes.mesh_scene(0, args.form, n_lights, bounding_sphere=rt.bounding_sphere)
        data.triangles = tris
        if args.form == "B":
            c, r = rt.bounding_sphere(tris)
            data.objects = scenes.sphere_object(c, r, -1, bounding=True, first=0, count=len(tris))
        else:
            data.objects = scenes.triangles_object(0, len(tris))
        return data, f"OBJ file {os.path.basename(args.obj)} ({len(tris)} triangles)"
    return scenes.mesh_scene(level, args.form, n_lights, bounding_sphere=rt.bounding_sphere), "stand-in (no pig OBJ available offline)"


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(argv))
    if args.emulated_cpu:
        os.environ.setdefault("RTMI_LIB_PATH", os.path.join(ROOT, "tests", "emu", "build_ns", "librtmi_emu.so"))
    rtm = Runtime(args)
    if rtm.world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {rtm.world} ranks")
    try:
        if args.config == "C5":
            run_orbit_batch(args, rtm)
        else:
            run_tiled_frames(args, rtm)
    finally:
        rtm.close()


def run_tiled_frames(args, rtm):
    """C2/C3/C4: a step is one frame; with N > 1 its 32x32 tiles are dealt cyclically over the ranks."""
    torch, dist = rtm.torch, rtm.dist
    import raytracer_amd as rt
    from raytracer_amd import _abi, scenes, sharding

    world, rank, local_rank, dev, comm_dev = rtm.world, rtm.rank, rtm.local_rank, rtm.dev, rtm.comm_dev
    W, H, sub, level, n_lights = config_dims(args)
    data, mesh_note = build_scene_data(args, rt, scenes, level, n_lights) if rank == 0 else (None, "")
    if world > 1:
        data = sharding.broadcast_scene(data, 0, device=comm_dev)  # RCCL byte broadcast, once
    scene = rt.Scene(data, device=local_rank)
    search = _abi.RTMI_SEARCH_CONE_TREE if args.search == "tree" else _abi.RTMI_SEARCH_LIST_SCAN
    flags = 0 if args.ray_stats else _abi.RTMI_FLAG_NO_RAY_STATISTICS
    params = scenes.make_params(W, H, sub, search=search, flags=flags)
    my_params = sharding.tile_params(params, rank, world) if world > 1 else params
    # the composer's frames are what the library renders into: device memory, except in the two logic-check modes
    composer = sharding.TileComposer(W, H, rank, world, comm_dev)
    staging = [torch.zeros((H, W, 4), dtype=torch.uint8, device=dev) for _ in range(2)] if args.rehearse_gloo else None
    rtm.sync()
    moving = not args.static_camera

    def camera_for(step):
        if moving:
            return scenes.make_camera(**scenes.orbit_camera(step % 64))
        return scenes.make_camera(**scenes.MESH_CAMERA)

    keys = ("samples", "primary_rays_traced", "rays_reflection", "rays_shadow", "shadow_rays_elided", "tri_tests",
            "tri_tests_primary_kernel", "ms_primary_kernel", "ms_total", "tri_search_launches", "tri_exact_tests", "primary_hits",
            "ms_shade", "node_tests", "tri_candidates", "hits_decided_fp32", "pixels_fixed_fp64", "tri_edge_tests")
    totals = {k: 0 for k in keys}
    work_bytes = [0]
    host = {"enqueue_s": 0.0, "wait_s": 0.0, "compose_s": 0.0}

    # With more than one rank a frame is short (1/N of the work) and the gaps between frames start to show: each
    # rank then keeps TWO frames in flight, on two scene objects with a stream and work buffers of their own.
    # Frame i is enqueued before frame i-1 is waited for; the gather of frame i-1 then overlaps the rendering of frame i.
    scenes_in_flight = [scene] + ([rt.Scene(data, device=local_rank)] if world > 1 else [])
    depth = len(scenes_in_flight)
    in_flight = []  # (scene, slot, record) of frames enqueued and not yet waited for

    def retire_oldest():
        sc_, slot_, record_ = in_flight.pop(0)
        t0 = time.perf_counter()
        cnt = sc_.render_end()
        t1 = time.perf_counter()
        if args.rehearse_gloo:
            composer.frames[slot_].copy_(staging[slot_])
        composer.compose(slot_)  # travels to rank 0 while the next frame renders
        t2 = time.perf_counter()
        if record_:
            for k in keys:
                totals[k] += cnt[k]
            work_bytes[0] = max(work_bytes[0], cnt["work_buffer_bytes"])
            host["wait_s"] += t1 - t0
            host["compose_s"] += t2 - t1

    def step(i, record):
        # Frames alternate between two buffers.  Frame i reuses the buffer of frame i-2: that frame has been waited
        # for and its gather enqueued (retire_oldest); the scene's stream is ordered behind that gather's last touch of
        # the buffer with an event (no host wait), and frame i-1 keeps the GPU busy meanwhile.
        slot = i & 1
        if len(in_flight) == depth:
            retire_oldest()
        t0 = time.perf_counter()
        sc_ = scenes_in_flight[i % depth]
        ev = composer.slot_event(slot)
        if ev is not None:
            sc_.wait_event(ev.cuda_event)
        target = staging[slot] if args.rehearse_gloo else composer.frames[slot]
        sc_.render_begin(camera_for(i), my_params, target.data_ptr())   # enqueue frame i, do not wait
        in_flight.append((sc_, slot, record))
        if record:
            host["enqueue_s"] += time.perf_counter() - t0

    def flush():
        while in_flight:
            retire_oldest()

    for i in range(args.warmup):
        step(i, False)
    flush()

    fence = rtm.fence
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i, True)
    flush()  # the last frame's gather belongs to the timed region
    fence()
    elapsed = time.perf_counter() - t0

    verified = None
    if world > 1 and not args.no_verify and rank == 0:
        alone = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev)
        rtm.sync()
        scene.render_device(camera_for(args.warmup + args.steps - 1), params, alone.data_ptr())
        last = composer.frames[(args.warmup + args.steps - 1) & 1]
        verified = bool((alone.cpu() == last.cpu()).all())
        assert verified, "the composed multi-rank frame differs from the single-GPU frame"

    last_frame_sha256 = None
    if rank == 0 and not args.no_verify:  # after the timed region: a digest of the last composed frame
        import hashlib
        rtm.sync()
        last_frame_sha256 = hashlib.sha256(composer.frames[(args.warmup + args.steps - 1) & 1].cpu().numpy().tobytes()).hexdigest()
    traced = totals["primary_rays_traced"] + totals["rays_reflection"] + totals["rays_shadow"] - totals["shadow_rays_elided"]
    mine = [float(traced), float(totals["tri_tests"]), elapsed, totals["ms_total"] / args.steps,
            host["enqueue_s"] / args.steps * 1e3, host["wait_s"] / args.steps * 1e3, host["compose_s"] / args.steps * 1e3]
    agg = torch.tensor(mine, dtype=torch.float64, device=comm_dev)
    if world > 1:
        every = [torch.zeros_like(agg) for _ in range(world)]
        dist.all_gather(every, agg)
        per_rank = [e.cpu().tolist() for e in every]
    else:
        per_rank = [mine]
    total_traced = sum(r[0] for r in per_rank)
    total_tests = sum(r[1] for r in per_rank)
    elapsed_max = max(r[2] for r in per_rank)

    aux = {}
    if rank == 0 and world == 1 and not args.no_aux and rtm.on_gpu:
        aux = aux_legs(rt, scene, data, params, composer.frames[0], camera_for, args, level, n_lights)

    if rank == 0:
        steps = args.steps
        ms_step = elapsed_max / steps * 1e3
        ref = aux.get("reference_counts")
        inert = args.form == "B" and not args.ray_stats and args.search == "tree"
        fast = totals["hits_decided_fp32"] > 0  # the path without a ray-generation launch ran (kernels_fast.h)
        if args.search != "tree":
            roofline = scan_roofline(totals["tri_tests_primary_kernel"] / max(totals["tri_search_launches"], 1),
                                     totals["ms_primary_kernel"] / max(totals["tri_search_launches"], 1), args.config)
        elif fast:
            # the two large kernels of the frame are close: the roofline is that of the one that took longer in this run,
            # the other one's figures ride along
            pair = sorted([walk_roofline(args, totals, steps), shade32_roofline(args, totals, steps, n_lights)],
                          key=lambda r: -r["avg_launch_ms"])
            roofline = pair[0]
            roofline["second_kernel"] = {k: pair[1][k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_ms",
                                                                  "traffic", "algorithmic_bytes_per_launch", "executed", "note")}
        else:
            roofline = shade_roofline(args, totals, steps, n_lights, inert)
        out = {
            "metric": "Mrays/s + ray-tri tests/s (% FP32 peak), pig OBJ 4K 16spp, 1/2/4/8 GPU",
            "value": round(total_traced / elapsed_max / 1e6, 3),
            "unit": "Mrays/s",
            "value_note": "rays actually traced per second: primary rays that were generated and went on to the triangle search, "
                          "plus the reflection and shadow rays that were traced.  Samples cleared beforehand by a "
                          "conservative test (whole 8x8 blocks, then single samples against the bounding sphere's cone) and rays "
                          "whose outcome cannot change a pixel are not counted.",
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "dtype_note": "the reference computes in FP64 and so does every decision here that FP32 cannot make with certainty: FP32 "
                          "filters nominate (ray, triangle) pairs, FP32 margins accept the pairs that are inside triangle and bounding "
                          "sphere beyond the FP32 error, the others are decided in FP64 in the reference's operation order; colours "
                          "are FP32 with an error bound, and the pixels whose byte that bound cannot settle are shaded again in FP64 "
                          "(see fp32_first)",
            "data": "synthetic",
            "config": {
                "workload": f"{args.config}: pig mesh [{mesh_note}; default = displaced icosphere level {level}], "
                            f"{len(data.triangles)} triangles, {W}x{H}, {scene_spp(sub)} spp, {n_lights} lights, "
                            f"form {args.form} ({'bounded by Sphere(boundingObjects:)' if args.form == 'B' else 'top-level triangles'}), "
                            "list-order hit policy, " + ("camera moving every frame (orbit of config 5)" if moving else "camera held still")
                            + (", full ray statistics" if args.ray_stats else ", pixels only (no ray statistics)")
                            + (" -- TINY logic-check frame" if args.tiny else ""),
                "mesh": mesh_note,
                "search": "cone tree (per-frame cone hierarchy; every hit of a ray is found, the lowest list position wins)"
                          if args.search == "tree" else "list scan (every front-facing triangle in list order until the ray's first hit)",
                "tiles": "32x32 cyclic over ranks, own pixels gathered on rank 0" if world > 1 else "single GPU, whole frame",
            },
            "verified_against_single_gpu_frame": verified,
            "last_frame_sha256": last_frame_sha256,
            "traced_rays_per_step": int(total_traced / steps),
            "traced_rays_per_step_rank0": {"primary": totals["primary_rays_traced"] // steps,
                                           "primary_cleared_untraced": (totals["samples"] - totals["primary_rays_traced"]) // steps,
                                           "reflection": totals["rays_reflection"] // steps,
                                           "shadow": (totals["rays_shadow"] - totals["shadow_rays_elided"]) // steps},
            "reference_rays": None if not ref else {
                "per_step": ref["rays"],
                "mrays_per_s_equivalent": round(ref["rays"] / (ms_step * 1e-3) / 1e6, 1),
                "breakdown": ref["breakdown"],
                "note": "rays the REFERENCE traces for one such frame (primary + reflection + shadow; equal to the oracle's "
                        "counters), from an extra frame with full ray statistics after the timed region, divided by the timed "
                        "frame time.  Most of them are never traced here: primary rays that a conservative test clears, and "
                        "on this scene every reflection and shadow ray (nothing they could hit changes a pixel).",
                "ms_per_step_with_full_ray_statistics": ref["ms"]},
            "tests_per_s": total_tests / elapsed_max,
            "tests_note": "(ray, triangle) pairs actually evaluated by a kernel lane per second over the timed region; pairs "
                          "skipped by the origin-side cull, the cone tree or a ray's early exit are not counted",
            "algorithmic_69flop_equiv_tflops": round(total_tests / elapsed_max * FLOPS_PER_TEST / 1e12, 3),
            "algorithmic_69flop_equiv_note": "evaluated pairs per second x the 69 flops of the reference's triangle test: NOT an "
                                            "executed-flop rate (the filter cascade executes about 7 FP32 flops per pair)",
            "search_rank0_per_step": {"ms_tri_search": round(totals["ms_primary_kernel"] / steps, 4),
                                      "ms_shade": round(totals["ms_shade"] / steps, 4),
                                      "tri_tests": totals["tri_tests_primary_kernel"] // steps,
                                      "node_tests": totals["node_tests"] // steps,
                                      "candidates": totals["tri_candidates"] // steps,
                                      "fp64_decisions": totals["tri_exact_tests"] // steps,
                                      "primary_hits": totals["primary_hits"] // steps},
            "fp32_first": None if not fast else {
                "hits_decided_by_fp32_margins": totals["hits_decided_fp32"] // steps,
                "pairs_decided_in_fp64": totals["tri_exact_tests"] // steps,
                "share_of_hits_decided_in_fp32": round(totals["hits_decided_fp32"] / max(totals["primary_hits"], 1), 5),
                "pixels_shaded_again_in_fp64": totals["pixels_fixed_fp64"] // steps,
                "share_of_hit_pixels_shaded_again": round(totals["pixels_fixed_fp64"] / max(totals["primary_hits"] / scene_spp(sub), 1), 5),
                "note": "rank 0, per step.  A hit is accepted by FP32 alone when the ray lies inside the triangle's three edge planes, "
                        "in front of its plane and inside the bounding sphere by margins above the FP32 error (bound on the "
                        "directions: rtmi_direction_error_bound); a pixel is shaded again when a sample's light faces the surface "
                        "within the error, or 255 x mean +- the summed error bound of its samples holds a whole number"},
            "work_buffer_bytes_rank0": work_bytes[0],
            "device_ms_per_step_rank0": round(totals["ms_total"] / steps, 4),
            "per_rank": [{"rank": r, "device_ms_per_step": round(v[3], 4), "wall_ms_per_step": round(v[2] / steps * 1e3, 4),
                          "host_enqueue_ms": round(v[4], 4), "host_wait_ms": round(v[5], 4), "host_compose_enqueue_ms": round(v[6], 4)}
                         for r, v in enumerate(per_rank)],
            "roofline": roofline,
        }
        for k in ("static_camera", "two_frames_in_flight", "list_scan", "form_T", "shadows_enter_bounded"):
            if k in aux:
                out[k] = aux[k]
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(data, params, args.cpu_threads, args.config, moving)
        print(json.dumps(out), flush=True)

    for sc_ in scenes_in_flight:
        sc_.close()



def run_orbit_batch(args, rtm):
    """C5: a batch of orbit frames (the frame of C3, camera i of 64 on a circle about the y axis).  A step is one WHOLE
    frame; frame i is rendered by rank i mod N into a buffer of its own, two frames in flight per rank (two scene
    objects, each with its stream and work buffers).  No collective inside the timed region unless --deliver rank0
    asks for the frames on rank 0: the scene is broadcast once before it, digests are gathered after it."""
    import hashlib
    torch, dist = rtm.torch, rtm.dist
    import raytracer_amd as rt
    from raytracer_amd import _abi, scenes, sharding

    world, rank, local_rank, dev, comm_dev = rtm.world, rtm.rank, rtm.local_rank, rtm.dev, rtm.comm_dev
    W, H, sub, level, n_lights = config_dims(args)
    data, mesh_note = build_scene_data(args, rt, scenes, level, n_lights) if rank == 0 else (None, "")
    if world > 1:
        data = sharding.broadcast_scene(data, 0, device=comm_dev)
    search = _abi.RTMI_SEARCH_CONE_TREE if args.search == "tree" else _abi.RTMI_SEARCH_LIST_SCAN
    flags = 0 if args.ray_stats else _abi.RTMI_FLAG_NO_RAY_STATISTICS
    params = scenes.make_params(W, H, sub, search=search, flags=flags)
    K, n_orbit = args.steps, 64
    mine = [i for i in range(K) if i % world == rank]
    depth = 2
    renderers = [rt.Scene(data, device=local_rank) for _ in range(depth)]
    # every frame of the batch stays resident where it was rendered (33 MB each at 4K); beyond 64 per rank the oldest are reused
    ring = max(depth, min(len(mine), 64))
    out = torch.zeros((ring, H, W, 4), dtype=torch.uint8, device=dev)
    spare = torch.zeros((depth, H, W, 4), dtype=torch.uint8, device=dev)  # warm-up frames
    to_rank0 = args.deliver == "rank0" and world > 1
    gathered = torch.zeros((K, H, W, 4), dtype=torch.uint8, device=dev) if (to_rank0 and rank == 0) else None
    host_stage = rtm.args.rehearse_gloo  # one-GPU rehearsal: the collectives run on host copies

    def camera_for(i):
        return scenes.make_camera(**scenes.orbit_camera(i % n_orbit, n_orbit))

    keys = ("samples", "primary_rays_traced", "rays_reflection", "rays_shadow", "shadow_rays_elided", "tri_tests", "primary_hits",
            "ms_total", "ms_shade", "ms_primary_kernel")
    totals = {k: 0 for k in keys}
    host = {"enqueue_s": 0.0, "wait_s": 0.0, "deliver_s": 0.0}
    in_flight, works, keep_alive = [], [], []

    def deliver(i, buf):
        # frame i has been waited for.  Peers send it whole to rank 0; rank 0 posts the receives of the round that its
        # own frame belongs to (one frame per peer, each over its own link) and copies its own frame on the device.
        if rank != 0:
            t = buf.cpu() if host_stage else buf
            keep_alive.append(t)
            works.extend(dist.batch_isend_irecv([dist.P2POp(dist.isend, t, 0)]))
            return
        gathered[i].copy_(buf, non_blocking=True)
        ops = []
        for r in range(1, world):
            if i + r < K:
                t = torch.empty((H, W, 4), dtype=torch.uint8) if host_stage else gathered[i + r]
                if host_stage:
                    keep_alive.append((i + r, t))
                ops.append(dist.P2POp(dist.irecv, t, r))
        if ops:
            works.extend(dist.batch_isend_irecv(ops))

    def retire_oldest():
        sc_, i, buf, record = in_flight.pop(0)
        t0 = time.perf_counter()
        cnt = sc_.render_end()
        t1 = time.perf_counter()
        if record and to_rank0:
            deliver(i, buf)
        t2 = time.perf_counter()
        if record:
            for k in keys:
                totals[k] += cnt[k]
            host["wait_s"] += t1 - t0
            host["deliver_s"] += t2 - t1

    def enqueue(j, i, buf, record):
        if len(in_flight) == depth:
            retire_oldest()
        t0 = time.perf_counter()
        sc_ = renderers[j % depth]
        sc_.render_begin(camera_for(i), params, buf.data_ptr())
        in_flight.append((sc_, i, buf, record))
        if record:
            host["enqueue_s"] += time.perf_counter() - t0

    def flush():
        while in_flight:
            retire_oldest()

    for j in range(args.warmup):  # every rank warms its own two renderers (buffers, code objects)
        enqueue(j, n_orbit - args.warmup + j, spare[j % depth], False)
    flush()
    rtm.fence()
    t0 = time.perf_counter()
    for j, i in enumerate(mine):
        enqueue(j, i, out[j % ring], True)
    flush()
    for w in works:
        w.wait()
    if host_stage and rank == 0 and to_rank0:
        for i, t in keep_alive:
            gathered[i].copy_(t)
    rtm.fence()
    elapsed = time.perf_counter() - t0

    # after the timed region: a digest of every frame that is still resident, from the rank that rendered it
    digests = []
    if not args.no_verify:
        for j, i in enumerate(mine):
            if j + ring < len(mine):
                continue  # its buffer has been reused
            f = out[j % ring]
            digests.append({"frame": i, "rank": rank, "camera": i % n_orbit,
                            "checksum": int(f.view(torch.int32).sum(dtype=torch.int64).item()),
                            "sha256": hashlib.sha256(f.cpu().numpy().tobytes()).hexdigest()})
    traced = totals["primary_rays_traced"] + totals["rays_reflection"] + totals["rays_shadow"] - totals["shadow_rays_elided"]
    nf = max(len(mine), 1)
    stats = {"rank": rank, "frames": len(mine), "traced": traced, "tri_tests": totals["tri_tests"], "elapsed_s": elapsed,
             "device_ms_per_frame": round(totals["ms_total"] / nf, 4), "wall_ms_per_frame": round(elapsed / nf * 1e3, 4),
             "host_enqueue_ms": round(host["enqueue_s"] / nf * 1e3, 4), "host_wait_ms": round(host["wait_s"] / nf * 1e3, 4),
             "deliver_ms": round(host["deliver_s"] / nf * 1e3, 4), "digests": digests}
    if world > 1:
        every = [None] * world
        dist.all_gather_object(every, stats)
    else:
        every = [stats]

    if rank == 0:
        all_digests = sorted((d for s in every for d in s["digests"]), key=lambda d: d["frame"])
        golden_checked = golden_equal = 0
        delivered_equal = None
        if not args.no_verify:
            golden = {}
            try:
                if not args.tiny and not args.obj and args.form == "B":
                    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "fullframe_hashes.json")))
            except OSError:
                pass
            for d in all_digests:
                rec = golden.get(f"C5_step{d['camera']:02d}")
                if rec and (rec["width"], rec["height"], rec["subdivision"], rec["lights"], rec["mesh_level"]) == (W, H, sub, n_lights, level):
                    golden_checked += 1
                    d["equals_oracle_digest"] = d["sha256"] == rec["rgba_sha256"]
                    golden_equal += d["equals_oracle_digest"]
            assert golden_equal == golden_checked, "a frame of the batch differs from the oracle's digest of it"
            if to_rank0:
                by_frame = {d["frame"]: d["sha256"] for d in all_digests}
                delivered_equal = all(hashlib.sha256(gathered[i].cpu().numpy().tobytes()).hexdigest() == h for i, h in by_frame.items())
                assert delivered_equal, "a frame delivered to rank 0 differs from what its rank rendered"
        elapsed_max = max(s["elapsed_s"] for s in every)
        total_traced = sum(s["traced"] for s in every)
        total_tests = sum(s["tri_tests"] for s in every)
        out_line = {
            "metric": "Mrays/s + ray-tri tests/s (% FP32 peak), pig OBJ 4K 16spp, 1/2/4/8 GPU",
            "value": round(total_traced / elapsed_max / 1e6, 3),
            "unit": "Mrays/s",
            "value_note": "rays actually traced per second over the whole batch and all ranks (see the C3 line for what counts as traced)",
            "n_gpus": world,
            "steps": K,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed_max / K * 1e3, 4),
            "frames_per_s": round(K / elapsed_max, 2),
            "higher_is_better": True,
            "scaling": "strong",
            "mode": "throughput: whole frames interleaved over the ranks (frame i on rank i mod N), two in flight per rank, "
                    + ("every finished frame sent whole to rank 0" if to_rank0 else "frames stay where they were rendered, no collective in the timed region"),
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"C5: batch of {K} orbit frames of the pig mesh [{mesh_note}; default = displaced icosphere level {level}], "
                            f"{len(data.triangles)} triangles, {W}x{H}, {scene_spp(sub)} spp, {n_lights} lights, form {args.form}, "
                            "list-order hit policy" + (", full ray statistics" if args.ray_stats else ", pixels only (no ray statistics)")
                            + (" -- TINY logic-check frame" if args.tiny else ""),
                "mesh": mesh_note,
                "partition": "whole frames, frame i on rank i mod N",
            },
            "tests_per_s": total_tests / elapsed_max,
            "frames_equal_to_oracle_digest": f"{golden_equal} of {golden_checked} frames that have one (tests/golden/fullframe_hashes.json)",
            "frames_delivered_to_rank0_intact": delivered_equal,
            "per_rank": [{k: v for k, v in s.items() if k not in ("digests", "traced", "tri_tests", "elapsed_s")} for s in every],
            "frames": all_digests,
        }
        print(json.dumps(out_line), flush=True)
    for sc_ in renderers:
        sc_.close()


def walk_roofline(args, totals, steps):
    """The dominant kernel of the pixels-only frame: the cone tree walk with FP32 ray generation (FP32 VALU)."""
    ms = totals["ms_primary_kernel"] / max(totals["tri_search_launches"], 1)
    node, tri, edge = (totals[k] / steps for k in ("node_tests", "tri_tests_primary_kernel", "tri_edge_tests"))
    traced = totals["primary_rays_traced"] / steps
    flops = 6.0 * (node + tri) + 18.0 * edge + 18.0 * traced
    achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    kernel = "k_search_tree_fast<4>"
    pmc = pmc_fractions(pmc_kernel(args.config, kernel), ms)
    return {
        "bound": "valu_fp32",
        "kernel": kernel,
        "achieved": round(achieved, 3),
        "peak": PEAK_FP32_TFLOPS,
        "unit": "TFLOP/s",
        "frac": round(achieved / PEAK_FP32_TFLOPS, 4),
        "traffic": pmc.get("hbm_bytes_per_launch") if pmc else None,
        "traffic_unit": "bytes of HBM traffic per launch (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes; "
                        + os.path.relpath(PMC_FILE, ROOT) + ")",
        "algorithmic_flops_per_launch": flops,
        "algorithmic_bytes_per_launch": 8.0 * traced + 8.0 * totals["tri_candidates"] / steps,
        "work_per_launch": {"ray_node_cone_tests": node, "ray_triangle_cone_tests": tri, "ray_triangle_edge_tests": edge, "rays": traced},
        "avg_launch_ms": round(ms, 4),
        "executed": pmc,
        "note": "achieved = the FP32 operations the launch has to perform -- 3 fused multiply-adds (6 flops) per (ray, node) and per "
                "(ray, triangle) cone test, 9 (18 flops) per (ray, triangle) edge test, 18 flops per ray for its direction -- / launch "
                "time (HIP events on the library's stream around this kernel alone).  The walk itself (which records a wave "
                "visits) is scalar code and is not counted.  peak = packed FP32 (two cone tests per v_pk_fma_f32); the edge test "
                "is unpacked.  algorithmic_bytes = what must cross HBM: 8 B per ray that is searched (best position and hit id "
                "initialised) + 8 B per candidate written.  `executed` holds what the hardware did according to the PMC counters.  "
                "The launch starts each wave's walk at records found once per 8x8 block (k_block_entries), which removed nine in "
                "ten of the cone tests it used to perform: frac counts the tests that are left, so it fell while the launch got "
                "faster; the VALU-issue share in `executed` says how busy the launch keeps the vector units with everything it does "
                "(table lookups, compare-and-vote, candidate emission included).",
    }


def shade32_roofline(args, totals, steps, n_lights):
    """The other large kernel of the pixels-only frame: decisions by FP32 margins and the FP32 lighting with its error
    bound (FP32 VALU).  The work it has to perform is calculateLighting for every hit and the decision of the hit's own
    pair, counted as the reference's operations (the error-bound arithmetic that rides along is not counted)."""
    ms = totals["ms_shade"] / max(totals["tri_search_launches"], 1)
    hits = totals["primary_hits"] / steps
    flops_hit = shade_flops_per_hit(n_lights, True, 1) + FLOPS_PER_TEST
    achieved = hits * flops_hit / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    kernel = "k_shade_candidates_fast"
    pmc = pmc_fractions(pmc_kernel(args.config, kernel), ms)
    return {
        "bound": "valu_fp32",
        "kernel": kernel,
        "achieved": round(achieved, 3),
        "peak": PEAK_FP32_TFLOPS,
        "unit": "TFLOP/s",
        "frac": round(achieved / PEAK_FP32_TFLOPS, 4),
        "traffic": pmc.get("hbm_bytes_per_launch") if pmc else None,
        "traffic_unit": "bytes of HBM traffic per launch (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes; "
                        + os.path.relpath(PMC_FILE, ROOT) + ")",
        "flops_per_hit_algorithmic": flops_hit,
        "hits_per_launch": hits,
        "algorithmic_bytes_per_launch": (8 + 4 + 20) * totals["tri_candidates"] / steps,
        "avg_launch_ms": round(ms, 4),
        "executed": pmc,
        "note": "achieved = hits x the operations of calculateLighting without secondary rays (192 with 3 lights; division, sqrt "
                "and pow counted as 1) plus the 69 of Triangle.intersections for the hit's own pair / launch time (HIP events on the "
                "library's stream around this kernel alone), against the FP32 peak: the kernel evaluates them in FP32 and "
                "carries a bound on the distance from the FP64 value along, which is not counted.  algorithmic_bytes = what "
                "must cross HBM per candidate: the entry 8 and the best position 4 read, hit id and colour with bound 20 written "
                "(the sample tables, planes and material indices it also reads are a few megabytes that stay in the caches).",
    }


def shade_roofline(args, totals, steps, n_lights, inert):
    """The dominant kernel of the timed region: the shading of the decided hits (FP64 VALU)."""
    flops_hit = shade_flops_per_hit(n_lights, inert, 1 if args.form == "B" else 0)
    flops_hit_ref = shade_flops_per_hit(n_lights, False, 1 if args.form == "B" else 0)
    ms_shade = totals["ms_shade"] / steps
    hits = totals["primary_hits"] / steps
    cands = totals["tri_candidates"] / steps
    # On a form-B scene the launch also decides the hit's own (ray, triangle) pair with Triangle.intersections when it is
    # the ray's only candidate (99.5 % of the hit rays of this mesh; the others were decided by k_search_exact_multi).
    # The single candidates that turn out to be no hit (2 to 3 % of the candidates) are not counted.
    decide = FLOPS_PER_TEST if args.form == "B" else 0
    flops_hit += decide
    flops_hit_ref += decide
    achieved = hits * flops_hit / (ms_shade * 1e-3) / 1e12 if ms_shade > 0 else 0.0
    # template arguments: <top-level runs, pixels only on an inert scene, nearest-hit policy>; the recorder of form T:
    # <nearest-hit policy, secondary-ray extensions, top-level runs, bounded tree walks>
    kernel = ("k_shade_candidates<false, true, false>" if inert else
              ("k_shade_candidates<false, false, false>" if args.form == "B" else "k_trace_top<false, false, true, true>"))
    # candidate entry 8 + best position 4 + direction 24 (+ the triangle 72 and its plane 32 where the pair is decided
    # here) read; hit id 4 + colour 24 written
    bytes_hit = 64 + (104 if args.form == "B" else 0)
    pmc = pmc_fractions(pmc_kernel(args.config, kernel), ms_shade)
    return {
        "bound": "valu_fp64",
        "kernel": kernel,
        "achieved": round(achieved, 3),
        "peak": PEAK_FP64_TFLOPS,
        "unit": "TFLOP/s",
        "frac": round(achieved / PEAK_FP64_TFLOPS, 4),
        "traffic": pmc.get("hbm_bytes_per_launch") if pmc else None,
        "traffic_unit": "bytes of HBM traffic per launch (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes; "
                        + os.path.relpath(PMC_FILE, ROOT) + ")",
        "flops_per_hit_algorithmic": flops_hit,
        "flops_per_hit_as_the_reference_evaluates_it": flops_hit_ref,
        "hits_per_launch": hits,
        "avg_launch_ms": round(ms_shade, 4),
        "executed": pmc,
        "hbm_unique_bytes": {"bytes_per_candidate": 64,
                             "achieved": round(cands * 64 / (ms_shade * 1e-3) / 1e9, 1) if ms_shade > 0 else 0.0,
                             "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                             "frac": round(cands * 64 / (ms_shade * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4) if ms_shade > 0 else 0.0,
                             "note": "bytes that must cross HBM once per candidate of the launch: entry 8 + best position 4 + "
                                     "direction 24 read, hit id 4 + colour 24 written.  The triangle and plane gathers "
                                     f"({bytes_hit - 64} B per decided pair) are served by the L2 (the mesh is well under a megabyte) "
                                     "and are not counted; `traffic` is what the counters measured"},
        "note": "achieved = shaded hits per launch x the FP64 operations the launch has to perform per hit -- calculateLighting "
                "(division, sqrt, pow counted as 1; with pixels only on this scene the secondary rays' operations are not "
                "performed and not counted) plus, on a form-B scene, the 69 of Triangle.intersections for the hit's own pair, "
                "which this launch decides when it is the ray's only candidate -- / launch time (HIP events on the library's "
                "stream).  The reference's arithmetic is "
                "kept bit for bit, and IEEE FP64 division and square root are instruction sequences on this ISA: measured "
                "(profiles/r02_ubench_fp64.txt) a division occupies a SIMD for 65 cycles per wave, a square root for 90, a "
                "multiply-add for 5; the 7 normalizations of a hit alone are 7 x 216 cycles.  `executed` holds what the "
                "hardware did according to the PMC counters.",
    }


def scan_roofline(tests_per_launch, avg_ms, config):
    """The list-scan kernel: evaluated pairs and what the hardware executed for them."""
    pmc = pmc_fractions(pmc_kernel(config, "k_search_primary<4>"), avg_ms)
    r = {
        "bound": "valu_fp32",
        "kernel": "k_search_primary<4>",
        "unit": "TFLOP/s",
        "peak": PEAK_FP32_TFLOPS,
        "tests_per_launch": tests_per_launch,
        "avg_launch_ms": round(avg_ms, 4),
        "algorithmic_69flop_equiv_tflops": round(tests_per_launch * FLOPS_PER_TEST / (avg_ms * 1e-3) / 1e12, 3) if avg_ms > 0 else 0.0,
        "executed": pmc,
        "traffic": pmc.get("hbm_bytes_per_launch") if pmc else None,
        "note": "The kernel decides most pairs with a conservative 3-FMA cone test (about 7 executed FP32 flops per pair incl. "
                "the max), the rest with a 9-FMA edge test and an FP64 reference-order test.  The roofline fraction is the "
                "EXECUTED FP32 rate from the PMC counters (`executed`), not pairs x 69.",
    }
    if pmc and "executed_fp32_frac_of_peak_if_unpacked" in pmc:
        r["achieved"] = pmc["executed_fp32_tflops_if_unpacked"]
        r["frac"] = pmc["executed_fp32_frac_of_peak_if_unpacked"]
    else:
        r["achieved"] = None
        r["frac"] = None
    return r


def aux_legs(rt, scene, data, params, frame, camera_for, args, level, n_lights):
    """Extra frames after the timed region (1 GPU): the reference's ray counts, the still camera, the list scan, form T."""
    import copy
    import torch
    from raytracer_amd import _abi, scenes
    out = {}
    cam = camera_for(args.warmup)

    def run(sc, p, cameras):
        sc.render_device(cameras[0], p, frame.data_ptr())  # warm-up: operands for this search / camera
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cnts = [sc.render_device(c, p, frame.data_ptr()) for c in cameras]
        return (time.perf_counter() - t0) / len(cameras) * 1e3, cnts

    # full ray statistics: what the reference traces for this frame
    p = copy.copy(params)
    p.flags = 0
    ms, cnts = run(scene, p, [cam] * 3)
    c = cnts[-1]
    out["reference_counts"] = {"rays": c["samples"] + c["rays_reflection"] + c["rays_shadow"], "ms": round(ms, 4),
                               "breakdown": {"primary": c["samples"], "reflection": c["rays_reflection"], "shadow": c["rays_shadow"],
                                             "of_shadow_not_traced_even_with_statistics": c["shadow_rays_elided"]}}
    # the camera held still: camera-dependent preparation (sample tables, cone tree) is reused
    ms, _ = run(scene, params, [scenes.make_camera(**scenes.MESH_CAMERA)] * 5)
    out["static_camera"] = {"ms_per_step": round(ms, 4), "steps": 5}
    # the timed frames again with two of them in flight (two scene objects, each with a stream and work buffers of its
    # own: what a rank of a multi-GPU run does): the launches of one frame fill the gaps of the other
    other = None if args.no_overlap_leg else rt.Scene(data, device=scene.device)
    frames2 = [frame, torch.empty_like(frame)]
    pair, flying, n2 = [scene, other], [], args.steps
    torch.cuda.synchronize()
    for leg in range(0 if other is None else 2):  # the first pass warms the second scene up
        traced = 0
        t0 = time.perf_counter()
        for i in range(n2):
            if len(flying) == 2:
                traced += flying.pop(0).render_end()["primary_rays_traced"]
            pair[i & 1].render_begin(camera_for(args.warmup + i), params, frames2[i & 1].data_ptr())
            flying.append(pair[i & 1])
        while flying:
            traced += flying.pop(0).render_end()["primary_rays_traced"]
        ms = (time.perf_counter() - t0) / n2 * 1e3
    if other is not None:
        other.close()
        out["two_frames_in_flight"] = {"ms_per_step": round(ms, 4), "steps": n2, "mrays_per_s": round(traced / n2 / (ms * 1e-3) / 1e6, 1),
                                   "note": "same frames as the timed region, frame i+1 enqueued before frame i is waited for"}
    if args.search == "tree" and not args.no_scan_leg:
        p = copy.copy(params)
        p.search = _abi.RTMI_SEARCH_LIST_SCAN
        ms, cnts = run(scene, p, [cam] * 3)
        launches = max(sum(c["tri_search_launches"] for c in cnts), 1)
        r = scan_roofline(sum(c["tri_tests_primary_kernel"] for c in cnts) / launches, sum(c["ms_primary_kernel"] for c in cnts) / launches,
                          args.config)
        r.update({"steps": 3, "ms_per_step": round(ms, 4),
                  "tests_per_s": sum(c["tri_tests_primary_kernel"] for c in cnts) / (ms * 3e-3)})
        out["list_scan"] = r
    if args.form == "B" and args.config != "C4":
        # the same mesh listed at top level: the only form in which reflection and shadow rays do triangle work
        dT = scenes.mesh_scene(level, "T", n_lights)
        sT = rt.Scene(dT, device=scene.device)
        ms, cnts = run(sT, params, [cam] * 2)
        c = cnts[-1]
        traced = c["primary_rays_traced"] + c["rays_reflection"] + c["rays_shadow"] - c["shadow_rays_elided"]
        out["form_T"] = {"ms_per_step": round(ms, 4), "steps": 2, "traced_rays_per_step": traced,
                         "mrays_per_s": round(traced / (ms * 1e-3) / 1e6, 1), "tri_tests_per_step": c["tri_tests"],
                         "ms_shade": round(c["ms_shade"], 4)}
        sT.close()
    if args.form == "B" and args.config != "C4":
        # shadow rays that enter the bounding sphere (RTMI_FLAG_SHADOWS_ENTER_BOUNDED, the TODO at RayTracingFunctions.swift:240;
        # not the reference's behaviour): the other way to make the frame's shadow rays do triangle work
        p = copy.copy(params)
        p.flags = _abi.RTMI_FLAG_SHADOWS_ENTER_BOUNDED
        ms, cnts = run(scene, p, [cam] * 2)
        c = cnts[-1]
        traced = c["primary_rays_traced"] + c["rays_reflection"] + c["rays_shadow"] - c["shadow_rays_elided"]
        out["shadows_enter_bounded"] = {"ms_per_step": round(ms, 4), "steps": 2, "traced_rays_per_step": traced,
                                        "mrays_per_s": round(traced / (ms * 1e-3) / 1e6, 1), "tri_tests_per_step": c["tri_tests"],
                                        "ms_shade": round(c["ms_shade"], 4),
                                        "note": "extension flag, default off: the mesh shadows itself; every hit is shaded from the queue"}
    return out


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


def cpu_baseline(data, params, threads, config, moving):
    """The CPU oracle (FP64 restatement of the reference) on a bounded sample of the same workload -- a centre window of
    the same frame, same scene, same spp -- extrapolated to the whole frame by ray-triangle test count (the oracle's
    full-frame counters are in tests/golden/fullframe_hashes.json)."""
    import oracle
    from raytracer_amd import scenes
    W, H = params.width, params.height
    ww, wh = min(W - 1, 1024), min(H - 1, 512)
    x0, y0 = (W - 1 - ww) // 2, (H - 1 - wh) // 2
    p = scenes.make_params(W, H, params.pixel_subdivision, tuple(params.ambient), window=(x0, y0, x0 + ww, y0 + wh))
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
    out = {
        "value": round(rays / dt / 1e6, 4),
        "unit": "Mrays/s",
        "cores": threads,
        "cpu_model": cpu_model(),
        "kind": "port",
        "sample": f"centre window {ww}x{wh} px of the same frame, still camera ({cnt['samples']} primary rays, "
                  f"{cnt['tri_tests']} ray-triangle tests, {dt:.1f} s); rays are the reference's count for the window; the window "
                  "lies inside the mesh silhouette, so its rays are costlier than the frame average",
        "tests_per_s": cnt["tri_tests"] / dt,
        "seconds": round(dt, 2),
        "single_thread": {"value": round((c1["samples"] + c1["rays_reflection"] + c1["rays_shadow"]) / d1 / 1e6, 4), "unit": "Mrays/s",
                          "cores": 1, "sample": f"centre window {sw}x{sh} px ({c1['samples']} primary rays, {d1:.1f} s)",
                          "tests_per_s": c1["tri_tests"] / d1},
    }
    try:
        full = json.load(open(os.path.join(ROOT, "tests", "golden", "fullframe_hashes.json")))[config]["counters"]
        frame_s = full["tri_tests"] / (cnt["tri_tests"] / dt)
        frame_rays = full["samples"] + full["rays_reflection"] + full["rays_shadow"]
        out["whole_frame_extrapolated"] = {
            "seconds_per_frame": round(frame_s, 2), "mrays_per_s": round(frame_rays / frame_s / 1e6, 4), "reference_rays_per_frame": frame_rays,
            "method": f"the oracle's whole-frame count of ray-triangle tests ({full['tri_tests']}, tests/golden/fullframe_hashes.json) "
                      "divided by the window's tests per second"}
    except (OSError, KeyError):
        pass
    return out


if __name__ == "__main__":
    main()
