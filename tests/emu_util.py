"""Runs scenes through the CPU-emulated build of the HIP kernels (tests/emu): test infrastructure."""
import ctypes as C
import os
import re
import struct
import subprocess

import numpy as np

from raytracer_amd import _abi

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_DIR = os.path.join(HERE, "emu")
# EMU_RENDER_BIN: another build of the driver, e.g. the coverage build (tools/emu_coverage_report.py)
EMU_BIN = os.environ.get("EMU_RENDER_BIN") or os.path.join(EMU_DIR, "build", "emu_render")


def build(sanitize=True):
    args = ["make", "-s", "-C", EMU_DIR]
    if not sanitize:
        args.append("SAN=")
    subprocess.check_call(args)


def run_emulated(data, cameras, params, workdir, want_hit=True, frames_per_scene=0, env=None, timeout=900, binary=None,
                 with_stderr=False, noalloc=False):
    """Render `cameras` (a list) of scene `data`; returns (frames, stats, stdout) where frames is a list of
    (rgba[H,W,4] u8, hit[H,W,spp] i32 or None, Counters) and stats maps kernel name -> dict of emulator totals."""
    workdir = str(workdir)
    os.makedirs(workdir, exist_ok=True)
    for k, a in data.to_arrays().items():
        np.ascontiguousarray(a).tofile(os.path.join(workdir, k + ".bin"))
    with open(os.path.join(workdir, "job.bin"), "wb") as f:
        f.write(struct.pack("<4i", len(cameras), int(want_hit), int(frames_per_scene), int(noalloc)))
        f.write(bytes(params))
        for c in cameras:
            f.write(bytes(c))
    e = dict(os.environ)
    e.setdefault("ASAN_OPTIONS", "detect_leaks=1:abort_on_error=0:detect_stack_use_after_return=0")
    e.setdefault("UBSAN_OPTIONS", "print_stacktrace=1")
    e.update(env or {})
    r = subprocess.run([binary or EMU_BIN, workdir], capture_output=True, text=True, timeout=timeout, env=e)
    if r.returncode != 0:
        raise AssertionError(f"emu_render exit {r.returncode}\n{r.stdout[-3000:]}\n{r.stderr[-6000:]}")
    W, H = params.width, params.height
    frames = []
    for k in range(len(cameras)):
        rgba = np.fromfile(os.path.join(workdir, f"rgba_{k}.bin"), np.uint8).reshape(H, W, 4)
        hit = None
        if want_hit:
            hit = np.fromfile(os.path.join(workdir, f"hit_{k}.bin"), np.int32).reshape(H, W, -1)
        with open(os.path.join(workdir, f"counters_{k}.bin"), "rb") as f:
            cnt = _abi.Counters.from_buffer_copy(f.read(C.sizeof(_abi.Counters)))
        frames.append((rgba, hit, cnt))
    stats = {}
    if "emu-bad-launches 0" not in r.stdout:
        raise AssertionError("a kernel launch was rejected (empty grid or workgroup)\n" + r.stdout[-2000:] + r.stderr[-2000:])
    for line in r.stdout.splitlines():
        if line.startswith("emu-stats "):
            head, name = line.split(" kernel=", 1)  # the name is the launch expression as written, spaces included
            stats[name] = {k: int(v) for k, v in re.findall(r"(\w+)=(\d+)", head)}
    return (frames, stats, r.stdout, r.stderr) if with_stderr else (frames, stats, r.stdout)


TSAN_BIN = os.path.join(EMU_DIR, "build_tsan", "emu_render")
CSRC = os.path.join(HERE, "..", "raytracer_amd", "csrc")
# Stores that several work-items may make at once ON PURPOSE: all of them store the same value (a flag that only
# ever goes from 0 to nonzero), and it is read after the kernel has ended.
BENIGN_RACY_STORES = ("cnt->overflow = 1", "cnt->multi_overflow = 1", "cnt->live_overflow = 1", "cnt->bug = code", "ap->cnt->bug = ", "CONE_FLAG] = 1")


def race_reports(stderr):
    """ThreadSanitizer's data-race reports as lists of (file, line, source text) of the two accesses, without the
    reports whose accesses are both one of the BENIGN_RACY_STORES."""
    out = []
    for rep in stderr.split("WARNING: ThreadSanitizer: data race")[1:]:
        tops = re.findall(r"\n\s+#0 \S+.*? (/\S+?):(\d+)(?::\d+)? ", rep)[:2]
        acc = []
        for path, line in tops:
            try:
                with open(path) as f:
                    text = f.read().split("\n")[int(line) - 1].strip()
            except (OSError, IndexError):
                text = "?"
            acc.append((os.path.basename(path), int(line), text))
        if acc and all(any(b in a[2] for b in BENIGN_RACY_STORES) for a in acc):
            continue
        out.append(acc or [("?", 0, rep[:400])])
    return out
