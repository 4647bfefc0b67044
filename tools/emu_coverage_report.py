#!/usr/bin/env python3
"""Which branches of the kernels have the emulated runs never taken?  (test infrastructure)

    make -C tests/emu coverage
    EMU_RENDER_BIN=tests/emu/build_cov/emu_render EMU_COVERAGE_FILE=/tmp/cov/run python -m pytest tests/test_emulated_kernels.py
    EMU_RENDER_BIN=tests/emu/build_cov/emu_render EMU_COVERAGE_FILE=/tmp/cov/run python tests/emu_fuzz.py --cases 200
    python tools/emu_coverage_report.py /tmp/cov/run.*

Every emulated render writes a file of "<pc> <ran>" lines for all instrumented edges of the kernel object
(tests/emu/emu_coverage.cpp); this script ORs them, maps each edge to its source line with llvm-symbolizer and
lists, per function, the lines that hold an edge no run has taken.  Host code of the library is left out
(`--all` includes it)."""
import argparse
import collections
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SYMBOLIZER = "/opt/rocm/lib/llvm/bin/llvm-symbolizer"
BINARY = os.path.join(HERE, "..", "tests", "emu", "build_cov", "emu_render")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="+")
    ap.add_argument("--all", action="store_true", help="include the host side of rtmi.hip")
    ap.add_argument("--binary", default=BINARY)
    a = ap.parse_args()
    ran = {}
    for path in a.files:
        with open(path) as f:
            for line in f:
                pc, hit = line.split()
                ran[pc] = ran.get(pc, 0) | int(hit)
    pcs = sorted(ran, key=lambda p: int(p, 16))
    out = subprocess.run([SYMBOLIZER, "-e", a.binary, "--functions=linkage", "--demangle", "--no-inlines"],
                         input="\n".join("0x" + p for p in pcs) + "\n", capture_output=True, text=True, check=True).stdout
    blocks = [b for b in out.strip().split("\n\n")]
    assert len(blocks) == len(pcs), (len(blocks), len(pcs))
    per_func = collections.defaultdict(lambda: {"edges": 0, "ran": 0, "missed": collections.Counter(), "first_line": 1 << 30, "file": "?"})
    for pc, block in zip(pcs, blocks):
        lines = block.split("\n")
        func, where = lines[0], lines[1]
        m = re.match(r"(.*):(\d+):(\d+)$", where)
        fname, lineno = (os.path.basename(m.group(1)), int(m.group(2))) if m else ("?", 0)
        if fname not in ("rtmi.hip", "rtmi_tree.h", "rtmi_exact.h") and not fname.startswith("kernels_"):
            continue
        func = re.sub(r"\(.*", "", func)
        e = per_func[func]
        e["edges"] += 1
        if lineno > 0 and lineno < e["first_line"]:
            e["first_line"], e["file"] = lineno, fname
        if ran[pc]:
            e["ran"] += 1
        elif lineno > 0:
            e["missed"][f"{fname}:{lineno}"] += 1
    if not a.all:  # device code: everything above the scene object in rtmi.hip, and the two headers' device functions
        host_from = host_code_starts()
        per_func = {f: e for f, e in per_func.items()
                    if (e["file"] == "rtmi.hip" and e["first_line"] < host_from) or e["file"] in ("rtmi_exact.h",)
                    or e["file"].startswith("kernels_")
                    or (e["file"] == "rtmi_tree.h" and (f.startswith("k_") or f in ("merge_cones", "rec_entry", "store_cone") or "k_pack_tree" in f))}
    tot = sum(e["edges"] for e in per_func.values())
    got = sum(e["ran"] for e in per_func.values())
    print(f"{got} of {tot} edges taken ({100.0 * got / max(tot, 1):.1f} %)")
    for func, e in sorted(per_func.items(), key=lambda kv: kv[1]["ran"] - kv[1]["edges"]):
        if e["ran"] == e["edges"]:
            continue
        print(f"\n{func}: {e['ran']}/{e['edges']}")
        print("   never taken at: " + ", ".join(sorted(e["missed"], key=lambda s: (s.split(':')[0], int(s.split(':')[1])))))


def host_code_starts():
    """First line of the host half of rtmi.hip (the scene object): device code lies above it."""
    with open(os.path.join(HERE, "..", "raytracer_amd", "csrc", "rtmi.hip")) as f:
        for n, line in enumerate(f, 1):
            if line.startswith("struct rtmi_scene {") or line.startswith("struct TriGroup {"):
                return n
    return 1 << 30


if __name__ == "__main__":
    main()
