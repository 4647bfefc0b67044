#!/usr/bin/env python3
"""Registers, scratch, LDS and occupancy of every kernel, from the compiler's own report (no GPU needed).

    python tools/kernel_resources.py [out.txt]

Runs `make -C raytracer_amd/csrc asm` (hipcc -S --cuda-device-only -Rpass-analysis=kernel-resource-usage) and prints
one line per kernel; kept under profiles/ so that a change in register pressure shows up in a diff."""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
FIELDS = (("VGPRs", r"\bVGPRs: (\d+)"), ("AGPRs", r"AGPRs: (\d+)"), ("SGPRs", r"TotalSGPRs: (\d+)"),
          ("scratch B/lane", r"ScratchSize \[bytes/lane\]: (\d+)"), ("LDS B/block", r"LDS Size \[bytes/block\]: (\d+)"),
          ("waves/SIMD", r"Occupancy \[waves/SIMD\]: (\d+)"))


def main():
    r = subprocess.run(["make", "-C", os.path.join(HERE, "..", "raytracer_amd", "csrc"), "asm"], capture_output=True, text=True)
    text = r.stdout + r.stderr
    blocks = re.split(r"remark: [^\n]*Function Name: ", text)[1:]
    rows = []
    for b in blocks:
        mangled = b.split()[0]
        name = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        vals = [re.search(pat, b) for _, pat in FIELDS]
        rows.append((name, [m.group(1) if m else "?" for m in vals]))
    rows.sort()
    lines = [f"{'kernel':<34}" + "".join(f"{f:>16}" for f, _ in FIELDS)]
    lines += [f"{n:<34}" + "".join(f"{v:>16}" for v in vals) for n, vals in rows]
    out = "\n".join(lines) + "\n"
    print(out, end="")
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            f.write("hipcc -Rpass-analysis=kernel-resource-usage, gfx950 (tools/kernel_resources.py)\n\n" + out)


if __name__ == "__main__":
    main()
