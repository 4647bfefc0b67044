#!/usr/bin/env python3
"""Per-kernel PMC counters (mean per launch) and launch durations of one set of tools/pmc_passes.sh passes, as the
JSON that bench.py reads (profiles/rNN_pmc_kernels.json), plus a readable text table.

    python tools/pmc_kernels_json.py <gpurun_out/pmc_TAG> <config> <out.json> <out.txt> [note]

Kernels already in <out.json> for <config> are kept unless this set of passes holds them too (a second set of passes, taken
with another switch of the library, adds the kernels that only run under that switch).  Every entry records the commit
the tree was at (`git rev-parse --short HEAD`, or $RTMI_PROFILED_COMMIT).

HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (both in KB in rocprofv3's output): on gfx950 FETCH_SIZE reports half
the bytes of wide streaming reads, so it is doubled; for narrow gathers that makes the figure an upper estimate."""
import csv
import glob
import json
import os
import sqlite3
import subprocess
import sys
from collections import defaultdict


def main():
    d, config, out_json, out_txt = sys.argv[1:5]
    note = sys.argv[5] if len(sys.argv) > 5 else ""
    acc = defaultdict(lambda: defaultdict(list))
    for f in sorted(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")) + glob.glob(os.path.join(d, "*", "*", "*_counter_collection.csv"))):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"].split("(")[0].replace("void ", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
    dur = {}
    dbs = glob.glob(os.path.join(d, "trace", "*.db")) + glob.glob(os.path.join(d, "trace", "*", "*.db"))
    if dbs:
        db = sqlite3.connect(dbs[0])
        for name, calls, total, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
            dur[name.split("(")[0].replace("void ", "")] = (calls, avg / 1e3)  # top_kernels holds microseconds
    commit = os.environ.get("RTMI_PROFILED_COMMIT") or subprocess.run(
        ["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip()
    kernels = {}
    for k, cs in acc.items():
        if k.startswith("__amd") or k.startswith("at::"):
            continue
        e = {c: sum(v) / len(v) for c, v in cs.items()}
        e["launches_profiled"] = max(len(v) for v in cs.values())
        if k in dur:
            e["avg_launch_ms"] = dur[k][1]
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_bytes_per_launch"] = (2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024
        e["commit"] = commit
        e["source"] = f"rocprofv3 --pmc passes (tools/pmc_passes.sh) and --kernel-trace --stats of the same command; {note}"
        kernels[k] = e
    try:
        allc = json.load(open(out_json))
    except OSError:
        allc = {}
    allc.setdefault(config, {}).update({k: v for k, v in kernels.items() if "avg_launch_ms" in v})
    json.dump(allc, open(out_json, "w"), indent=1, sort_keys=True)
    with open(out_txt, "w") as f:
        f.write(f"{note}\nconfig {config}; mean counter value per launch (tools/pmc_kernels_json.py)\n\n")
        for k in sorted(kernels):
            f.write(k + "\n")
            for c, v in sorted(kernels[k].items()):
                if isinstance(v, (int, float)):
                    f.write(f"   {c:<28} {v:.6g}\n")
    print(open(out_txt).read()[:3000])


if __name__ == "__main__":
    main()
