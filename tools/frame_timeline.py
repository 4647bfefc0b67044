#!/usr/bin/env python3
"""The GPU timeline of the last frames of a `rocprofv3 --kernel-trace` run of bench.py: start, duration and the idle gap
before every kernel (copies and fills are blit kernels and show up too).

    rocprofv3 --kernel-trace -d gpurun_out/tl/prof -o run -- python3 bench.py --no-cpu --no-aux --steps 6 --warmup 2
    python tools/frame_timeline.py gpurun_out/tl/prof/run_results.db [frames]
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    rows = sorted((s, e, n.split("(")[0].replace("void ", "")[:44]) for s, e, n in db.execute("select start, end, name from kernels"))
    starts = [i for i, r in enumerate(rows) if r[2].startswith("k_block_cull")]
    if len(starts) < frames + 1:
        raise SystemExit("not enough frames in the trace")
    i0 = starts[-frames - 1]
    t0, prev_end = rows[i0][0], None
    for s, e, n in rows[i0:starts[-1]]:
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  idle before {max(gap, 0.0):6.1f}  {n}")
        prev_end = max(prev_end or e, e)


if __name__ == "__main__":
    main()
