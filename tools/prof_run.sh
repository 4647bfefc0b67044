#!/bin/bash
# Per-kernel times of one bench.py command on the GPU box: rocprofv3 --kernel-trace --stats, summarised as text.
#   bash tools/prof_run.sh <tag> [bench.py arguments...]      -> gpurun_out/<tag>/kernel_stats.txt, bench.json
set -u
tag=$1; shift
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/prof" -o run -- python3 bench.py "$@" > "$out/bench.json" 2> "$out/prof.err"
rc=$?
echo "rocprofv3 exit $rc"
[ $rc -eq 0 ] || { tail -20 "$out/prof.err"; exit $rc; }
db=$(find "$out/prof" -name '*.db' | head -1)
python3 tools/prof_summary.py "$db" "$out/kernel_stats.txt" "python bench.py $*"
rm -rf "$out/prof"
