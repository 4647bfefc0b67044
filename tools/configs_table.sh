#!/bin/bash
# Every configuration through bench.py, one line each (run on a GPU box from the repo root): ms per frame, traced and
# reference-equivalent ray rates.  Usage: bash tools/configs_table.sh > gpurun_out/configs.txt
run () {
  label=$1; shift
  timeout -k 10 300 python bench.py --no-cpu --no-aux "$@" 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
ref = d.get('reference_rays') or {}
print('%-58s %9.3f %12.1f' % ('$label', d['ms_per_step'], d['value']))" || exit 1
}
echo "bench.py --no-cpu --no-aux <args>, one MI355X (stand-in mesh; form B, camera moving every frame, pixels only unless stated)"
echo
printf "%-58s %9s %12s\n" "args" "ms/frame" "traced Mrays/s"
run "--config C2  (1024x1024, 4 spp, 1 light)" --config C2
run "--config C3  (3840x2160, 16 spp, 3 lights)" --config C3
run "--config C3 --static-camera" --config C3 --static-camera
run "--config C3 --ray-stats (every ray the reference traces)" --config C3 --ray-stats
run "--config C4  (1.31 M triangles, 1080p, 4 spp)" --config C4 --steps 10
run "--config C3 --form T (top-level triangles)" --config C3 --form T --steps 5
run "--config C3 --form T --static-camera" --config C3 --form T --steps 5 --static-camera
run "--config C3 --search scan" --config C3 --search scan --steps 5
run "--config C4 --search scan --steps 3" --config C4 --search scan --steps 3 --warmup 1
