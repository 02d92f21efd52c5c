#!/bin/bash
# what the windows that leave their plane (kRSlow: gathered dword by dword) cost: the typical GOP against the same GOP with
# every vector kept inside the plane (--profile typical_inside), SIF and 1080p, interleaved
set -u
R=$GRAFT_REPO_ROOT; cd $R; OUT=gpurun_out/r6e; mkdir -p $OUT
Q="--cpu-seconds 0 --audio-streams 0 --legs  --host-fed-seconds 0 --single-stream 0 --steps 40 --warmup 13 --sidecar "
for r in 1 2 3; do for prof in typical typical_inside; do
  for geo in "--width 352 --height 240 --streams 8192" "--streams 1024"; do
    timeout 300 python bench.py --profile $prof $geo --cpu-seconds 0 --audio-streams 0 --legs "" --host-fed-seconds 0 --single-stream 0 --steps 40 --warmup 13 --sidecar "" > /tmp/o.json 2>/tmp/o.err || tail -3 /tmp/o.err
    python - <<PY | tee -a $OUT/ab.txt
import json
d = json.load(open("/tmp/o.json"))
print("round $r %-15s %-42s frac %.4f  launch %.3f ms  alg %.3f GB  parity %s" % ("$prof", "$geo", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"]["alg_bytes_per_launch"] / 1e9, d["parity_ok"]))
PY
  done
done; done
