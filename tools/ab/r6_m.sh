#!/bin/bash
# the mixed leg with its warm-up of >= 80 ms of device work, three times on one box
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r6zz; mkdir -p $OUT; cd $R
for i in 1 2 3; do
  s=$(date +%s.%N)
  timeout 600 python bench.py --legs mixed --audio-streams 0 --cpu-seconds 0 --host-fed-seconds 0 --single-stream 0 --steps 20 --warmup 5 --sidecar $OUT/legs_$i.json > $OUT/line_$i.json 2> $OUT/err_$i.txt
  e=$(date +%s.%N)
  python - <<PY
import json
d = json.load(open("$OUT/legs_$i.json")); m = d["mixed"]
print("run $i: mixed frac %.4f  %.3f ms per step, %d untimed steps; primary frac %.4f; wall %.1f s; %s" % (m["roofline"]["frac"], m["ms_per_step"], m["untimed_warm_steps"], d["roofline"]["frac"], $e - $s, m["parity"][:70]))
PY
done
