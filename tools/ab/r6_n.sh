#!/bin/bash
# the driver's bench command twice on one box (time-based prewarm of every video leg, the mixed leg's 80 ms)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r6n2; mkdir -p $OUT; cd $R
for i in 1 2; do
  s=$(date +%s.%N)
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --sidecar $OUT/legs_$i.json > $OUT/line_$i.json 2> $OUT/err_$i.txt; echo "rc=$?"
  e=$(date +%s.%N)
  python - <<PY
import json
d = json.load(open("$OUT/line_$i.json"))
print("run $i: %.1f s wall, %d chars; typical %.4f (prewarm %d) dense %.4f fused %.4f dense_fused %.4f mixed %.4f sif %.4f audio %.4f / %.4f / %.4f  video_test_mpg %d" % ($e - $s, len(json.dumps(d, separators=(",", ":"))),
      d["roofline"]["frac"], d["config"]["untimed_prewarm_steps"], d["dense"]["frac"], d["rgba_fused"]["frac"], d["dense_rgba_fused"]["frac"], d["mixed"]["frac"], d["sif"]["frac"],
      d["audio"]["frac"], d["audio_fma_window"]["frac"], d["audio_large"]["frac"], d["reference_benchmarks"]["decode_video_test_mpg"]))
PY
done
