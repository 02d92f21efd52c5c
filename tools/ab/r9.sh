#!/bin/bash
set -u
OUT=gpurun_out/r9; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
PROFILES="typical dense" bash tools/gpu_ab_lib.sh r9_ab 3 --steps 26 --warmup 13 --host-fed-seconds 0 --single-stream 0
PROFILES="typical dense" bash tools/gpu_ab_lib.sh r9_ab_rgba 2 --steps 26 --warmup 13 --rgba 1 --host-fed-seconds 0 --single-stream 0
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 400 $OUT/bench_default.err
python - <<PY
import json
d = json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
print("typical frac %.4f" % d["roofline"]["frac"], {k: round(d[k]["roofline"]["frac"], 4) for k in ("dense", "rgba_fused", "dense_rgba_fused", "audio", "audio_large") if d.get(k)})
print("host_fed", d.get("host_fed"))
PY
