#!/bin/bash
set -u
OUT=gpurun_out/r8; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; tail -5 $OUT/pytest_gpu.txt
# audio A/B first (the variants are copied over the product for their runs)
sed -i 's/--cpu-seconds 0 --check 1/--cpu-seconds 0 --check 1 --host-fed-seconds 0 --single-stream 0 --audio-tile 8/' tools/ab/audio_ab.sh
bash tools/ab/audio_ab.sh 2>&1 | tee $OUT/audio_ab.txt | tail -12
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 400 $OUT/bench_default.err
python - <<PY
import json
d = json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
print("typical frac %.4f" % d["roofline"]["frac"], {k: round(d[k]["roofline"]["frac"], 4) for k in ("dense", "rgba_fused", "dense_rgba_fused", "audio", "audio_large") if d.get(k)})
print("host_fed", d.get("host_fed"))
PY
