#!/bin/bash
# round 6: the gpu suite after the look-ahead decoders and the per-picture refusal + the default bench line
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r6c; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
cp bench_legs.json $OUT/bench_legs.json 2>/dev/null
python - <<PY
import json
d = json.load(open("$OUT/bench_default.json"))
print("typical %.4f  dense %.4f  audio %.4f  sif %.4f  sif_single %.2f us" % (d["roofline"]["frac"], d["dense"]["frac"], d["audio"]["frac"], d["sif"]["frac"], d["sif_single"]["us_per_picture"]))
print("reference_benchmarks", d["reference_benchmarks"])
print("host_parsed", d["host_parsed"])
print("audio_host_parsed", d["audio_host_parsed"])
PY
