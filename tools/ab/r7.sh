#!/bin/bash
# round 3, after the consolidation: the whole -m gpu suite, one default bench run, the product against round 2's library,
# and the two tile instances against each other on every leg (the library's per-batch choice is the third arm)
set -u
TAG=${1:-r7}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 600 $OUT/bench_default.err
python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
    for k in ("roofline",):
        print("typical frac %.4f  launch %.3f ms" % (d[k]["frac"], d[k]["avg_launch_ms"]))
    for leg in ("dense", "rgba_fused", "dense_rgba_fused", "audio", "audio_large"):
        if d.get(leg):
            print(leg, "frac %.4f" % d[leg]["roofline"]["frac"])
    print("single_stream", json.dumps(d.get("single_stream"))[:600])
    print("host_fed", d.get("host_fed"))
    print("parity", d.get("parity"))
except Exception as e:
    print("bench default FAILED", e)
PY
PROFILES="typical dense" bash tools/gpu_ab_lib.sh ${TAG}_vs_r2 2 --steps 26 --warmup 13 --host-fed-seconds 0 --single-stream 0
PROFILES="typical dense" bash tools/gpu_ab_lib.sh ${TAG}_vs_r2_rgba 2 --steps 26 --warmup 13 --rgba 1 --host-fed-seconds 0 --single-stream 0
for r in 1 2; do for tile in 0 1 2; do for prof in typical dense; do for rgba in 0 1; do
  timeout 300 python bench.py --profile $prof --rgba $rgba --tile $tile --cpu-seconds 0 --audio-streams 0 --legs "" --host-fed-seconds 0 --single-stream 0 --steps 26 --warmup 13 > /tmp/t.json 2>/tmp/t.err || tail -3 /tmp/t.err
  python - <<PY | tee -a $OUT/tile_ab.txt
import json
try:
    d = json.loads(open('/tmp/t.json').read().strip().splitlines()[-1])
    print("round $r tile $tile $prof rgba $rgba: frac %.4f  launch %.3f ms" % (d['roofline']['frac'], d['roofline']['avg_launch_ms']))
except Exception as e:
    print("round $r tile $tile $prof rgba $rgba: FAILED", e)
PY
done; done; done; done
