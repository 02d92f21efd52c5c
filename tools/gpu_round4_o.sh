#!/bin/bash
# Round 4, call o: the audio legs timed after a clock-ramp warm-up; are the video legs sensitive to a longer warm-up too?
set -u
mkdir -p gpurun_out
t0=$(date +%s)
python bench.py --cpu-seconds 0 --host-fed-seconds 0 --reference-benchmarks 0 --single-stream 0 > gpurun_out/r4o_bench_w13.json 2> gpurun_out/r4o_bench_w13.err
echo "bench w13: rc $? $(( $(date +%s) - t0 )) s"
t0=$(date +%s)
python bench.py --steps 104 --warmup 52 --cpu-seconds 0 --host-fed-seconds 0 --reference-benchmarks 0 --single-stream 0 > gpurun_out/r4o_bench_w52.json 2> gpurun_out/r4o_bench_w52.err
echo "bench w52: rc $? $(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
for n in ("w13", "w52"):
    d = json.loads(open("gpurun_out/r4o_bench_%s.json" % n).read().strip().splitlines()[-1])
    row = {"typical": d["roofline"]["frac"]}
    for k in ("dense", "rgba_fused", "dense_rgba_fused", "mixed", "audio", "audio_large", "audio_fma_window"):
        if k in d:
            row[k] = d[k]["roofline"]["frac"]
    print(n, " ".join("%s %.4f" % kv for kv in row.items()))
    for k in ("audio", "audio_large", "audio_fma_window"):
        print("   ", k, d[k]["launches_timed"], "launches", "%.4f ms" % d[k]["ms_per_launch"])
PY
