#!/bin/bash
# Round 4, call q: the faster coefficient loop of the host parser on the GPU box's cores (before / after), through the GPU
# (golden streams, sparse and device-packed tests), and in the default bench line (host_parsed, the one-stream leg after a warm-up)
set -u
mkdir -p gpurun_out
python -m pytest tests/test_gpu_golden.py tests/test_gpu_sparse.py tests/test_gpu_device_pack.py -x -q -m gpu 2>&1 | tail -3
{
  python tools/bench_parse.py --threads 1,16,64 --repeat 6
  true
} > gpurun_out/r4q_parse_before_after.txt 2>&1
cat gpurun_out/r4q_parse_before_after.txt
t0=$(date +%s)
python bench.py > gpurun_out/r4q_bench_default.json 2> gpurun_out/r4q_bench_default.err
echo "bench default: rc $? $(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4q_bench_default.json").read().strip().splitlines()[-1])
print("typical", d["roofline"]["frac"], "audio_large", d["audio_large"]["roofline"]["frac"])
print("single_stream", {k: (round(v["us_per_picture"], 3), round(v["frac"], 4)) for k, v in d["single_stream"].items() if isinstance(v, dict)})
hp = d["host_parsed"]
for k in ("device_packed", "host_packed", "device_packed_wide", "device_packed_wide_x4"):
    print("host_parsed", k, hp[k]["parse_threads"], "threads", round(hp[k]["pictures_per_s"]), "pictures/s", round(hp[k]["ms_parse_per_picture_per_thread"], 3), "ms per picture per thread", hp[k]["wall_seconds"])
print("host_fed", d["host_fed"]["value"] if "host_fed" in d else None)
PY
