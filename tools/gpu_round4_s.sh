#!/bin/bash
# Round 4, call s: the parser after its last changes on the GPU box's cores, the parser-through-GPU tests, the default bench line
set -u
mkdir -p gpurun_out
python -m pytest tests/test_gpu_golden.py tests/test_gpu_sparse.py tests/test_gpu_device_pack.py -x -q -m gpu 2>&1 | tail -2
{
  python tools/bench_parse.py --threads 1,16 --repeat 6
  python tools/bench_parse.py --root tools/parse_history/pairs_by_vlc_loop --threads 1 --repeat 6
} > gpurun_out/r4s_parse_before_after.txt 2>&1
cat gpurun_out/r4s_parse_before_after.txt
python bench.py > gpurun_out/r4s_bench_default.json 2> gpurun_out/r4s_bench_default.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4s_bench_default.json").read().strip().splitlines()[-1])
print("typical", round(d["roofline"]["frac"], 4), {k: round(d[k]["roofline"]["frac"], 4) for k in ("dense", "rgba_fused", "dense_rgba_fused", "mixed", "audio", "audio_large")})
hp = d["host_parsed"]
print({k: (hp[k]["parse_threads"], round(hp[k]["pictures_per_s"]), round(hp[k]["ms_parse_per_picture_per_thread"], 3)) for k in ("device_packed", "host_packed", "device_packed_wide", "device_packed_wide_x4")})
PY
