#!/bin/bash
# Round 4, call t: AudioBatch with a parse pool on the GPU (test), MP2 from bitstreams in the default bench line
set -u
mkdir -p gpurun_out
python -m pytest tests/test_gpu_golden.py tests/test_gpu_mp2_written.py -x -q -m gpu 2>&1 | tail -2
python bench.py > gpurun_out/r4t_bench_default.json 2> gpurun_out/r4t_bench_default.err; echo "bench rc $?"
tail -3 gpurun_out/r4t_bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4t_bench_default.json").read().strip().splitlines()[-1])
print("typical", round(d["roofline"]["frac"], 4), {k: round(d[k]["roofline"]["frac"], 4) for k in ("dense", "mixed", "audio", "audio_large")})
print("host_parsed", round(d["host_parsed"]["value"]))
a = d["audio_host_parsed"]
for k, v in a.items():
    if isinstance(v, dict):
        print(k, {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items()})
print("audio_host_parsed value", round(a["value"]), "frames/s =", round(a["realtime_streams_44k1"]), "real-time 44.1 kHz streams")
PY
