#!/bin/bash
# Round 4, call v: run-to-run spread on ONE box: the driver's command five times, the legs' roofline fractions of each run
set -u
mkdir -p gpurun_out
for i in 1 2 3 4 5; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --host-fed-seconds 0 --single-stream 0 --reference-benchmarks 0 2>/dev/null | tail -1 > gpurun_out/r4v_run_$i.json
  python - $i <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r4v_run_%s.json" % sys.argv[1]).read())
print("run %s: typical %.4f (%.3f ms/step)  " % (sys.argv[1], d["roofline"]["frac"], d["ms_per_step"]) +
      "  ".join("%s %.4f" % (k, d[k]["roofline"]["frac"]) for k in ("dense", "rgba_fused", "dense_rgba_fused", "mixed", "audio", "audio_large", "audio_fma_window")))
PY
done | tee gpurun_out/r4v_repeatability.txt
