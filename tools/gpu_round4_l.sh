#!/bin/bash
# round 4, GPU call l: the whole -m gpu suite + smoke() + the default bench line (timed) + copy / kernel overlap of the hand-over
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T0=$(date +%s); timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4; echo "pytest -m gpu: $(( $(date +%s) - T0 )) s"
T0=$(date +%s); python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2; echo "smoke: $(( $(date +%s) - T0 )) s"
T0=$(date +%s); python bench.py > gpurun_out/r4l_bench_default_no_flags.json 2> gpurun_out/r4l_bench.err; echo "bench.py default: $(( $(date +%s) - T0 )) s"
tail -2 gpurun_out/r4l_bench.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r4l_bench_default_no_flags.json").read().strip().splitlines()[-1])
print("steps", j["steps"], "warmup", j["warmup"], "value %.4g frac %.4f" % (j["value"], j["roofline"]["frac"]), "matches", j["roofline"]["traffic_source_matches_build"])
print({k: round(j[k]["roofline"]["frac"], 4) for k in ("dense", "rgba_fused", "dense_rgba_fused", "mixed", "audio", "audio_large")})
print("host_fed", round(j["host_fed"]["pictures_per_s"]), "host_parsed", round(j["host_parsed"]["value"]))
PY
