#!/bin/bash
# round 4, fourth GPU call: streaming puts, mixed leg, parser history, the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r4d_build.log 2>&1
timeout 900 python -m pytest tests/test_gpu_device_pack.py tests/test_gpu_mixed.py -x -q > gpurun_out/r4d_pytest.log 2>&1
echo "tests rc=$?"
tail -5 gpurun_out/r4d_pytest.log
timeout 400 python tools/hostbench/sweep.py > gpurun_out/r4d_sweep.txt 2>&1
echo "sweep rc=$?"
grep "pictures/s" gpurun_out/r4d_sweep.txt | grep -v host-packed
nproc
( for r in "" tools/parse_history/before_two_level_tables tools/parse_history/two_level_tables tools/parse_history/pairs_by_vlc_loop; do
    if [ -z "$r" ]; then timeout 600 python tools/bench_parse.py --threads 1,8,16; else timeout 600 python tools/bench_parse.py --root $r --threads 1,8,16; fi
  done ) > gpurun_out/r4d_parse_history.txt 2>&1
echo "parse rc=$?"
cat gpurun_out/r4d_parse_history.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r4d_bench.json 2> gpurun_out/r4d_bench.err
echo "bench rc=$?"
tail -5 gpurun_out/r4d_bench.err
python - <<'PY'
import json
j = json.load(open("gpurun_out/r4d_bench.json"))
print("value", j["value"], "frac", j["roofline"]["frac"])
for k in ("dense", "rgba_fused", "dense_rgba_fused", "mixed", "audio", "audio_large", "audio_fma_window"):
    print(k, j[k]["roofline"]["frac"] if j.get(k) else None)
print(json.dumps(j["host_fed"], indent=1)[:2500])
print(json.dumps(j["host_parsed"], indent=1)[:2500])
print(json.dumps(j["cpu_baseline"], indent=1)[:3000])
PY
