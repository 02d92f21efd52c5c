#!/bin/bash
# the tree at HEAD on the GPU box: the whole -m gpu suite, smoke(), the driver's bench command
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r6h}; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
cp bench_legs.json $OUT/bench_legs.json 2>/dev/null
wc -c $OUT/bench_default.json
