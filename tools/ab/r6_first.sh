#!/bin/bash
# round 6, first GPU call: the new bench line (compact + sidecar + SIF legs) and the self-launching --gpus N path
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r6a; mkdir -p $OUT; cd $R
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
cp bench_legs.json $OUT/bench_legs.json 2>/dev/null
wc -c $OUT/bench_default.json
timeout 600 python bench.py --gpus 2 --share-devices --streams 64 --steps 20 --warmup 5 --audio-streams 32 --host-fed-seconds 1 --cpu-seconds 4 --sidecar $OUT/bench_2_ranks_legs.json > $OUT/bench_2_ranks_sharing.json 2> $OUT/bench_2_ranks_sharing.err; echo "2 ranks sharing rc=$?"
timeout 300 python bench.py --gpus 2 --streams 16 --steps 2 --warmup 1 > $OUT/bench_2_ranks_refused.out 2> $OUT/bench_2_ranks_refused.err; echo "2 ranks without --share-devices rc=$?" | tee $OUT/bench_2_ranks_refused.txt
grep -h "bench.py:" $OUT/bench_2_ranks_refused.err | head -2 >> $OUT/bench_2_ranks_refused.txt
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.txt
