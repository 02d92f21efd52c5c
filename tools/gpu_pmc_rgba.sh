#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the fused-RGBA instance: bench.py --rgba 1 at 512 streams.
# usage: tools/gpu_pmc_rgba.sh <tag>
OUT=gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for PROF in typical dense; do
  for SET in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $SET --output-format csv -d $GRAFT_REPO_ROOT/$OUT/${PROF}/pmc_$SET -o pmc -- python $GRAFT_REPO_ROOT/bench.py --rgba 1 --streams 512 --profile $PROF --steps 4 --warmup 2 --cpu-seconds 0 --check 0 --audio-streams 0 --rgba-streams 0 > $GRAFT_REPO_ROOT/$OUT/${PROF}_$SET.log 2>&1
    echo "pmc $PROF [$SET] rc=$?"
  done
done
cd $GRAFT_REPO_ROOT
for PROF in typical dense; do echo "#### $PROF, Frame.RGBA fused, 512 streams"; python tools/pmc_summary.py $OUT/$PROF; done | tee $OUT/pmc_rgba_summary.txt
find $OUT -name "*.csv" -size +5M -delete
