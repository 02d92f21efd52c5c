#!/bin/bash
# which kernels a lone decoder's frame costs: with the host mirror (one reconstruction launch) and without (+ one untiling launch)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r6t; mkdir -p $OUT; export TMPDIR=/tmp
for M in 1 0; do
  cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$M -o trace -- python $R/tools/lone_decoder_trace.py $M > $OUT/trace_$M.log 2>&1; echo "mirror=$M rc=$?"
  f=$(find $OUT/trace_$M -name "*kernel_stats.csv" | head -1); ( echo "# tools/lone_decoder_trace.py $M"; grep "host mirror" $OUT/trace_$M.log; cat $f ) > $OUT/kernel_stats_mirror_$M.csv
  find $OUT/trace_$M -name "*kernel_trace.csv" -delete
done
cat $OUT/kernel_stats_mirror_1.csv $OUT/kernel_stats_mirror_0.csv | cut -c1-200
