export TMPDIR=/tmp
T=r4t
mkdir -p gpurun_out/$T
python bench.py > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err; echo "bench rc=$?"
tools/gpu_phase_timing.sh run ${T}_phase > /dev/null 2>&1
tools/gpu_pmc.sh ${T}_pmc_typical typical > /dev/null 2>&1
tools/gpu_pmc.sh ${T}_pmc_dense dense > /dev/null 2>&1
tools/gpu_traffic.sh ${T}_traffic
for leg in typical dense; do
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$T/trace_$leg -o trace -- python $GRAFT_REPO_ROOT/bench.py --profile $leg --legs "" --audio-streams 0 --cpu-seconds 0 --check 0 > $GRAFT_REPO_ROOT/gpurun_out/$T/trace_$leg.log 2>&1; echo "trace $leg rc=$?"
cd $GRAFT_REPO_ROOT
for f in $(find gpurun_out/$T/trace_$leg -name "*kernel_stats.csv" | head -1); do cp $f gpurun_out/$T/kernel_stats_$leg.csv; done
done
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$T/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 --check 0 > $GRAFT_REPO_ROOT/gpurun_out/$T/trace.log 2>&1; echo "trace rc=$?"
cd $GRAFT_REPO_ROOT
for f in $(find gpurun_out/$T/trace -name "*kernel_stats.csv" | head -1); do cp $f gpurun_out/$T/kernel_stats.csv; head -8 $f | cut -c1-200; done
find gpurun_out/$T -name "*kernel_trace.csv" -delete
(rocminfo | grep -E "Name:|Compute Unit|Max Clock" | head -8; nproc; lscpu | grep -E "Model name|Socket" ; go version 2>&1 | head -1) > gpurun_out/$T/env.txt 2>&1
