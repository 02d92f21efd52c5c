#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprof.  Everything lands in gpurun_out/.
# usage: tools/gpu_round.sh <tag> [bench extra args...]
set -u
TAG=${1:-r01}; shift || true
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== env" | tee $OUT/env.txt
(rocminfo | grep -E "Name:|Compute Unit|Max Clock" | head -12; nproc; lscpu | grep -E "Model name|Socket|Thread|Core" ; free -g | head -2; go version 2>&1 | head -1) >> $OUT/env.txt 2>&1
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
tail -3 $OUT/smoke.log
echo "== bench typical"
timeout 900 python bench.py "$@" > $OUT/bench_typical.json 2> $OUT/bench_typical.err; echo "bench rc=$?"
tail -c 3000 $OUT/bench_typical.json; tail -5 $OUT/bench_typical.err
echo "== bench dense"
timeout 600 python bench.py --profile dense --cpu-seconds 0 --audio-streams 0 "$@" > $OUT/bench_dense.json 2> $OUT/bench_dense.err; echo "bench dense rc=$?"
tail -c 2500 $OUT/bench_dense.json; tail -5 $OUT/bench_dense.err
echo "== rocprofv3 kernel trace"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 13 --warmup 13 --cpu-seconds 0 --check 0 "$@" > $GRAFT_REPO_ROOT/$OUT/prof_trace.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
for f in $(find $OUT/prof_trace -name "*kernel_stats.csv" | head -1); do head -12 $f | cut -c1-220; done
find $OUT/prof_trace -name "*kernel_trace.csv" -delete
echo "== rocprofv3 kernel trace: Frame.RGBA"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_rgba -o trace -- python $GRAFT_REPO_ROOT/tools/bench_rgba.py 512 > $GRAFT_REPO_ROOT/$OUT/bench_rgba.txt 2>&1; echo "rocprof rgba rc=$?"
cd $GRAFT_REPO_ROOT
grep rgba_kernel $OUT/bench_rgba.txt; for f in $(find $OUT/prof_rgba -name "*kernel_stats.csv" | head -1); do head -4 $f | cut -c1-220; done
find $OUT/prof_rgba -name "*kernel_trace.csv" -delete
echo "== rocprofv3 PMC (HBM traffic; separate passes)"
cd /tmp
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $SET | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --pmc $SET --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_$N -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --cpu-seconds 0 --check 0 --audio-streams 0 --rgba-streams 0 "$@" > $GRAFT_REPO_ROOT/$OUT/pmc_$N.log 2>&1
  echo "pmc [$SET] rc=$?"
done
mkdir -p $GRAFT_REPO_ROOT/$OUT/dense
for SET in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 600 rocprofv3 --pmc $SET --output-format csv -d $GRAFT_REPO_ROOT/$OUT/dense/pmc_$SET -o pmc -- python $GRAFT_REPO_ROOT/bench.py --profile dense --steps 4 --warmup 2 --cpu-seconds 0 --check 0 --audio-streams 0 --rgba-streams 0 "$@" > $GRAFT_REPO_ROOT/$OUT/dense/pmc_$SET.log 2>&1
  echo "pmc dense [$SET] rc=$?"
done
cd $GRAFT_REPO_ROOT
mv $OUT/dense /tmp/dense_pmc_$TAG
python tools/pmc_summary.py $OUT 2>&1 | tee $OUT/pmc_summary.txt
echo "#### dense profile" | tee -a $OUT/pmc_summary.txt
python tools/pmc_summary.py /tmp/dense_pmc_$TAG 2>&1 | tee -a $OUT/pmc_summary.txt
find $OUT -name "*.csv" -size +5M -delete
du -sh $OUT
echo "== bench with the host-fed leg (separate run: it launches the kernel on small batches)"
timeout 600 python bench.py --cpu-seconds 0 --audio-streams 0 --rgba-streams 0 --host-fed-seconds 2 "$@" > $OUT/bench_host_fed.json 2> $OUT/bench_host_fed.err; echo "bench host-fed rc=$?"
python -c "import json,sys; d=json.loads(open('$OUT/bench_host_fed.json').read().strip().splitlines()[-1]); print(d['host_fed'])"
