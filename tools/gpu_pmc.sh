#!/bin/bash
# PMC counters of the reconstruction kernel, separate passes (rocprofv3 --pmc only).  usage: tools/gpu_pmc.sh <tag> <profile> [bench args]
set -u
TAG=${1:-pmc}; PROF=${2:-typical}; shift 2 || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $SET | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --pmc $SET --output-format csv -d $OUT/pmc_$N -o pmc -- python $GRAFT_REPO_ROOT/bench.py --profile $PROF --steps 4 --warmup 2 --cpu-seconds 0 --check 0 --audio-streams 0 --rgba-streams 0 --legs "" "$@" > $OUT/pmc_$N.log 2>&1
  echo "pmc [$SET] rc=$?"
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py gpurun_out/$TAG 2>&1 | tee gpurun_out/$TAG/pmc_summary.txt
find gpurun_out/$TAG -name "*.csv" -size +2M -delete
