#!/bin/bash
# PMC counters of the audio kernel (BASELINE config 4: 256 streams x 100 frames), separate passes.  usage: tools/gpu_pmc_audio.sh <tag>
set -u
TAG=${1:-pmc_audio}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_LEVEL_WAVES"; do
  N=$(echo $SET | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $SET --output-format csv -d $OUT/pmc_$N -o pmc -- python $GRAFT_REPO_ROOT/bench.py --streams 16 --steps 2 --warmup 1 --cpu-seconds 0 --check 0 --legs "" > $OUT/pmc_$N.log 2>&1
  echo "pmc [$SET] rc=$?"
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py gpurun_out/$TAG 2>&1 | grep -A40 "audio_kernel" | tee gpurun_out/$TAG/pmc_summary.txt
find gpurun_out/$TAG -name "*.csv" -size +2M -delete
