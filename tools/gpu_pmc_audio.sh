#!/bin/bash
# PMC passes over the audio kernel (2048 streams x 50 frames, auto time slicing). usage: gpu_pmc_audio.sh <tag>
OUT=gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
N=0
for SET in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVES" "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  N=$((N+1))
  MPEGHIP_AB_ONLY=auto timeout 100 rocprofv3 --pmc $SET --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_$N -o pmc -- python $GRAFT_REPO_ROOT/tools/ab_audio.py 2048 50 > $GRAFT_REPO_ROOT/$OUT/pmc_$N.log 2>&1
  echo "pmc [$SET] rc=$?"
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT 2>&1 | tee $OUT/pmc_summary.txt
