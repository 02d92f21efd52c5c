#!/bin/bash
# VALU / SALU / time per macroblock of the reconstruction kernel for several workload profiles (attribution).
OUT=gpurun_out/$1; mkdir -p $OUT
cd /tmp
for PROF in typical typical_nocoef typical_fullpel dense; do
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_$PROF -o pmc -- python $GRAFT_REPO_ROOT/bench.py --streams 256 --steps 4 --warmup 2 --gop 13 --profile $PROF --cpu-seconds 0 --check 0 --audio-streams 0 > /dev/null 2>&1
  echo "#### $PROF"; python $GRAFT_REPO_ROOT/tools/pmc_summary.py $GRAFT_REPO_ROOT/$OUT/pmc_$PROF | grep -v "^=="
done | tee $GRAFT_REPO_ROOT/$OUT/profiles.txt
