#!/bin/bash
# wide PMC sweep over the reconstruction kernel (typical profile): who waits for what
OUT=gpurun_out/$1; mkdir -p $OUT
cd /tmp
N=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG" \
           "SQ_WAIT_INST_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_LEVEL_WAVES" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_GDS SQ_INSTS_EXP_GDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM"; do
  N=$((N+1))
  timeout 200 rocprofv3 --pmc $SET --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_$N -o pmc -- python $GRAFT_REPO_ROOT/bench.py --streams 256 --steps 4 --warmup 2 --gop 13 --profile ${2:-typical} --cpu-seconds 0 --check 0 --audio-streams 0 > $GRAFT_REPO_ROOT/$OUT/pmc_$N.log 2>&1
  echo "pass $N rc=$?"; tail -2 $GRAFT_REPO_ROOT/$OUT/pmc_$N.log | cut -c1-200
done
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $GRAFT_REPO_ROOT/$OUT | tee $GRAFT_REPO_ROOT/$OUT/pmc_wide.txt
