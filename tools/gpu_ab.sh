#!/bin/bash
# A/B of kernel variants + PMC counters.  usage: tools/gpu_ab.sh <tag> [streams] [variants...]
set -u
TAG=${1:-ab}; shift || true
STREAMS=${1:-256}; shift || true
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== quick parity"
timeout 600 python -m pytest tests/test_gpu_video.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for V in ${PARITY_VARIANTS:-"0,8,4" "3,8,4" "4,8,4" "4,4,4" "4,16,4"}; do
  MPEGHIP_RECON=$V timeout 300 python -m pytest tests/test_gpu_video.py -m gpu -x -q -k "reconstruction or streams" > $OUT/pytest_$V.log 2>&1; echo "variant $V: $(tail -1 $OUT/pytest_$V.log)"
done
echo "== A/B"
timeout 900 python tools/ab_variants.py $STREAMS "$@" 2>&1 | tee $OUT/ab.txt | grep -v amdgpu.ids
echo "== per-kernel times (kernel trace)"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/ktrace -o kt -- python $GRAFT_REPO_ROOT/tools/ab_variants.py $STREAMS ${TRACE_VARIANT:-4,8,4} > $GRAFT_REPO_ROOT/$OUT/ktrace.log 2>&1
cd $GRAFT_REPO_ROOT
for f in $(find $OUT/ktrace -name "*kernel_stats.csv" | head -1); do head -8 $f | cut -c1-200; done
find $OUT/ktrace -name "*kernel_trace.csv" -delete
if [ "${SKIP_PMC:-0}" = "1" ]; then du -sh $OUT; exit 0; fi
echo "== PMC"
cd /tmp
for V in ${PMC_VARIANTS:-"4,8,4"}; do
  for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
    N=$(echo $SET | tr ' ' '_' | cut -c1-40)
    MPEGHIP_RECON=$V timeout 600 rocprofv3 --pmc $SET --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_${V}_$N -o pmc -- python $GRAFT_REPO_ROOT/bench.py --streams $STREAMS --steps 4 --warmup 2 --gop 5 --cpu-seconds 0 --check 0 --audio-streams 0 > $GRAFT_REPO_ROOT/$OUT/pmc_${V}_$N.log 2>&1
    echo "pmc $V [$SET] rc=$?"
  done
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT 2>&1 | tee $OUT/pmc_summary.txt
find $OUT -name "*.csv" -size +5M -delete
du -sh $OUT
