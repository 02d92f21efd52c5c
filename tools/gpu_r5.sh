#!/bin/bash
# Round 5 GPU calls, one parametrised script (replaces the per-call tools/gpu_round4_*.sh of round 4).
#   usage: tools/gpu_r5.sh <step> [args]
#     ab <tag> <rounds> [bench args]   interleaved A/B of mpeg_amd/libmpeghip.so against every tools/ab/libmpeghip_*.so (PROFILES="typical dense")
#     insts <tag> <profile>            SQ instruction counters (one rocprofv3 --pmc pass) for cur and every variant
#     tests [pytest args]              the -m gpu suite
set -u
step=${1:-ab}; shift || true
export TMPDIR=/tmp
case $step in
ab) bash tools/gpu_ab_lib.sh "$@" ;;
insts)
  TAG=${1:-insts}; PROF=${2:-typical}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
  cp mpeg_amd/libmpeghip.so /tmp/lib_cur.so
  for which in cur $(ls tools/ab/libmpeghip_*.so 2>/dev/null | sed 's/.*libmpeghip_\(.*\)\.so/\1/' | grep -v -E "${SKIP:-^$}"); do
    if [ $which = cur ]; then cp /tmp/lib_cur.so mpeg_amd/libmpeghip.so; else cp tools/ab/libmpeghip_$which.so mpeg_amd/libmpeghip.so; fi
    ( cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_$which -o pmc -- \
        python $GRAFT_REPO_ROOT/bench.py --profile $PROF --steps 4 --warmup 2 --cpu-seconds 0 --check 0 --audio-streams 0 --rgba-streams 0 --legs "" --host-fed-seconds 0 --single-stream 0 ${BENCH_ARGS:-} > $OUT/pmc_$which.log 2>&1 )
    echo "== $which ($PROF)" | tee -a $OUT/insts.txt
    python tools/pmc_summary.py $OUT/pmc_$which 2>&1 | tee -a $OUT/insts.txt
    find $OUT/pmc_$which -name "*.csv" -size +1M -delete
  done
  cp /tmp/lib_cur.so mpeg_amd/libmpeghip.so ;;
tests) python -m pytest tests -m gpu -x -q "$@" 2>&1 | tail -15 ;;
*) echo "unknown step $step"; exit 2 ;;
esac
