#!/bin/bash
# A/B of two BUILDS of libmpeghip (mpeg_amd/libmpeghip.so vs mpeg_amd/libmpeghip_base.so), interleaved
# in one GPU session.  usage: tools/gpu_ab_builds.sh <tag> [streams] [variant]
TAG=${1:-abb}; STREAMS=${2:-1024}; V=${3:-6,4,4}
OUT=gpurun_out/$TAG; mkdir -p $OUT
cp mpeg_amd/libmpeghip.so /tmp/new.so; cp mpeg_amd/libmpeghip_base.so /tmp/base.so
for rep in 1 2 3; do
  for which in base new; do
    cp /tmp/$which.so mpeg_amd/libmpeghip.so
    echo "== $which (rep $rep)"
    timeout 300 python tools/ab_variants.py $STREAMS $V 2>&1 | grep "variant"
  done
done | tee $OUT/ab_builds.txt
cp /tmp/new.so mpeg_amd/libmpeghip.so
if [ "${PMC:-0}" = "1" ]; then
  cd /tmp
  for which in base new; do
    cp /tmp/$which.so $GRAFT_REPO_ROOT/mpeg_amd/libmpeghip.so
    for SET in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
      N=$(echo $SET | tr ' ' '_' | cut -c1-24)
      for PROF in typical dense; do
        MPEGHIP_RECON=$V timeout 300 rocprofv3 --pmc $SET --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_${which}_${PROF}_$N -o pmc -- python $GRAFT_REPO_ROOT/bench.py --streams $STREAMS --steps 4 --warmup 2 --gop 5 --profile $PROF --cpu-seconds 0 --check 0 --audio-streams 0 > /dev/null 2>&1
      done
    done
  done
  cd $GRAFT_REPO_ROOT
  cp /tmp/new.so mpeg_amd/libmpeghip.so
  for which in base new; do for PROF in typical dense; do
    mkdir -p $OUT/sum_${which}_$PROF; cp -r $OUT/pmc_${which}_${PROF}_* $OUT/sum_${which}_$PROF/ 2>/dev/null
    echo "#### $which $PROF"; python tools/pmc_summary.py $OUT/sum_${which}_$PROF | grep -v "^=="
  done; done | tee $OUT/pmc_builds.txt
  rm -rf $OUT/sum_*
fi
