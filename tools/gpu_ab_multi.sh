#!/bin/bash
# A/B of several BUILDS of libmpeghip (mpeg_amd/libmpeghip_<name>.so), interleaved in one GPU session.
# usage: tools/gpu_ab_multi.sh <tag> <streams> <name> [<name> ...]
TAG=$1; STREAMS=$2; shift; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
cp mpeg_amd/libmpeghip.so /tmp/keep.so
for rep in 1 2 3; do
  for which in "$@"; do
    cp mpeg_amd/libmpeghip_$which.so mpeg_amd/libmpeghip.so
    echo "== $which (rep $rep)"
    timeout 300 python tools/ab_variants.py $STREAMS 6,4,4 2>&1 | grep "variant"
  done
done | tee $OUT/ab_multi.txt
cp /tmp/keep.so mpeg_amd/libmpeghip.so
