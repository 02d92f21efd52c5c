#!/bin/bash
# audio A/B of two builds: mpeg_amd/libmpeghip.so vs mpeg_amd/<other>.so   usage: gpu_ab_audio_builds.sh <tag> <other.so>
OUT=gpurun_out/$1; mkdir -p $OUT
cp mpeg_amd/libmpeghip.so /tmp/cur.so; cp mpeg_amd/$2 /tmp/other.so
for rep in 1 2 3; do for w in cur other; do
  cp /tmp/$w.so mpeg_amd/libmpeghip.so
  echo "== $w"; MPEGHIP_AB_ONLY=auto timeout 120 python tools/ab_audio.py 256 100 | grep chunks; MPEGHIP_AB_ONLY=auto timeout 120 python tools/ab_audio.py 2048 50 | grep chunks
done; done | tee $OUT/ab_audio_builds.txt
cp /tmp/cur.so mpeg_amd/libmpeghip.so
