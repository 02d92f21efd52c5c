#!/bin/bash
# A/B on ONE box: bench.py against the current libmpeghip.so ("cur") and against every tools/ab/libmpeghip_<name>.so,
# interleaved.  The libraries share the C ABI, so a variant is simply copied over the product file for its runs.
# usage: tools/gpu_ab_lib.sh <tag> <rounds> [bench args...]     (CHECK=0: timing-only variants whose frames are wrong)
set -u
TAG=${1:-ab}; ROUNDS=${2:-2}; shift 2 || true
OUT=gpurun_out/$TAG; mkdir -p $OUT
cp mpeg_amd/libmpeghip.so /tmp/lib_cur.so
VARIANTS="cur $(ls tools/ab/libmpeghip_*.so 2>/dev/null | sed 's/.*libmpeghip_\(.*\)\.so/\1/' | grep -v -E "${SKIP:-^$}")"
for r in $(seq 1 $ROUNDS); do
  for which in $VARIANTS; do
    if [ $which = cur ]; then cp /tmp/lib_cur.so mpeg_amd/libmpeghip.so; else cp tools/ab/libmpeghip_$which.so mpeg_amd/libmpeghip.so; fi
    for prof in ${PROFILES:-typical dense}; do
      timeout 300 python bench.py --profile $prof --cpu-seconds 0 --audio-streams 0 --legs "" --check ${CHECK:-1} "$@" > /tmp/ab.json 2> /tmp/ab.err || tail -3 /tmp/ab.err
      python - <<PY | tee -a $OUT/ab.txt
import json
try:
    d = json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])
    print("round $r %-10s $prof: %.4g MB/s  frac %.4f  launch %.3f ms  %s" % ("$which", d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d.get('parity', d.get('parity_ok'))))
except Exception as e:
    print("round $r $which $prof: FAILED", e)
PY
    done
  done
done
cp /tmp/lib_cur.so mpeg_amd/libmpeghip.so
