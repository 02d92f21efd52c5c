#!/bin/bash
# On the GPU box, after `bash tools/ab/next_round.sh all` in the build container: for every tools/ab/libmpeghip_<variant>.so
#   1. parity: the video part of the -m gpu suite with the variant in the product's place (golden streams, geometry sweeps,
#      refusals, fused RGBA, the reference's own sweeps) — a variant that fails is deleted and not timed;
#   2. interleaved A/B against the product: typical + dense, then the same with Frame.RGBA fused.
# usage: bash tools/ab/next_round_gpu.sh <tag> [rounds]        output: gpurun_out/<tag>/{parity.txt, ab.txt}, gpurun_out/<tag>_rgba/ab.txt
set -u
TAG=${1:-r5}; ROUNDS=${2:-3}
mkdir -p gpurun_out/$TAG
cp mpeg_amd/libmpeghip.so /tmp/lib_product.so
for lib in tools/ab/libmpeghip_*.so; do
  [ -f "$lib" ] || continue
  name=$(basename $lib .so); name=${name#libmpeghip_}
  cp $lib mpeg_amd/libmpeghip.so
  if timeout 300 python -m pytest tests/test_gpu_video.py tests/test_gpu_golden.py tests/test_gpu_parity_holes.py -m gpu -x -q > /tmp/parity_$name.log 2>&1; then
    echo "$name: parity ok ($(tail -1 /tmp/parity_$name.log))" | tee -a gpurun_out/$TAG/parity.txt
  else
    echo "$name: PARITY FAILED — not timed" | tee -a gpurun_out/$TAG/parity.txt
    tail -15 /tmp/parity_$name.log >> gpurun_out/$TAG/parity.txt
    rm -f $lib
  fi
done
cp /tmp/lib_product.so mpeg_amd/libmpeghip.so
SKIP="${SKIP1:-fused}" PROFILES="typical dense" bash tools/gpu_ab_lib.sh $TAG $ROUNDS --steps 26 --warmup 13
PROFILES="typical dense" bash tools/gpu_ab_lib.sh ${TAG}_rgba $ROUNDS --steps 26 --warmup 13 --rgba 1
python - <<PY
import re, collections
for tag in ("$TAG", "${TAG}_rgba"):
    acc = collections.defaultdict(list)
    try:
        for line in open("gpurun_out/%s/ab.txt" % tag):
            m = re.match(r"round \d+ (\S+)\s+(\S+): .* frac ([0-9.]+)", line)
            if m:
                acc[(m.group(2), m.group(1))].append(float(m.group(3)))
    except OSError:
        continue
    print("==", tag)
    for prof in sorted({k[0] for k in acc}):
        base = sum(acc[(prof, "cur")]) / max(1, len(acc[(prof, "cur")]))
        for (p, name), v in sorted(acc.items()):
            if p == prof:
                mean = sum(v) / len(v)
                print("%-8s %-22s frac %.4f  %+.1f %% vs product  (%d rounds)" % (prof, name, mean, (mean / base - 1) * 100 if base else 0, len(v)))
PY
