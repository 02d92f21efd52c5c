#!/bin/bash
# Does the cross-stream content check (oracle/crosscheck.py) catch a per-stream base that is off by a stream?
#   here:        bash tools/ab/cross_stream_proof.sh build    -> tools/ab/broken/libmpeghip_shifted_reference_base.so: the product's
#                sources with ONE change: replicate_kernel gives stream 517 the reference-frame base of stream 516
#   on the box:  bash tools/ab/cross_stream_proof.sh run      -> the config-5 test against that library (must FAIL), then against the
#                product (must pass); output: gpurun_out/cross_stream_check_on_a_shifted_base.txt  (profiles/r10_cross_stream_check_*.txt)
set -u
cd "$(dirname "$0")/../.."
if [ "${1:-}" = build ]; then
  mkdir -p tools/ab/broken /tmp/broken
  sed 's/r.v\[1\] += s \* k.frames256;/r.v[1] += (s == 517 ? 516u : s) * k.frames256;/' mpeg_amd/csrc/mpeghip.hip > /tmp/broken/mpeghip.hip
  cmp -s mpeg_amd/csrc/mpeghip.hip /tmp/broken/mpeghip.hip && { echo "the line to break was not found"; exit 1; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-kernarg-preload-count=14 -fPIC -shared -I include -I mpeg_amd/csrc \
      /tmp/broken/mpeghip.hip -o tools/ab/broken/libmpeghip_shifted_reference_base.so && echo built
else
  OUT=gpurun_out/cross_stream_check_on_a_shifted_base.txt; mkdir -p gpurun_out
  cp mpeg_amd/libmpeghip.so /tmp/good.so
  cp tools/ab/broken/libmpeghip_shifted_reference_base.so mpeg_amd/libmpeghip.so
  ( echo "# library built from mpeghip.hip with ONE change (tools/ab/cross_stream_proof.sh): replicate_kernel: r.v[1] += (s == 517 ? 516u : s) * k.frames256";
    echo "# -> stream 517 predicts from stream 516's frames.  Every stream holds the same bytes, so the all-streams hash cannot see it;";
    echo "# the per-stream content check (oracle/crosscheck.py) must.";
    timeout 600 python -m pytest tests/test_gpu_parity_holes.py -m gpu -q -k config5 2>&1 | tail -25 ) > $OUT
  cp /tmp/good.so mpeg_amd/libmpeghip.so
  tail -6 $OUT
  timeout 600 python -m pytest tests/test_gpu_parity_holes.py -m gpu -q -k config5 2>&1 | tail -2
fi
