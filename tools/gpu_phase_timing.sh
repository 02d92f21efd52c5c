#!/bin/bash
# builds the instrumented library (needs hipcc: done HERE before gpurun) and runs tools/phase_timing.py on the box
# usage (in the container): tools/gpu_phase_timing.sh build ; then gpurun -- 'tools/gpu_phase_timing.sh run <tag>'
set -u
if [ "${1:-}" = build ]; then
  /opt/rocm/bin/hipcc $(python -c "from mpeg_amd import _build; print(' '.join(_build.HIPCC_FLAGS))") -DMPG_PHASE_TIMING -I include -I mpeg_amd/csrc mpeg_amd/csrc/mpeghip.hip -o mpeg_amd/libmpeghip_timing.so
else
  TAG=${2:-phase}; mkdir -p gpurun_out/$TAG
  for prof in typical dense; do python tools/phase_timing.py $prof 2>&1 | tee -a gpurun_out/$TAG/phase_timing.txt; done
fi
