#!/bin/bash
# build the instrumented library HERE (no GPU needed), then on the GPU box:  python tools/phase_timing.py [typical|dense]
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DMPG_PHASE_TIMING \
  -Iinclude -Impeg_amd/csrc mpeg_amd/csrc/mpeghip.hip -o mpeg_amd/libmpeghip_timing.so
