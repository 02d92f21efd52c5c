#!/bin/bash
# Build ONE variant of libmpeghip into tools/ab/ (hipcc cross-compiles here): the A/B scripts on the GPU box
# (tools/gpu_ab_lib.sh, tools/ab/next_round_gpu.sh, tools/ab/audio_ab.sh) run every tools/ab/libmpeghip_<name>.so they find,
# interleaved with the product.
#   usage: bash tools/ab/next_round.sh <name> [-DFLAG ...]
# Options still in the sources (unmeasured so far): -DMPG_NT_AUDIO_OUT (audio output stored non-temporally), -DMPG_NT_AUDIO_IN
# (sub-band samples loaded direct-to-LDS with `nt`).  What round 3 measured and settled (profiles/r5_ab_*, r6_ab_*):
# non-temporal coefficient loads (-4 %, removed), v_med3 oddification (-1 %, removed), chroma pairs (+0.3 .. 1.9 %, now THE
# layout), non-temporal stores of the fused-RGBA instance (+1.6 .. 3.8 %, adopted), int16 tile / 8 waves per SIMD (+3 .. 4 %
# typical, -4 .. -6 % dense and fused: now a kernel instance the library picks per batch).
set -eu
cd "$(dirname "$0")/../.."
name=$1; shift
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-kernarg-preload-count=14 -fPIC -shared -I include -I mpeg_amd/csrc"
/opt/rocm/bin/hipcc $FLAGS "$@" mpeg_amd/csrc/mpeghip.hip -o tools/ab/libmpeghip_$name.so && echo built $name
