#!/bin/bash
# tools/ab/libmpeghip_<name>.so built from the kernel sources of a git revision (same flags as mpeg_amd/_build.py):
# the "before" arm of tools/gpu_ab_lib.sh.   usage: tools/ab/build_variant.sh <name> <rev>
set -e
cd "$(dirname "$0")/../.."
name=$1; rev=$2
tmp=$(mktemp -d)
git archive "$rev" mpeg_amd/csrc include | tar -x -C "$tmp"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-kernarg-preload-count=14 \
    -fPIC -shared -I "$tmp/include" -I "$tmp/mpeg_amd/csrc" "$tmp/mpeg_amd/csrc/mpeghip.hip" -o "tools/ab/libmpeghip_$name.so"
rm -rf "$tmp"
echo "built tools/ab/libmpeghip_$name.so from $rev"
