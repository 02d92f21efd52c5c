#!/bin/bash
# tools/ab/libmpeghip_<name>.so built from the kernel sources of a git revision (same flags as mpeg_amd/_build.py):
# the "before" arm of tools/gpu_ab_lib.sh.   usage: tools/ab/build_variant.sh <name> <rev>
set -e
cd "$(dirname "$0")/../.."
name=$1; rev=$2
tmp=$(mktemp -d)
git archive "$rev" mpeg_amd/csrc include | tar -x -C "$tmp"
FLAGS=${FLAGS:-$(python -c "from mpeg_amd import _build; print(' '.join(_build.HIPCC_FLAGS))")}   # (FLAGS=...: another set, e.g. a previous round's)
/opt/rocm/bin/hipcc $FLAGS -I "$tmp/include" -I "$tmp/mpeg_amd/csrc" "$tmp/mpeg_amd/csrc/mpeghip.hip" -o "tools/ab/libmpeghip_$name.so"
rm -rf "$tmp"
echo "built tools/ab/libmpeghip_$name.so from $rev"
