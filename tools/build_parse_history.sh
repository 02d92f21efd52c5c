#!/bin/bash
# Older source trees with their host libraries built, for tools/bench_parse.py --root (the before / after of parser changes):
#   tools/parse_history/<name>/  = git archive of the commit + its libmpeghip / libmpeghost / test emulators
# (untracked, git-ignored; travels to the GPU box with the snapshot).   usage: tools/build_parse_history.sh name=commit ...
set -e
cd "$(dirname "$0")/.."
for spec in "$@"; do
    name=${spec%%=*}
    commit=${spec#*=}
    dir=tools/parse_history/$name
    rm -rf "$dir"
    mkdir -p "$dir"
    git archive "$commit" | tar -x -C "$dir"
    (cd "$dir" && python -c "
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from mpeg_amd import _build
_build.build_all()
import hostlib
hostlib.host(); hostlib.host_emu()
print('built', '$name', '$commit')")
    rm -rf "$dir/profiles" "$dir/gpurun_out"
done
