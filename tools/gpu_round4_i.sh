#!/bin/bash
# round 4, GPU call i: where does the audio kernel's time go at 2048 streams — without its stores / without its loads (timing only)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
CHECK=0 bash tools/ab/audio_ab.sh 2>&1 | tee gpurun_out/r4i_audio_diag.txt | cut -c1-70
