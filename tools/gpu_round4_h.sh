#!/bin/bash
# round 4, GPU call h: audio at 2048 streams — burst stores per run, workgroups out of lockstep (diagnostics)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/ab/audio_ab.sh 2>&1 | tee gpurun_out/r4h_ab_audio.txt | cut -c1-100
