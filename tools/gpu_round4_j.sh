#!/bin/bash
# round 4, GPU call j: the audio kernel's pull-ahead (the step after next towards L2) — parity, A/B at 256 and 2048 streams
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_audio.py tests/test_gpu_mp2_written.py -x -q 2>&1 | tail -3
bash tools/ab/audio_ab.sh 2>&1 | tee gpurun_out/r4j_ab_audio_pull_ahead.txt | cut -c1-100
