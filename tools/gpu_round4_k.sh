#!/bin/bash
# round 4, GPU call k: the deep instance of the audio kernel — parity, then shallow against deep at 256 .. 2048 streams
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_audio.py tests/test_gpu_mp2_written.py tests/test_gpu_golden.py -x -q 2>&1 | tail -4
bash tools/ab/audio_depth_ab.sh 2>&1 | tee gpurun_out/r4k_ab_audio_depth.txt
