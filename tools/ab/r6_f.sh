#!/bin/bash
# the gather of plane-leaving windows by direct-to-LDS loads issued with the other loads (cur) against the gather into registers when the motion compensation gets there (reggather = HEAD before it), interleaved
set -u
R=$GRAFT_REPO_ROOT; cd $R
Q="--host-fed-seconds 0 --single-stream 0 --steps 40 --warmup 13"
PROFILES=typical bash tools/gpu_ab_lib.sh r6g_gather_1080p 4 $Q
PROFILES=typical bash tools/gpu_ab_lib.sh r6g_gather_sif 3 --width 352 --height 240 --streams 8192 $Q
PROFILES=typical bash tools/gpu_ab_lib.sh r6g_gather_160x120 3 --width 160 --height 120 --streams 32768 $Q
PROFILES=dense bash tools/gpu_ab_lib.sh r6g_gather_dense 2 $Q
timeout 900 python -m pytest tests/test_gpu_video.py tests/test_gpu_golden.py tests/test_gpu_parity_holes.py tests/test_gpu_soak.py -m gpu -x -q > gpurun_out/r6g_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r6f_pytest.txt
