#!/bin/bash
# round 6, A/Bs on ONE box, interleaved: (1) runs across row ends (cur) against round 5's rule (r5rows: a run only inside one row)
# at SIF and 160x120, planes and fused RGBA; (2) the upper bound of 16-bit entries (probe16: timing only, frames wrong) at 1080p
set -u
R=$GRAFT_REPO_ROOT; cd $R
Q="--host-fed-seconds 0 --single-stream 0 --steps 40 --warmup 13"
SKIP=probe16 PROFILES=typical bash tools/gpu_ab_lib.sh r6b_rows_sif 4 --width 352 --height 240 --streams 8192 $Q
SKIP=probe16 PROFILES=typical bash tools/gpu_ab_lib.sh r6b_rows_sif_rgba 3 --width 352 --height 240 --streams 8192 --rgba 1 $Q
SKIP=probe16 PROFILES=typical bash tools/gpu_ab_lib.sh r6b_rows_160x120 3 --width 160 --height 120 --streams 32768 $Q
SKIP=probe16 PROFILES=typical bash tools/gpu_ab_lib.sh r6b_rows_1080p 2 $Q
CHECK=0 SKIP=r5rows PROFILES="typical dense" bash tools/gpu_ab_lib.sh r6b_probe16 4 $Q
timeout 900 python -m pytest tests/test_gpu_video.py -m gpu -x -q > gpurun_out/r6b_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r6b_pytest.txt
