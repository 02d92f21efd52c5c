#!/bin/bash
# round 4, GPU call g: IDCT rounding folded into m0 — parity, A/B typical + dense against the two previous kernels
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_video.py tests/test_gpu_golden.py -x -q > gpurun_out/r4g_pytest.log 2>&1
echo "tests rc=$?"
tail -3 gpurun_out/r4g_pytest.log
PROFILES="typical dense" bash tools/gpu_ab_lib.sh r4g_ab_idct_rounding 2 --streams 1024 --host-fed-seconds 0 --single-stream 0 --reference-benchmarks 0 > /dev/null
cut -c1-90 gpurun_out/r4g_ab_idct_rounding/ab.txt
