#!/bin/bash
# round 4, GPU call f: the packed 16-bit dense dequantisation — parity, then A/B against the previous kernel sources
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_video.py tests/test_gpu_sparse.py -x -q > gpurun_out/r4f_pytest.log 2>&1
echo "tests rc=$?"
tail -4 gpurun_out/r4f_pytest.log
PROFILES="dense" bash tools/gpu_ab_lib.sh r4f_ab_dense_packed16 2 --streams 1024 --host-fed-seconds 0 --single-stream 0 --reference-benchmarks 0
cat gpurun_out/r4f_ab_dense_packed16/ab.txt
