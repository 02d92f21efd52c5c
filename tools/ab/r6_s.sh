#!/bin/bash
# the tree at HEAD: the whole -m gpu suite, then the randomised soak on a new master seed (a quarter of its video cases with the host mirror on)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r6s}; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.txt
timeout 900 python tools/gpu_soak.py ${2:-420} ${3:-20261001} ${4:-90} > $OUT/soak.txt 2>&1; echo "soak rc=$?"; tail -4 $OUT/soak.txt
