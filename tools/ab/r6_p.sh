#!/bin/bash
# one 1080p picture per launch (recon_wide_kernel<true, false>): the kernel with its old signature (cur: the mirroring instance takes its two
# arguments in the place of the RGBA ones) against two arguments more behind the preloaded ones (extraargs), interleaved
set -u
R=$GRAFT_REPO_ROOT; cd $R
bash tools/gpu_ab_lib.sh r6p_wide_args 4 --streams 1 --rgba 1 --host-fed-seconds 0 --single-stream 0 --steps 200 --warmup 50
