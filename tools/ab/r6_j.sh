#!/bin/bash
# cheap probes, interleaved with the shipped library on one box: cache-policy bits of the frame stores (nt sc1 / nt sc0 sc1 / sc0 sc1 / sc1
# against nt) and of the four window loads (sc0 / sc1 / nt / sc0 sc1 against none), issue priority early (until the loads are out) and late
# (once all data is there)
set -u
R=$GRAFT_REPO_ROOT; cd $R
Q="--host-fed-seconds 0 --single-stream 0 --steps 39 --warmup 13"
bash tools/gpu_ab_lib.sh r6j_bits 3 $Q
