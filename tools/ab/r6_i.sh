#!/bin/bash
# probe: what two B pictures of one stream between the same anchors would gain when launched TOGETHER with their chunks
# interleaved (pairs: waves 2j / 2j+1 of an XCD's range take chunk j of the launch's two halves, the second half reads the first
# half's frames) against the shipped order (cur) and against the interleaved order alone (pairsnoshare); typical GOP, 1024 streams
set -u
R=$GRAFT_REPO_ROOT; cd $R
Q="--host-fed-seconds 0 --single-stream 0 --steps 39 --warmup 13"
CHECK=0 PROFILES=typical bash tools/gpu_ab_lib.sh r6i_pairs 3 $Q   # (CHECK=0: the cross-stream content check gives streams their own frames, which the probe does not survive)
