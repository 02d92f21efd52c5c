#!/bin/bash
set -u
OUT=gpurun_out/r11; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
# audio: scalar-base stores; video: the DC-word chunk flag — against the library of the commit before
sed -i 's/--audio-tile 8/--audio-tile 1/' tools/ab/audio_ab.sh
bash tools/ab/audio_ab.sh 2>&1 | tee $OUT/audio_ab.txt | tail -6
PROFILES="typical dense" bash tools/gpu_ab_lib.sh r11_ab 3 --steps 26 --warmup 13 --host-fed-seconds 0 --single-stream 0
