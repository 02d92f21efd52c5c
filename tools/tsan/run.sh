#!/bin/bash
# ThreadSanitizer over mpeg::VideoBatch's parallel parse + staged replay (no GPU: the lane-emulator store stands in
# for the device).  Prints the frame count; any data race is reported by TSan on stderr.  The same sources build with
# -fsanitize=address,undefined (replace the flag): clean on both golden streams, lane functions included.
set -e
cd "$(dirname "$0")/../.."
g++ -O1 -g -fsanitize=thread -std=c++17 -pthread -DMPG_EMU=1 -Iinclude -Impeg_amd/host -Impeg_amd/csrc \
    tools/tsan/videobatch_threads.cpp tools/tsan/device_stubs.cpp mpeg_amd/host/{buffer,video,audio,demux,batch}.cpp \
    tests/host_emu/emu_backend.cpp tests/kernel_emu/emu.cpp -o /tmp/tsan_videobatch 2>&1 | grep -v "warning\|note" || true
/tmp/tsan_videobatch tests/golden/test.mpeg1video
