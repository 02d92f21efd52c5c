#!/bin/bash
# Sanitizer runs of the host stack (no GPU: the lane-emulator backends stand in for the device, so the kernels'
# lane functions run under the sanitizers too).
#   videobatch_threads: mpeg::VideoBatch's thread pool + staged replay on the damaged golden stream
#   audiobatch_threads: mpeg::AudioBatch's pooled parse (four threads, six streams) on the golden MP2 stream
#   video_mirror:       a lone Video with the look-ahead and the host mirror on / off over the damaged golden stream (reference hash)
#   mpeg_facade:        Demux, MPEG (Decode with callbacks, Seek, SeekFrame, Rewind), Video and Audio on test.mpg
#   fuzz_facade:        mutated program streams through the same stack (address,undefined only)
#   fuzz_streams:       mutated elementary streams, mutations aimed at the headers (address,undefined only)
# usage: tools/tsan/run.sh [thread|address,undefined]      (default: both; the emulator translation unit takes minutes
# to compile under a sanitizer — this is a tool, not part of the test suite)
set -e
cd "$(dirname "$0")/../.."
SRC="tools/tsan/device_stubs.cpp mpeg_amd/host/buffer.cpp mpeg_amd/host/video.cpp mpeg_amd/host/audio.cpp mpeg_amd/host/demux.cpp mpeg_amd/host/batch.cpp mpeg_amd/host/mpeg.cpp tests/host_emu/emu_backend.cpp tests/kernel_emu/emu.cpp"
for SAN in ${1:-thread address,undefined}; do
  for H in videobatch_threads audiobatch_threads mpeg_facade video_mirror; do
    g++ -O1 -g1 -fno-var-tracking-assignments -fsanitize=$SAN -std=c++17 -pthread -w -DMPG_EMU=1 -Iinclude -Impeg_amd/host -Impeg_amd/csrc tools/tsan/$H.cpp $SRC -o /tmp/san_$H
    echo "== $SAN / $H"
    if [ $H = videobatch_threads ] || [ $H = video_mirror ]; then /tmp/san_$H tests/golden/test.mpeg1video; elif [ $H = audiobatch_threads ]; then /tmp/san_$H tests/golden/test.mp2; else /tmp/san_$H tests/golden/test.mpg; fi
  done
  if [ $SAN = address,undefined ]; then
    for H in fuzz_facade fuzz_streams; do
      g++ -O1 -g1 -fno-var-tracking-assignments -fsanitize=$SAN -std=c++17 -pthread -w -DMPG_EMU=1 -Iinclude -Impeg_amd/host -Impeg_amd/csrc tools/tsan/$H.cpp $SRC -o /tmp/san_$H
    done
    echo "== $SAN / fuzz_facade";  ASAN_OPTIONS=detect_leaks=0 /tmp/san_fuzz_facade tests/golden/test.mpg 0 ${FUZZ_SEEDS:-200} | grep -v "^seed\|^  " || true
    echo "== $SAN / fuzz_streams"; ASAN_OPTIONS=detect_leaks=0 /tmp/san_fuzz_streams video tests/golden/test.mpeg1video 0 ${FUZZ_SEEDS:-200} | grep -v "^seed\|^  " || true
    ASAN_OPTIONS=detect_leaks=0 /tmp/san_fuzz_streams audio tests/golden/test.mp2 0 ${FUZZ_SEEDS:-200} | grep -v "^seed\|^  " || true
  fi
done
