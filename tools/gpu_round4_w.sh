#!/bin/bash
# Round 4, call w: AudioBatch with page-locked host arrays: tests, then MP2 from bitstreams
set -u
python -m pytest tests/test_gpu_golden.py tests/test_gpu_mp2_written.py tests/test_gpu_audio.py -x -q -m gpu 2>&1 | tail -2
python - <<'PY'
import argparse, json, sys
sys.path.insert(0, ".")
import bench
a = bench.audio_host_parsed_leg(argparse.Namespace(), 0)
print(json.dumps(a, indent=1))
a = bench.audio_host_parsed_leg(argparse.Namespace(), 0, streams=1024)
print("1024 streams:", {k: (round(v["frames_per_s"]), round(v["ms_per_tick"], 3)) for k, v in a.items() if isinstance(v, dict)})
PY
