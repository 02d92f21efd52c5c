#!/bin/bash
# Round 4, call x: bench.py under other --steps / --warmup than the default (the driver may pass any), and with 2 ranks on this one GPU
set -u
mkdir -p gpurun_out
for args in "--steps 5 --warmup 2" "--steps 1 --warmup 0" "--steps 60 --warmup 0"; do
  t0=$(date +%s)
  python bench.py --gpus 1 $args --cpu-seconds 2 --host-fed-seconds 0.5 2> gpurun_out/r4x.err | tail -1 > gpurun_out/r4x.json; rc=$?
  python - "$args" $rc $(( $(date +%s) - t0 )) <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r4x.json").read())
print("bench.py %s: rc %s, %s s, steps %d warmup %d, value %.4g, frac %.4f, audio %.4f / %.4f, host_fed %s, keys %d" % (
    sys.argv[1], sys.argv[2], sys.argv[3], d["steps"], d["warmup"], d["value"], d["roofline"]["frac"], d["audio"]["roofline"]["frac"],
    d["audio_large"]["roofline"]["frac"], round(d["host_fed"]["pictures_per_s"]) if d.get("host_fed") else None, len(d)))
PY
done
t0=$(date +%s)
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 10 --warmup 3 --streams 256 --cpu-seconds 2 --host-fed-seconds 0.5 2> gpurun_out/r4x2.err | tail -1 > gpurun_out/r4x2.json
python - $(( $(date +%s) - t0 )) <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r4x2.json").read())
print("2 ranks on one GPU: %s s, n_gpus %d, value %.4g, audio n_gpus %s value %.4g launches %s, host_fed %s" % (
    sys.argv[1], d["n_gpus"], d["value"], d["audio"]["n_gpus"], d["audio"]["value"], d["audio"]["launches_timed"], d.get("host_fed", {}).get("pictures_per_s")))
PY
tail -2 gpurun_out/r4x2.err
