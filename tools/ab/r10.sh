#!/bin/bash
set -u
OUT=gpurun_out/r10; mkdir -p $OUT
# 1. the cross-stream check against a library whose replicate_kernel gives stream 517 the reference base of stream 516
cp mpeg_amd/libmpeghip.so /tmp/good.so
cp tools/ab/broken/libmpeghip_shifted_reference_base.so mpeg_amd/libmpeghip.so
( echo "# library built from mpeghip.hip with ONE change (tools/ab/r10.sh): replicate_kernel: r.v[1] += (s == 517 ? 516u : s) * k.frames256";
  echo "# -> stream 517 predicts from stream 516's frames.  Every stream holds the same bytes, so the all-streams hash cannot see it;";
  echo "# the per-stream content check (oracle/crosscheck.py) must.";
  timeout 600 python -m pytest tests/test_gpu_parity_holes.py -m gpu -q -k config5 2>&1 | tail -25 ) > $OUT/cross_stream_check_on_a_shifted_base.txt
cp /tmp/good.so mpeg_amd/libmpeghip.so
tail -8 $OUT/cross_stream_check_on_a_shifted_base.txt
timeout 600 python -m pytest tests/test_gpu_parity_holes.py -m gpu -q -k config5 2>&1 | tail -2
# 2. the in-process multi-GPU driver on one GPU
timeout 900 python tools/bench_sharded.py --contexts 1,2,8 --streams-per-context 32 --threads 8 > $OUT/sharded.json 2> $OUT/sharded.err; tail -c 300 $OUT/sharded.err; cat $OUT/sharded.json | cut -c1-1500
# 3. bench.py --gpus 8: eight ranks sharing the one GPU, 128 streams each
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 --steps 20 --warmup 5 --streams 128 > $OUT/bench_8_ranks_one_gpu.json 2> $OUT/bench_8_ranks.err; tail -c 300 $OUT/bench_8_ranks.err; tail -1 $OUT/bench_8_ranks_one_gpu.json | cut -c1-900
# 4. the default line
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 300 $OUT/bench_default.err
python - <<PY
import json
d = json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
print("typical frac %.4f" % d["roofline"]["frac"], {k: round(d[k]["roofline"]["frac"], 4) for k in ("dense", "rgba_fused", "dense_rgba_fused", "audio", "audio_large") if d.get(k)})
print("host_fed", {k: v for k, v in d["host_fed"].items() if k not in ("metric", "note")}, d["config"].get("host_numa"))
PY
