#!/bin/bash
# The round's profile set OF THE BUILD IN THE TREE (run on the GPU box; ~12 minutes):
#   1. bench.py exactly as the driver runs it                      -> <tag>_bench_default.json
#   2. rocprofv3 --kernel-trace --stats, one workload per run      -> <tag>_kernel_stats_<leg>.csv   (typical, dense, fused, dense_fused, audio)
#      + the whole default command (no host-fed leg: it launches the kernel on small batches)
#   3. rocprofv3 --pmc (separate passes, nothing else traced)      -> <tag>_pmc_<typical|dense|audio>.txt
#   4. HBM traffic FETCH_SIZE / WRITE_SIZE per workload            -> <tag>_pmc_traffic.json = profiles/pmc_traffic.json
# Every output names the sha256 of mpeg_amd/csrc/* it was measured on; bench.py compares it with the sources it loaded
# (roofline.traffic_source_matches_build).      usage: bash tools/ab/final_profiles.sh <tag> [quick]
set -u
T=${1:-rfinal}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$T; mkdir -p $OUT
export TMPDIR=/tmp
SHA=$(python - <<PY
import sys; sys.path.insert(0, "$R")
import bench; print(bench.sources_sha256())
PY
)
echo "csrc sha256 $SHA"
( echo "csrc_sha256 $SHA"; echo "head $(cd $R && git rev-parse --short HEAD 2>/dev/null || echo '(snapshot without .git)')"; date -u;
  rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6; nproc; lscpu | grep -E "Model name|Socket|NUMA node\(s\)"; go version 2>&1 | head -1 ) > $OUT/env.txt 2>&1
cd $R
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
QUIET="--legs  --audio-streams 0 --cpu-seconds 0 --check 0 --host-fed-seconds 0 --single-stream 0 --steps 20 --warmup 5"
stats() { # name, bench args...
  local name=$1; shift
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$name -o trace -- python $R/bench.py "$@" > $OUT/trace_$name.log 2>&1; echo "stats $name rc=$?"
  cd $R
  for f in $(find $OUT/trace_$name -name "*kernel_stats.csv" | head -1); do ( echo "# csrc_sha256 $SHA   bench.py $*"; cat $f ) > $OUT/kernel_stats_$name.csv; done
  # per dispatch: the stats' average includes the warm-up launches (the first ones run on cold caches); bench.py times the
  # launches after them — the average of the last --steps dispatches of the reconstruction kernel is the comparable figure
  for f in $(find $OUT/trace_$name -name "*kernel_trace.csv" | head -1); do python - "$f" "$OUT/kernel_stats_$name.csv" <<PY
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "recon_kernel" in r["Kernel_Name"] or "recon_wide_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
if d:
    timed = d[-20:] if len(d) > 20 else d
    line = "# %s: %d dispatches; the last %d (= the timed steps): average %.4f ms, min %.4f, max %.4f; the first %d (warm-up, cold caches): average %.4f ms" % (
        rows[-1]["Kernel_Name"][:40], len(d), len(timed), sum(timed) / len(timed), min(timed), max(timed), len(d) - len(timed), sum(d[:len(d) - len(timed)]) / max(1, len(d) - len(timed)))
    print(line)
    open(sys.argv[2], "a").write(line + "\n")
# the audio kernel: launches of 256 and of 2048 streams share a name and differ in grid size; bench.py times the LAST ones of
# each size (after >= 40 ms of warm-up launches), the stats' average above mixes sizes and includes the ramp
import collections
by = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "audio_kernel" in r["Kernel_Name"]:
        by[(r["Kernel_Name"][:34], int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0))].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
for (name, grid), xs in sorted(by.items()):
    xs.sort()
    d = [x[1] for x in xs]
    timed = d[len(d) // 2:]
    line = "# %s grid %d: %d dispatches; the later half (bench.py times the last ones): average %.4f ms, min %.4f, max %.4f; the first three: %s" % (
        name, grid, len(d), sum(timed) / len(timed), min(timed), max(timed), " ".join("%.4f" % v for v in d[:3]))
    print(line)
    open(sys.argv[2], "a").write(line + "\n")
PY
  done
  find $OUT/trace_$name -name "*kernel_trace.csv" -delete
}
stats typical --profile typical --legs "" --audio-streams 0 --cpu-seconds 0 --check 0 --host-fed-seconds 0 --single-stream 0 --steps 20 --warmup 5
stats dense --profile dense --legs "" --audio-streams 0 --cpu-seconds 0 --check 0 --host-fed-seconds 0 --single-stream 0 --steps 20 --warmup 5
stats fused --profile typical --rgba 1 --legs "" --audio-streams 0 --cpu-seconds 0 --check 0 --host-fed-seconds 0 --single-stream 0 --steps 20 --warmup 5
stats dense_fused --profile dense --rgba 1 --legs "" --audio-streams 0 --cpu-seconds 0 --check 0 --host-fed-seconds 0 --single-stream 0 --steps 20 --warmup 5
stats audio --streams 16 --legs "" --cpu-seconds 0 --check 0 --host-fed-seconds 0 --single-stream 0 --steps 2 --warmup 1 --audio-tile 8
stats sif --profile typical --width 352 --height 240 --streams 8192 --legs "" --audio-streams 0 --cpu-seconds 0 --check 0 --host-fed-seconds 0 --single-stream 0 --steps 20 --warmup 5
stats mixed --profile typical --legs mixed --audio-streams 0 --cpu-seconds 0 --check 0 --host-fed-seconds 0 --single-stream 0 --steps 20 --warmup 5
stats bench_default --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --check 0 --host-fed-seconds 0
# the device-packed hand-over under the kernel trace: pack_kernel / pack_gate_kernel / recon_kernel per commit of 64 pictures
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_hand_over -o trace -- python $R/tools/hostbench/sweep.py quick > $OUT/trace_hand_over.log 2>&1; echo "stats hand_over rc=$?"
cd $R
for f in $(find $OUT/trace_hand_over -name "*kernel_stats.csv" | head -1); do ( echo "# csrc_sha256 $SHA   tools/hostbench/sweep.py quick (64 typical 1080p pictures per commit)"; cat $f ) > $OUT/kernel_stats_hand_over.csv; done
find $OUT/trace_hand_over -name "*kernel_trace.csv" -delete
# the N > 1 code on this ONE GPU (bench.py --gpus N starts its ranks by itself since round 6): eight ranks share it (128 streams each): audio + video + host_fed per rank, cpu_baseline on rank 0
timeout 900 python bench.py --gpus 8 --share-devices --sidecar "" --steps 20 --warmup 5 --streams 128 --audio-streams 32 --host-fed-seconds 1 --cpu-seconds 6 > $OUT/bench_8_ranks_sharing_one_gpu.json 2> $OUT/bench_8_ranks.err; echo "8 ranks rc=$?"
tail -2 $OUT/bench_8_ranks.err
# ... and the same launch WITHOUT --share-devices: refused (ranks on one physical GPU), non-zero exit, no line
timeout 300 python bench.py --gpus 2 --sidecar "" --steps 2 --warmup 1 --streams 16 > $OUT/bench_2_ranks_refused.out 2> $OUT/bench_2_ranks_refused.err; echo "2 ranks on one GPU without --share-devices: rc=$?" | tee $OUT/bench_2_ranks_refused.txt
grep -h "bench.py:" $OUT/bench_2_ranks_refused.err | head -2 >> $OUT/bench_2_ranks_refused.txt
# ONE 1080p picture per launch (BASELINE config 3 as written): recon_wide_kernel under the kernel trace
stats single_picture --streams 1 --rgba 1 --profile typical --legs "" --audio-streams 0 --cpu-seconds 0 --check 0 --host-fed-seconds 0 --single-stream 0 --steps 40 --warmup 10
[ "${2:-}" = quick ] && { ls $OUT; exit 0; }   # (bench line + kernel stats only)
# PMC
bash tools/gpu_pmc.sh ${T}/pmc_typical typical --host-fed-seconds 0 --single-stream 0 > $OUT/pmc_typical.log 2>&1; ( echo "# csrc_sha256 $SHA"; cat $OUT/pmc_typical/pmc_summary.txt ) > $OUT/pmc_typical.txt
bash tools/gpu_pmc.sh ${T}/pmc_dense dense --host-fed-seconds 0 --single-stream 0 > $OUT/pmc_dense.log 2>&1; ( echo "# csrc_sha256 $SHA"; cat $OUT/pmc_dense/pmc_summary.txt ) > $OUT/pmc_dense.txt
bash tools/gpu_pmc.sh ${T}/pmc_single_picture typical --streams 1 --rgba 1 --steps 40 --warmup 10 --host-fed-seconds 0 --single-stream 0 > $OUT/pmc_single_picture.log 2>&1; ( echo "# csrc_sha256 $SHA   one 1080p picture per launch (recon_wide_kernel)"; cat $OUT/pmc_single_picture/pmc_summary.txt ) > $OUT/pmc_single_picture.txt
sed -i 's/--cpu-seconds 0 --check 0 --legs ""/--cpu-seconds 0 --check 0 --legs "" --host-fed-seconds 0 --single-stream 0 --audio-tile 1/' tools/gpu_pmc_audio.sh
bash tools/gpu_pmc_audio.sh ${T}/pmc_audio > $OUT/pmc_audio.log 2>&1; ( echo "# csrc_sha256 $SHA"; cat $OUT/pmc_audio/pmc_summary.txt ) > $OUT/pmc_audio.txt
# HBM traffic
cd /tmp
for PROF in typical dense; do for RGBA in 0 1; do for SET in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $SET --output-format csv -d $OUT/traffic/${PROF}_${RGBA}_$SET -o pmc -- python $R/bench.py --profile $PROF --rgba $RGBA --steps 6 --warmup 2 --cpu-seconds 0 --check 0 --audio-streams 0 --legs "" --host-fed-seconds 0 --single-stream 0 > $OUT/traffic_${PROF}_${RGBA}_$SET.log 2>&1
  echo "traffic $PROF rgba=$RGBA $SET rc=$?"
done; done; done
for SET in FETCH_SIZE WRITE_SIZE; do   # SIF 352x240, 8192 streams (bench.py's sif leg)
  timeout 600 rocprofv3 --pmc $SET --output-format csv -d $OUT/traffic/sif_$SET -o pmc -- python $R/bench.py --profile typical --width 352 --height 240 --streams 8192 --steps 6 --warmup 2 --cpu-seconds 0 --check 0 --audio-streams 0 --legs "" --host-fed-seconds 0 --single-stream 0 > $OUT/traffic_sif_$SET.log 2>&1
  echo "traffic sif $SET rc=$?"
done
for SET in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $SET --output-format csv -d $OUT/traffic/audio_$SET -o pmc -- python $R/bench.py --streams 16 --steps 2 --warmup 1 --cpu-seconds 0 --check 0 --legs "" --host-fed-seconds 0 --single-stream 0 --audio-tile 8 > $OUT/traffic_audio_$SET.log 2>&1
  echo "traffic audio $SET rc=$?"
done
cd $R
python - <<PY
import csv, json, glob, collections
out = {"csrc_sha256": "$SHA", "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/ab/final_profiles.sh); FETCH_SIZE doubled per "
       "MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads), WRITE_SIZE as reported; mean over the kernel's dispatches of the run "
       "(dense: without the priming I picture)"}
def rows(pattern, kernel, counter):
    xs = []
    for f in sorted(glob.glob(pattern, recursive=True)):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
                xs.append((int(r.get("Dispatch_Id", 0) or 0), r["Kernel_Name"], float(r["Counter_Value"])))
    return sorted(xs)
for prof in ("typical", "dense"):
    for rgba in (0, 1):
        v = {}
        for s in ("FETCH_SIZE", "WRITE_SIZE"):
            xs = rows("gpurun_out/$T/traffic/%s_%d_%s/**/*counter_collection.csv" % (prof, rgba, s), "recon_kernel", s)
            if prof == "dense":
                xs = xs[1:]
            v[s] = sum(x[2] for x in xs) / max(1, len(xs))
            kern = xs[-1][1] if xs else ""
        key = prof + ("_rgba" if rgba else "")
        out[key] = {"streams": 1024, "kernel": kern[:60], "FETCH_SIZE_KB_per_launch": v["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": v["WRITE_SIZE"],
                    "hbm_bytes_per_launch_raw": (v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024,
                    "hbm_bytes_per_launch": (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024,
                    "source": "profiles/${T}_pmc_traffic.json (rocprofv3 --pmc, separate passes; a figure of that run of the same kernel sources, not measured by bench.py)"}
v = {}
for s in ("FETCH_SIZE", "WRITE_SIZE"):
    xs = rows("gpurun_out/$T/traffic/sif_%s/**/*counter_collection.csv" % s, "recon_kernel", s)
    v[s] = sum(x[2] for x in xs) / max(1, len(xs))
if v["WRITE_SIZE"]:
    out["typical_352x240"] = {"streams": 8192, "kernel": "recon_kernel<1, false, true>", "FETCH_SIZE_KB_per_launch": v["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": v["WRITE_SIZE"],
                              "hbm_bytes_per_launch_raw": (v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024, "hbm_bytes_per_launch": (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024,
                              "source": "profiles/${T}_pmc_traffic.json (rocprofv3 --pmc, separate passes; a figure of that run of the same kernel sources, not measured by bench.py)"}
# audio: the launches of 256 streams and of 2048 streams are told apart by their grid size
by_grid = collections.defaultdict(lambda: collections.defaultdict(list))
for s in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in sorted(glob.glob("gpurun_out/$T/traffic/audio_%s/**/*counter_collection.csv" % s, recursive=True)):
        for r in csv.DictReader(open(f)):
            if "audio_kernel" in r["Kernel_Name"] and r["Counter_Name"] == s:
                by_grid[int(r["Grid_Size"])][s].append(float(r["Counter_Value"]))
for n, grid in zip((256, 2048), sorted(by_grid)):
    v = {s: sum(by_grid[grid][s]) / max(1, len(by_grid[grid][s])) for s in ("FETCH_SIZE", "WRITE_SIZE")}
    out["audio_%d" % n] = {"streams": n, "kernel": "audio_kernel<false, F32N>", "grid_size": grid, "FETCH_SIZE_KB_per_launch": v["FETCH_SIZE"],
                           "WRITE_SIZE_KB_per_launch": v["WRITE_SIZE"], "hbm_bytes_per_launch_raw": (v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024,
                           "hbm_bytes_per_launch": (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024,
                           "source": "profiles/${T}_pmc_traffic.json (rocprofv3 --pmc, separate passes; a figure of that run of the same kernel sources, not measured by bench.py)"}
json.dump(out, open("gpurun_out/$T/pmc_traffic.json", "w"), indent=1)
for k, v in out.items():
    if isinstance(v, dict):
        print(k, "%.3f GB per launch (raw %.3f)" % (v["hbm_bytes_per_launch"] / 1e9, v["hbm_bytes_per_launch_raw"] / 1e9))
PY
find $OUT -name "*.csv" -size +2M -delete
find $OUT -name "*.db" -delete
ls $OUT | head -40
# the driver's command once more, now that the traffic figures of THESE sources exist: the line whose roofline.traffic is of the build it ran
# (round 5's cited line had been taken before its traffic file: traffic_matches_build false)
cp $OUT/pmc_traffic.json $R/profiles/${T}_pmc_traffic.json
echo "{\"current\": \"${T}_pmc_traffic.json\"}" > $R/profiles/pmc_traffic.json
cd $R
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default_with_traffic.json 2> $OUT/bench_default_with_traffic.err; echo "bench (with traffic) rc=$?"
cp bench_legs.json $OUT/bench_legs.json 2>/dev/null
ls $OUT | head -60
