#!/bin/bash
# timeline of a lone decoder's frame: HIP API calls and kernel executions with timestamps (rocprofv3 --hip-runtime-trace --kernel-trace, no counters)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r6x; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp && timeout 300 rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d $OUT/trace -o trace -- python $R/tools/lone_decoder_trace.py 1 > $OUT/trace.log 2>&1; echo "rc=$?"
ls -la $OUT/trace/* | head; cd $R
python - <<'PY'
import csv, glob, collections
api = [r for f in glob.glob("gpurun_out/r6x/trace/**/*hip_api_trace.csv", recursive=True) for r in csv.DictReader(open(f))]
ker = [r for f in glob.glob("gpurun_out/r6x/trace/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(f))]
print(len(api), "api calls,", len(ker), "kernels")
ev = []
for r in api:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "api", r["Function"]))
for r in ker:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "kernel", r["Kernel_Name"][:28]))
ev.sort()
# a window of ~6 frames in the middle of the last pass
mid = ev[len(ev) * 5 // 6][0]
t0 = None
out = open("gpurun_out/r6x/timeline.txt", "w")
for s, e, kind, name in ev:
    if s < mid or s > mid + 170000:
        continue
    t0 = t0 or s
    out.write("%9.2f us  +%6.2f  %-6s %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, kind, name))
out.close()
dur = collections.defaultdict(list)
for s, e, kind, name in ev[len(ev) // 2:]:
    dur[(kind, name)].append((e - s) / 1e3)
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print("%-6s %-34s n=%5d  mean %6.2f us  total %8.0f us" % (k[0], k[1], len(v), sum(v) / len(v), sum(v)))
PY
head -70 gpurun_out/r6x/timeline.txt
find $OUT/trace -name "*.csv" -size +1M -delete
