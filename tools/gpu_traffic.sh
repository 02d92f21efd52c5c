#!/bin/bash
# HBM traffic of the reconstruction kernel for the four bench workloads (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
# passes) -> gpurun_out/<tag>/pmc_traffic.json (copy to profiles/pmc_traffic.json).  usage: tools/gpu_traffic.sh <tag>
set -u
TAG=${1:-traffic}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for PROF in typical dense; do for RGBA in 0 1; do for SET in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $SET --output-format csv -d $OUT/${PROF}_${RGBA}_$SET -o pmc -- python $GRAFT_REPO_ROOT/bench.py --profile $PROF --rgba $RGBA --steps 4 --warmup 2 --cpu-seconds 0 --check 0 --audio-streams 0 --legs "" > $OUT/${PROF}_${RGBA}_$SET.log 2>&1
  echo "$PROF rgba=$RGBA $SET rc=$?"
done; done; done
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, json, glob
out = {}
for prof in ("typical", "dense"):
    for rgba in (0, 1):
        v = {}
        for s in ("FETCH_SIZE", "WRITE_SIZE"):
            xs = []
            for f in glob.glob("gpurun_out/$TAG/%s_%d_%s/**/*counter_collection.csv" % (prof, rgba, s), recursive=True):
                for r in csv.DictReader(open(f)):
                    if "recon_kernel" in r["Kernel_Name"] and r["Counter_Name"] == s:
                        xs.append(float(r["Counter_Value"]))
            v[s] = sum(xs) / max(1, len(xs))
        key = prof + ("_rgba" if rgba else "")
        raw = (v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
        out[key] = {"streams": 1024, "kernel": "recon_kernel<1, %s>" % ("true" if rgba else "false"),
                    "FETCH_SIZE_KB_per_launch": v["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": v["WRITE_SIZE"],
                    "hbm_bytes_per_launch_raw": raw, "hbm_bytes_per_launch": (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024,
                    "source": "profiles/$TAG (rocprofv3 --pmc, separate passes; a builder constant of that run, not measured by bench.py)",
                    "note": "mean over the recon_kernel dispatches of bench.py --profile %s --rgba %d --steps 4 --warmup 2 at 1024 streams; "
                            "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); WRITE_SIZE uncalibrated" % (prof, rgba)}
json.dump(out, open("gpurun_out/$TAG/pmc_traffic.json", "w"), indent=1)
for k, v in out.items():
    print(k, "%.3f GB per launch (raw %.3f)" % (v["hbm_bytes_per_launch"] / 1e9, v["hbm_bytes_per_launch_raw"] / 1e9))
PY
find gpurun_out/$TAG -name "*.csv" -size +1M -delete
