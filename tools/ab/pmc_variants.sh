# SQ_INSTS_VALU / SALU / LDS per wave of every tools/ab/libmpeghip_*.so (diagnostic builds that leave the kernel early)
export TMPDIR=/tmp
cp mpeg_amd/libmpeghip.so /tmp/lib_cur.so
for which in cur $(ls tools/ab/libmpeghip_*.so | sed 's/.*libmpeghip_\(.*\)\.so/\1/'); do
  if [ $which = cur ]; then cp /tmp/lib_cur.so mpeg_amd/libmpeghip.so; else cp tools/ab/libmpeghip_$which.so mpeg_amd/libmpeghip.so; fi
  (cd /tmp && timeout 60 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH --output-format csv -d /tmp/pv_$which -o pmc -- python $GRAFT_REPO_ROOT/bench.py --profile ${PROFILE:-typical} --steps 4 --warmup 2 --cpu-seconds 0 --check 0 --audio-streams 0 --legs "" > /tmp/pv_$which.log 2>&1)
  python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/pv_$which/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "recon_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
w = m.get("SQ_WAVES", 1)
print("%-8s per wave: VALU %6.1f  SALU %6.1f  LDS %5.1f  SMEM %4.1f  VMEM rd %4.1f wr %4.1f  branch %5.1f" % ("$which", m.get("SQ_INSTS_VALU", 0) / w, m.get("SQ_INSTS_SALU", 0) / w, m.get("SQ_INSTS_LDS", 0) / w, m.get("SQ_INSTS_SMEM", 0) / w, m.get("SQ_INSTS_VMEM_RD", 0) / w, m.get("SQ_INSTS_VMEM_WR", 0) / w, m.get("SQ_INSTS_BRANCH", 0) / w))
PY
done
cp /tmp/lib_cur.so mpeg_amd/libmpeghip.so
