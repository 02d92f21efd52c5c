export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 13 --warmup 13 --cpu-seconds 0 --check 0 --audio-streams 0 --rgba-streams 0 > /dev/null 2>&1
python - <<PY
import csv, glob
f = glob.glob('/tmp/tr/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'recon_kernel' in r['Kernel_Name']]
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in rows]
print(len(d), 'dispatches; last 13 (one GOP, ms):', ' '.join('%.2f' % x for x in d[-13:]))
PY
