# audio kernel A/B on one box: the current library against every tools/ab/libmpeghip_<name>.so, interleaved
# (BASELINE config 4: 256 streams x 100 frames; parity of all streams against the oracle in every run)
cp mpeg_amd/libmpeghip.so /tmp/cur.so
VARIANTS="cur $(ls tools/ab/libmpeghip_*.so 2>/dev/null | sed 's/.*libmpeghip_\(.*\)\.so/\1/')"
for r in 1 2 3; do for v in $VARIANTS; do
  if [ $v = cur ]; then cp /tmp/cur.so mpeg_amd/libmpeghip.so; else cp tools/ab/libmpeghip_$v.so mpeg_amd/libmpeghip.so; fi
  python bench.py --steps 2 --warmup 1 --streams 16 --legs "" --cpu-seconds 0 --check ${CHECK:-1} --host-fed-seconds 0 --single-stream 0 --audio-tile 8 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); d=j['audio']
print('round $r %-12s audio %.4g pairs/s frac %.4f %.4f ms %s' % ('$v', d['value'], d['roofline']['frac'], d['ms_per_launch'], d['parity']))
for k in ('audio_large', 'audio_fma_window'):
    a = j.get(k)
    if a: print('round $r %-12s %s frac %.4f %.4f ms %s' % ('$v', k, a['roofline']['frac'], a['ms_per_launch'], a['parity']))"
done; done
cp /tmp/cur.so mpeg_amd/libmpeghip.so
