# ONE 1080p stream, a picture per launch (BASELINE config 3 as written), and the 1024-stream typical leg: the current library
# against every tools/ab/libmpeghip_<name>.so, interleaved
cp mpeg_amd/libmpeghip.so /tmp/cur.so
VARIANTS="cur $(ls tools/ab/libmpeghip_*.so 2>/dev/null | sed 's/.*libmpeghip_\(.*\)\.so/\1/')"
for r in 1 2 3; do for v in $VARIANTS; do
  if [ $v = cur ]; then cp /tmp/cur.so mpeg_amd/libmpeghip.so; else cp tools/ab/libmpeghip_$v.so mpeg_amd/libmpeghip.so; fi
  python bench.py --steps 20 --warmup 5 --legs "" --audio-streams 0 --cpu-seconds 0 --check 1 --host-fed-seconds 0 --reference-benchmarks 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=j['single_stream']
print('round $r %-8s typical frac %.4f  %.3f ms | one stream: typical %.2f us per picture (wall %.2f), dense %.2f us (wall %.2f)  %s' % ('$v', j['roofline']['frac'], j['ms_per_step'], s['typical']['us_per_picture'], s['typical']['wall_us_per_picture'], s['dense']['us_per_picture'], s['dense']['wall_us_per_picture'], s['typical']['parity'][:40]))"
done; done
cp /tmp/cur.so mpeg_amd/libmpeghip.so
