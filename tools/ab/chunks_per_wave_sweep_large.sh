# ... and at the sizes of the bench legs: 256 / 512 / 1024 pictures per launch, three interleaved rounds
cp mpeg_amd/libmpeghip.so /tmp/cur.so
for r in 1 2 3; do for v in two_above_slots one_always; do
  cp tools/ab/libmpeghip_$v.so mpeg_amd/libmpeghip.so
  SWEEP_PROFILES=dense python tools/sweep_small_launches.py "r$r-$v" 256 512 1024 2>/dev/null
done; done
cp /tmp/cur.so mpeg_amd/libmpeghip.so
