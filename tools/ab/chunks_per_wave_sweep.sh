# dense launches of k 1080p pictures: two chunks per wave as soon as the chunks outnumber the wave slots (two_above_slots: what
# launch_batch did) against one chunk per wave at any size (one_always), interleaved, two rounds
# (profiles/round5_k_ab_chunks_per_wave_by_launch_size.txt; both libraries were builds of 4f40d9b's csrc, the parent of the commit that removed
# the two-chunk form: launch_batch's `a.n_chunks <= wave_slots` as it was / replaced by `true`, _build.HIPCC_FLAGS.)
cp mpeg_amd/libmpeghip.so /tmp/cur.so
for r in 1 2; do for v in two_above_slots one_always; do
  cp tools/ab/libmpeghip_$v.so mpeg_amd/libmpeghip.so
  SWEEP_PROFILES=dense python tools/sweep_small_launches.py "r$r-$v" 3 4 6 8 12 16 24 32 48 64 128 256 2>/dev/null
done; done
cp /tmp/cur.so mpeg_amd/libmpeghip.so
