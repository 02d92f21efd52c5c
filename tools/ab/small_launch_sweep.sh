# launches of 1 .. 16 1080p pictures: recon_kernel only (nowide) against recon_wide_kernel always (wall), interleaved, two rounds
cp mpeg_amd/libmpeghip.so /tmp/cur.so
for r in 1 2; do for v in nowide wall; do
  cp tools/ab/libmpeghip_$v.so mpeg_amd/libmpeghip.so
  python tools/sweep_small_launches.py "r$r-$v" 1 2 3 4 5 6 8 12 16 2>/dev/null
done; done
cp /tmp/cur.so mpeg_amd/libmpeghip.so
