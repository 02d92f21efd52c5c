# launches of 1 .. 16 1080p pictures: recon_kernel only (nowide) against recon_wide_kernel always (wall), interleaved, two rounds
# (profiles/round5_k_ab_wide_kernel_by_launch_size.txt).  The two libraries: the tree's mpeghip.hip with launch_batch's condition
# `a.n_chunks * 4 <= n_cu * 4 * 8 [* 2]` replaced by `false` (nowide) / `true` (wall) in a copy of csrc, built with _build.HIPCC_FLAGS into
# tools/ab/libmpeghip_{nowide,wall}.so (as tools/ab/build_variant.sh does for a git revision).
cp mpeg_amd/libmpeghip.so /tmp/cur.so
for r in 1 2; do for v in nowide wall; do
  cp tools/ab/libmpeghip_$v.so mpeg_amd/libmpeghip.so
  python tools/sweep_small_launches.py "r$r-$v" 1 2 3 4 5 6 8 12 16 2>/dev/null
done; done
cp /tmp/cur.so mpeg_amd/libmpeghip.so
