# tools/sweep_dense_share.py with the library in the tree and with every tools/ab/libmpeghip_<name>.so (policy 1 = the int16-tile
# instance, policy 2 = whatever the library's other instance is)
cp mpeg_amd/libmpeghip.so /tmp/cur.so
for v in cur $(ls tools/ab/libmpeghip_*.so 2>/dev/null | sed 's/.*libmpeghip_\(.*\)\.so/\1/' | grep -v -E "${SKIP:-^$}"); do
  if [ $v = cur ]; then cp /tmp/cur.so mpeg_amd/libmpeghip.so; else cp tools/ab/libmpeghip_$v.so mpeg_amd/libmpeghip.so; fi
  echo "== $v"; python tools/sweep_dense_share.py 256 "$@" 2>/dev/null | grep -v "^#"
done
cp /tmp/cur.so mpeg_amd/libmpeghip.so
