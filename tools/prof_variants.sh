# developer aid: per-layer timeline (SRBH_PT_PROF) of libsrbh.so variants built by tools/build_variant.py.  usage: prof_variants.sh tag1 ...
P=super-resolution-building-height-estimation_amd
cp $P/libsrbh.so /tmp/base.so
for t in base "$@"; do
  if [ $t = base ]; then cp /tmp/base.so $P/libsrbh.so; else cp build/variants/libsrbh_$t.so $P/libsrbh.so; fi
  echo "== $t"; SRBH_PT_PROF=1 python tools/pt_clock.py 2>&1 | grep -v amdgpu.ids | grep -A6 "avg" | head -7 | sed 's/(of which: waiting for the own LDS-DMA/vm/; s/, at the step barriers/ bar/; s/start-to-start/s2s/'
done
cp /tmp/base.so $P/libsrbh.so
