# developer aid: same-box A/B of libsrbh.so variants built by tools/build_variant.py.  usage: ab_variants.sh tag1 tag2 ... (2 rounds)
P=super-resolution-building-height-estimation_amd
cp $P/libsrbh.so /tmp/base.so
run() { timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; }
for r in 1 2; do
  cp /tmp/base.so $P/libsrbh.so; run base
  for t in "$@"; do cp build/variants/libsrbh_$t.so $P/libsrbh.so; run $t; done
done
cp /tmp/base.so $P/libsrbh.so
