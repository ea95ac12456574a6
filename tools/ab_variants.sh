# developer aid: same-box A/B of libsrbh.so variants built by tools/build_variant.py.  usage: ab_variants.sh tag1 tag2 ... (2 rounds)
# (variants are loaded through SRBH_LIB_PATH; the in-tree libsrbh.so is never touched)
run() { SRBH_LIB_PATH=$2 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['parity']['vs_strict_f32_gpu_path']['rel_l2'])"; }
for r in 1 2; do
  run base ""
  for t in "$@"; do run $t build/variants/libsrbh_$t.so; done
done
