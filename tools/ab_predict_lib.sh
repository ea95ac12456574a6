# developer aid: same-box A/B of the predict and train workloads between the in-tree libsrbh.so and variants (tools/build_variant.py)
# usage: ab_predict_lib.sh tag1 tag2 ...
run() { for w in predict train; do
  SRBH_LIB_PATH=$2 timeout 400 python bench.py --workload $w $( [ $w = predict ] && echo "--steps 12 --warmup 2" ) --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', '$w', d['value'], d['ms_per_step'])"; done; }
for r in 1 2; do
  run base ""
  for t in "$@"; do run $t build/variants/libsrbh_$t.so; done
done
