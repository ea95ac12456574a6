# round 6: second same-box A/B of hconv16's counted waits (orig = build/variants/libsrbh_hc16orig.so): serial and pipelined train step, predict
O=gpurun_out; mkdir -p $O
run() { SRBH_TRAIN_PIPELINE=$3 SRBH_LIB_PATH=$2 timeout 300 python bench.py --workload train --steps 30 --warmup 8 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('train pipe=$3 $1', d['ms_per_step'])"; }
runp() { SRBH_LIB_PATH=$2 timeout 300 python bench.py --workload predict --steps 16 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('predict $1', d['value'])"; }
for r in 1 2 3; do run orig build/variants/libsrbh_hc16orig.so 0; run new "" 0; run orig build/variants/libsrbh_hc16orig.so 1; run new "" 1; runp orig build/variants/libsrbh_hc16orig.so; runp new ""; done | tee $O/r06i_ab_hconv16_waits.txt
