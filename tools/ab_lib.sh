# developer aid: same-box A/B of two builds of libsrbh.so (the previous one kept as libsrbh_old.so.keep)
P=super-resolution-building-height-estimation_amd
cp $P/libsrbh.so /tmp/new.so
run() { timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['roofline']['avg_launch_ms'])"; }
for r in 1 2; do
cp $P/libsrbh_old.so.keep $P/libsrbh.so; run old
cp /tmp/new.so $P/libsrbh.so; run new

done
