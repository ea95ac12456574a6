# round 6: stem kernel tests; rocprofv3 kernel durations of the fused eval block vs the two-launch chain; predict A/B with the stem kernel
export TMPDIR=/tmp; O=$PWD/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_stem.py tests/test_gpu_dwconv.py -q -x 2>&1 | tail -4 | tee $O/r06f_test_stem.txt
cd /tmp && rocprofv3 --kernel-trace --stats -d $O/prof_hblk -o hblk -- python $GRAFT_REPO_ROOT/tools/time_hblock16.py 128 > $O/r06f_time_hblock16_prof.txt 2>&1; cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_hblk/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:8]:
    print(r['Name'][:70], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
for v in 0 1 0 1; do SRBH_STEM_EVAL=$v timeout 600 python bench.py --workload predict --steps 16 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('stem=$v', d['value'], d['p50_city_latency_ms'])"; done | tee $O/r06f_ab_predict_stem.txt
rm -rf $O/prof_hblk
