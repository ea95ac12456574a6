# developer aid: same-box A/B of a bench workload under an environment switch    usage: ab_env.sh <workload> VAR=a VAR=b ...
w=$1; shift
run() { env $1 timeout 400 python bench.py --workload $w $( [ $w = predict ] && echo "--steps 12 --warmup 2" ) --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['ms_per_step'])"; }
for r in 1 2; do for v in "$@"; do run $v; done; done
