# developer aid: the training step (configs[2]) under side-stream / HIP-graph combinations, same box
run() { echo "$1: $(env $2 timeout 300 python bench.py --workload train --steps 10 --warmup 3 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], d['final_loss'])")"; }
run "eager side=auto" "SRBH_SIDE_STREAM=auto"
run "eager side=1   " "SRBH_SIDE_STREAM=1"
run "graph side=auto" "SRBH_SIDE_STREAM=auto SRBH_TRAIN_GRAPH=1"
run "graph side=1   " "SRBH_SIDE_STREAM=1 SRBH_TRAIN_GRAPH=1"
