# developer aid: the training step (configs[2]) with fp32 vs 16-bit block-internal tensors in the head, same box
run() { echo "$1: $(env $2 timeout 300 python bench.py --workload train --steps 10 --warmup 3 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], d['final_loss'])")"; }
run "io16=0     " "SRBH_TRAIN_IO16=0"
run "io16 none  " "SRBH_TRAIN_IO16=1 SRBH_TRAIN_IO16_ACT=none"
run "io16 c1    " "SRBH_TRAIN_IO16=1 SRBH_TRAIN_IO16_ACT=c1"
run "io16 c1c2  " "SRBH_TRAIN_IO16=1 SRBH_TRAIN_IO16_ACT=c1c2"
