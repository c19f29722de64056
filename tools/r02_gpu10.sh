#!/bin/bash
cd "$GRAFT_REPO_ROOT"
B="python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-limb-split --check 0"
run() { echo "$1: $(env $1 timeout -k 10 100 $B 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["median_step_ms"], d["roofline"]["frac"], d["roofline_k_expand"]["frac"])')"; }
for rep in 1 2; do
run "P2E_X=0"
run "P2E_EXPAND_LDS=80000"
run "P2E_EXPAND_LDS=160000"
run "GPU_MAX_HW_QUEUES=16"
run "P2E_BINV_SPLIT_LOG2_BIG=1"
done
