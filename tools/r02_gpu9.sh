#!/bin/bash
cd "$GRAFT_REPO_ROOT"
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-limb-split --check 0"
run() { echo "$1 | k=$2: $(env $1 timeout -k 10 100 $B --batch-log2 $2 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["median_step_ms"])')"; }
for k in 13 14; do
for rep in 1 2; do
run "P2E_X=0" $k
run "P2E_SMALL_TAKES=25,22,18" $k
run "P2E_SMALL_TAKES=27,24,16" $k
run "P2E_SMALL_TAKES=26,24,19" $k
run "P2E_SMALL_TAKES=30,25,14" $k
run "P2E_SMALL_TAKES=25,22,18 P2E_BINV_SPLIT_LOG2_LAST=2" $k
done; done
