#!/bin/bash
# one process per configuration (stream -> hardware queue mapping depends on the order streams are created in a process)
cd "$GRAFT_REPO_ROOT"
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-limb-split --check 0"
run() { echo "$1 | k=$2: $(env $1 timeout -k 10 100 $B --batch-log2 $2 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["median_step_ms"])')"; }
for k in 13 14; do
for rep in 1 2; do
run "P2E_X=0" $k
run "P2E_BINV_SPLIT_LOG2=1" $k
run "P2E_SMALL_TAKES=24,20,16,9" $k
run "P2E_SMALL_TAKES=24,20,16,9 P2E_BINV_SPLIT_LOG2=1" $k
run "P2E_SMALL_TAKES=24,20,16,9 P2E_BINV_SPLIT_LOG2=1 P2E_BINV_SPLIT_LOG2_LAST=4" $k
run "P2E_SMALL_TAKES=22,20,17,10 P2E_BINV_SPLIT_LOG2=1" $k
run "P2E_SMALL_TAKES=28,22,14,6 P2E_BINV_SPLIT_LOG2=1" $k
done; done
