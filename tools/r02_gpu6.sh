#!/bin/bash
cd "$GRAFT_REPO_ROOT"
B="python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-limb-split --check 0"
for q in 8 16; do for d in 1 2 3; do
  echo "HWQ=$q depth=$d: $(GPU_MAX_HW_QUEUES=$q timeout -k 10 200 $B --pipeline-depth $d 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')"
done; done
