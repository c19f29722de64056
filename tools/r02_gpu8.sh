#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for pad in 0 16 32 48 64 96 160 272 528 1040 2064 4112; do
  echo "ld_pad=$pad: $(timeout -k 10 100 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-limb-split --check 0 --ld-pad $pad 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["median_step_ms"], d["roofline"]["frac"], d["roofline_k_expand"]["frac"])')"
done
