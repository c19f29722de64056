#!/bin/bash
# smaller chain pieces (scratch of a piece within the 256 MB Infinity Cache): (P2E_MSM_PIECES, P2E_RUN_ITERS) sweep,
# one process per setting, two repetitions
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for rep in 1 2; do
for cfg in "8 9" "16 5" "14 6" "12 7" "16 9"; do
  set -- $cfg
  P2E_MSM_PIECES=$1 P2E_RUN_ITERS=$2 timeout -k 10 120 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-limb-split --no-p256 --check 0 > gpurun_out/sw_$1_$2_$rep.json 2> gpurun_out/sw_$1_$2_$rep.err
  python - <<PY
import json
d = json.load(open("gpurun_out/sw_$1_$2_$rep.json"))
print("pieces $1 run_iters $2 rep $rep:", d["ms_per_step"], d["median_step_ms"], d["roofline"]["frac"], d["roofline"]["launches_per_step"], d["phase_ms_per_step"])
PY
done
done
