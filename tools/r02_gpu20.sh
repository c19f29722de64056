#!/bin/bash
# 2^15 per call (N = 2 of the strong-scaled shape): lane-per-signature plan against the four-lane plan forced up to 2^15
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for rep in 1 2; do
for q in 24576 32768; do
  P2E_QUAD_MAX_N=$q timeout -k 10 120 python bench.py --steps 12 --warmup 3 --batch-log2 15 --no-cpu-baseline --no-limb-split --no-p256 --check 0 > gpurun_out/q15_${q}_$rep.json 2> gpurun_out/q15_${q}_$rep.err
  python - <<PY
import json
d = json.load(open("gpurun_out/q15_${q}_$rep.json"))
print("quad_max_n $q rep $rep:", d["value"], d["ms_per_step"], d["median_step_ms"], d["phase_ms_per_step"])
PY
done
done
