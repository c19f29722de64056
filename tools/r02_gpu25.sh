#!/bin/bash
# small-batch plan at 2^14 / 2^13: loop iterations per piece (P2E_SMALL_TAKES; default = equal pieces), one process each
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for lg in 14 13; do
for takes in default "22,20,16,10" "24,20,14,9" "20,18,15,11,6" "26,21,15,8" "19,18,16,12" "16,15,14,12,10"; do
  if [ "$takes" = default ]; then unset P2E_SMALL_TAKES P2E_MSM_PIECES_SMALL; else export P2E_SMALL_TAKES=$takes P2E_MSM_PIECES_SMALL=8; fi
  timeout -k 10 100 python bench.py --steps 20 --warmup 5 --batch-log2 $lg --no-limb-split --no-p256 --no-cpu-baseline --check 0 > gpurun_out/takes.json 2> gpurun_out/takes.err
  python -c "
import json; d=json.load(open('gpurun_out/takes.json')); print('2^$lg takes=$takes:', d['value'], d['ms_per_step'], d['median_step_ms'])"
done
done
