#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for rep in 1 2 3; do
  python tools/mid_sweep.py 32768 2>&1 | grep "^n="
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --batch-log2 15 --no-limb-split --no-p256 --no-cpu-baseline --check 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench 2^15:', d['value'], d['ms_per_step'], d['median_step_ms'], d['phase_ms_per_step'])"
  python tools/mid_sweep.py 16384 2>&1 | grep "^n="
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --batch-log2 14 --no-limb-split --no-p256 --no-cpu-baseline --check 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench 2^14:', d['value'], d['ms_per_step'], d['median_step_ms'])"
done
