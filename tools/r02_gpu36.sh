#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for v in default runs; do
  if [ $v = runs ]; then export P2E_CP_RUNS_MIN_N=1; else unset P2E_CP_RUNS_MIN_N; fi
  REPS=7 python tools/bench_curve_programs.py 14 15 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l)
    if d['curve']=='p256' or d['program']=='curve_scalar_mul_windowed': print('$v', d['program'], d['curve'], d['n'], d['ms'], d['fills_per_s'])"
done
done
