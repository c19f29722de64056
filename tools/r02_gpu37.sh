#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for v in alt noalt; do
  if [ $v = noalt ]; then export P2E_CP_NO_ALT_B=1; else unset P2E_CP_NO_ALT_B; fi
  REPS=7 python tools/bench_curve_programs.py 13 14 15 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l)
    print('$v', d['program'], d['curve'], d['n'], d['ms'], d['fills_per_s'], d['valid'])"
done
done
timeout -k 10 600 python -m pytest tests/test_curve_programs.py -m gpu -x -q 2>&1 | tail -3
