#!/bin/bash
# four-lane plan of the curve programs: ops per chain piece / ops of the short last piece, one process per setting
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${TAG}_cp_quad_pieces.txt
: > $OUT
for rep in 1 2; do
for po in 45 55 66 83 110; do
  for to in 17 12 7; do
    P2E_CP_PIECE_OPS_QUAD=$po P2E_CP_TAIL_OPS_QUAD=$to REPS=9 timeout -k 10 200 python tools/bench_curve_programs.py 13 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l)
        if d['program']!='curve_scalar_mul': print('piece=$po tail=$to', d['program'], d['curve'], d['ms'])" | tee -a $OUT
  done
done
done
