#!/bin/bash
# four-lane plan with the MSM loop expanded as RUNS of R iterations (P2E_RUNS_MIN_N=0, P2E_RUN_ITERS=R): phase B then skips
# the Jacobian -> affine conversion of the results inside a run (3 instead of 8 multiplications per op).  One process per setting.
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${TAG}_quad_runs.txt
: > $OUT
for rep in 1 2; do
for n in 8192 12288; do
  timeout -k 10 120 python tools/stream_order.py $n torch_first 15 2>&1 | grep "^n=" | sed "s/^/default /" | tee -a $OUT
  for pieces in 5 7; do
    for R in 2 3 4 6; do
      P2E_RUNS_MIN_N=0 P2E_RUN_ITERS=$R P2E_MSM_PIECES_SMALL=$pieces timeout -k 10 120 python tools/stream_order.py $n torch_first 15 2>&1 | grep "^n=" | sed "s/^/runs R=$R pieces=$pieces /" | tee -a $OUT
    done
  done
done
done
