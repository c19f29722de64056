#!/bin/bash
# mid-size plan (21 504 < n < 49 152): run length x pieces x inversion split, one process per setting
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${TAG}_mid_plan.txt
: > $OUT
for n in 32768 24576; do
  timeout -k 10 120 python tools/stream_order.py $n torch_first 15 2>&1 | grep "^n=" | sed "s/^/default /" | tee -a $OUT
  for R in 6 9 12 18; do
    for P in 5 7; do
      for S in 1 2; do
        P2E_RUN_ITERS_MID=$R P2E_MSM_PIECES_MID=$P P2E_BINV_MID_SPLIT_LOG2=$S timeout -k 10 120 python tools/stream_order.py $n torch_first 15 2>&1 | grep "^n=" | sed "s/^/R=$R P=$P S=$S /" | tee -a $OUT
      done
    done
  done
  timeout -k 10 120 python tools/stream_order.py $n torch_first 15 2>&1 | grep "^n=" | sed "s/^/default /" | tee -a $OUT
done
