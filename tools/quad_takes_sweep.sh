#!/bin/bash
# four-lane plan: runs (of 4 iterations) per loop piece, front-loaded so that the last piece -- whose inversion batch and
# expansion are the exposed tail -- is short (P2E_SMALL_TAKES, the last piece takes the rest of the 19 runs)
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${TAG}_quad_takes.txt
: > $OUT
for rep in $(seq ${REPS:-2}); do
  for n in 8192 12288; do
    timeout -k 10 120 python tools/stream_order.py $n torch_first 15 2>&1 | grep "^n=" | sed "s/^/takes=default /" | tee -a $OUT
    for takes in ${TAKES_LIST:-6,6,5 6,6,4 7,6,4 7,6,5 6,5,5,2 5,5,4,3 8,7}; do
      P2E_SMALL_TAKES=$takes timeout -k 10 120 python tools/stream_order.py $n torch_first 15 2>&1 | grep "^n=" | sed "s/^/takes=$takes /" | tee -a $OUT
    done
  done
done
