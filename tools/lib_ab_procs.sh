#!/bin/bash
# two builds of the library, alternating PROCESSES (launch-plan / stream changes cannot be A/B-ed inside one process):
# tools/lib_ab_procs.sh TAG OTHER.so "N ..." [REPS]     (OTHER.so relative to the repo root, e.g. gpurun_exp_r02.so)
TAG=${1:-r03}; OTHER=$2; NS=${3:-"65536"}; REPS=${4:-4}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${TAG}_lib_ab.txt
: > $OUT
for rep in $(seq $REPS); do
  for n in $NS; do
    for mode in torch_first ctx_first; do
      timeout -k 10 120 python tools/stream_order.py $n $mode 15 2>&1 | grep "^n=" | sed "s/^/lib=HEAD /" | tee -a $OUT
      P2E_LIB=$GRAFT_REPO_ROOT/$OTHER timeout -k 10 120 python tools/stream_order.py $n $mode 15 2>&1 | grep "^n=" | sed "s/^/lib=$OTHER /" | tee -a $OUT
    done
  done
done
