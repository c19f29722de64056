#!/bin/bash
# where the four-lane plan (round-3 form: short runs, front-loaded pieces) hands over to the mid-size plan
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${TAG}_quad_threshold.txt
: > $OUT
for rep in 1 2; do
  for n in ${NS:-12288 14336 16384 20480 24576}; do
    for q in 0 1000000; do
      P2E_QUAD_MAX_N=$q timeout -k 10 120 python tools/stream_order.py $n torch_first 15 2>&1 | grep "^n=" | sed "s/^/quad_max_n=$q /" | tee -a $OUT
    done
  done
done
