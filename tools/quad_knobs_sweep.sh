#!/bin/bash
# four-lane plan with the round-3 defaults (short runs, own expansion stream): LDS cap of the expansion kernels x inversion
# split, 2^13 per call, one process per setting
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${TAG}_quad_knobs.txt
: > $OUT
for rep in $(seq ${REPS:-2}); do
  for lds in 160000 80000 54000 0; do
    for split in 2 1 3; do
      for last in 3 2; do
        P2E_EXPAND_LDS_SMALL=$lds P2E_BINV_SPLIT_LOG2=$split P2E_BINV_SPLIT_LOG2_LAST=$last timeout -k 10 120 python tools/stream_order.py 8192 torch_first 15 2>&1 | grep "^n=" | sed "s/^/lds=$lds split=$split last=$last /" | tee -a $OUT
      done
    done
  done
done
