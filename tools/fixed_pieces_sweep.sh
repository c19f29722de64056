#!/bin/bash
# large-batch plan: number of fixed-base chain pieces (P2E_FIXED_PIECES; more pieces = smaller first piece = earlier first
# expansion) x MSM pieces, 2^16 per call, one process per setting
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${TAG}_fixed_pieces.txt
: > $OUT
for rep in 1 2 3; do
  for fp in 2 3 4 6 8; do
    P2E_FIXED_PIECES=$fp timeout -k 10 120 python tools/stream_order.py 65536 torch_first 15 2>&1 | grep "^n=" | sed "s/^/fixed_pieces=$fp /" | tee -a $OUT
  done
done
