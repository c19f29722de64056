#!/bin/bash
# large-batch plan, 2^16 per call: round-2 stream placement (MFB2) against the round-3 default (M1FB2), alternating processes
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${TAG}_layout_ab_2e16.txt
: > $OUT
for rep in 1 2 3 4; do
  for layout in MFB2 M1FB2; do
    for mode in ctx_first torch_first; do
      P2E_STREAM_LAYOUT=$layout timeout -k 10 120 python tools/stream_order.py 65536 $mode 15 2>&1 | grep "^n=" | sed "s/^/layout=$layout /" | tee -a $OUT
    done
  done
done
