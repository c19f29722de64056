#!/bin/bash
# where the mid-size plan hands over to the large-batch plan (P2E_BINV_ALT_MAX_N): one process per setting
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${TAG}_mid_large_threshold.txt
: > $OUT
for rep in 1 2; do
  for n in 32768 40960 49152 57344 65536; do
    for t in 0 1000000; do
      P2E_BINV_ALT_MAX_N=$t timeout -k 10 120 python tools/stream_order.py $n torch_first 15 2>&1 | grep "^n=" | sed "s/^/alt_max_n=$t /" | tee -a $OUT
    done
  done
done
