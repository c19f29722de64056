#!/bin/bash
# four-lane plan (2^13 per call): expansion streams restricted to k compute units (P2E_QUAD_EXPAND_CUS), with and without
# the dynamic-LDS cap of the expansion kernels (P2E_EXPAND_LDS_SMALL); one process per setting
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${TAG}_quad_cu_mask.txt
: > $OUT
for rep in 1 2; do
for n in 8192 12288; do
  for lds in 160000 0; do
    for pat in low stride; do
      for k in 0 224 192 160 128 96; do
        [ $k = 0 ] && [ $pat = stride ] && continue
        P2E_QUAD_EXPAND_CUS=$k P2E_QUAD_EXPAND_CU_PATTERN=$pat P2E_EXPAND_LDS_SMALL=$lds timeout -k 10 120 python tools/stream_order.py $n torch_first 15 2>&1 | grep "^n=" | sed "s/^/cus=$k pat=$pat lds=$lds /" | tee -a $OUT
      done
    done
  done
done
done
