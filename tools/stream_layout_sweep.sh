#!/bin/bash
# P2E_STREAM_LAYOUT sweep: creation (= hardware-queue binding) order of the library's streams, with and without an own
# first expansion stream ('1'), spare streams ('P'); both caller orders; one process per setting.
# Usage (gpurun): tools/stream_layout_sweep.sh TAG "LAYOUT ..." "N ..." [REPS]
TAG=${1:-r03}; LAYOUTS=${2:-"MFB2 1MFB2 12MFB 1M2FB M1FB2 MF1B2 MFB12 MFB21 P1MFB2 PP1MFB2 1MFBP2 1MFPB2"}; NS=${3:-"8192 16384"}; REPS=${4:-2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${TAG}_stream_layout_sweep.txt
: > $OUT
for rep in $(seq $REPS); do
  for n in $NS; do
    for layout in $LAYOUTS; do
      for mode in ctx_first torch_first; do
        P2E_STREAM_LAYOUT=$layout timeout -k 10 120 python tools/stream_order.py $n $mode 15 2>&1 | grep "^n=" | sed "s/^/layout=$layout /" | tee -a $OUT
      done
    done
  done
done
