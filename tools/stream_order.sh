#!/bin/bash
# creation-order sensitivity of the small / mid-size plans: timing (one process per setting, REPS repetitions each, for the
# round-2 stream placement (MFB2: expansions on the caller's stream) and the round-3 default (M1FB2)), then one rocprofv3 kernel trace per order at 2^13 with the
# hardware queue of every kernel.  Usage (gpurun): tools/stream_order.sh TAG [REPS]
TAG=${1:-r03}; REPS=${2:-3}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PY=$(readlink -f "$(which python3)")
OUT=gpurun_out/${TAG}_stream_order.txt
: > $OUT
for layout in MFB2 M1FB2; do
  export P2E_STREAM_LAYOUT=$layout
  for n in 8192 16384 32768; do
    for rep in $(seq $REPS); do
      for mode in ctx_first torch_first side_stream; do
        timeout -k 10 120 python tools/stream_order.py $n $mode 2>&1 | grep "^n=" | sed "s/^/layout=$layout /" | tee -a $OUT
      done
    done
  done
done
for layout in MFB2 M1FB2; do
  export P2E_STREAM_LAYOUT=$layout
  for mode in ctx_first torch_first; do
    rm -rf gpurun_out/tl_$mode
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_$mode -o tl -- $PY tools/stream_order.py 8192 $mode 3 > /dev/null 2> gpurun_out/tl_$mode.err
    echo "== 2^13, $mode, P2E_STREAM_LAYOUT=$layout: last step, kernel / hardware queue / start / end (ms)" | tee -a gpurun_out/${TAG}_stream_order_timelines.txt
    python3 tools/print_timeline.py gpurun_out/tl_$mode/tl_kernel_trace.csv >> gpurun_out/${TAG}_stream_order_timelines.txt 2>&1
  done
done
tail -n 5 gpurun_out/${TAG}_stream_order_timelines.txt
