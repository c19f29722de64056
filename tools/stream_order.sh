#!/bin/bash
# creation-order sensitivity of the small / mid-size plans: timing (one process per setting, REPS repetitions each, with
# and without the fixed queue-binding order of p2e_ctx_create), then one rocprofv3 kernel trace per order at 2^13 with the
# hardware queue of every kernel.  Usage (gpurun): tools/stream_order.sh TAG [REPS]
TAG=${1:-r03}; REPS=${2:-3}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PY=$(readlink -f "$(which python3)")
OUT=gpurun_out/${TAG}_stream_order.txt
: > $OUT
for touch in 0 1; do
  for n in 8192 16384 32768; do
    for rep in $(seq $REPS); do
      for mode in ctx_first torch_first side_stream; do
        P2E_TOUCH_STREAMS=$touch timeout -k 10 120 python tools/stream_order.py $n $mode 2>&1 | grep "^n=" | tee -a $OUT
      done
    done
  done
done
for touch in 0 1; do
  for mode in ctx_first torch_first; do
    rm -rf gpurun_out/tl_$mode
    P2E_TOUCH_STREAMS=$touch rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_$mode -o tl -- $PY tools/stream_order.py 8192 $mode 3 > /dev/null 2> gpurun_out/tl_$mode.err
    echo "== 2^13, $mode, touch=$touch: last step, kernel / hardware queue / start / end (ms)" | tee -a gpurun_out/${TAG}_stream_order_timelines.txt
    python3 tools/print_timeline.py gpurun_out/tl_$mode/tl_kernel_trace.csv >> gpurun_out/${TAG}_stream_order_timelines.txt 2>&1
  done
done
tail -n 5 gpurun_out/${TAG}_stream_order_timelines.txt
