#!/bin/bash
# kernel timeline (rocprofv3 --kernel-trace) of the last step of tools/stream_order.py at N signatures per call.
# Usage (gpurun): tools/timeline_small.sh TAG N [MODE]
TAG=${1:-r03}; N=${2:-8192}; MODE=${3:-torch_first}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PY=$(readlink -f "$(which python3)")
rm -rf gpurun_out/tl_small
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_small -o tl -- $PY tools/stream_order.py $N $MODE 3 > /dev/null 2> gpurun_out/tl_small.err
echo "== n=$N, $MODE: last step, kernel / hardware queue / start / end (ms)" > gpurun_out/${TAG}_timeline_${N}.txt
python3 tools/print_timeline.py gpurun_out/tl_small/tl_kernel_trace.csv >> gpurun_out/${TAG}_timeline_${N}.txt 2>&1
cat gpurun_out/${TAG}_timeline_${N}.txt
