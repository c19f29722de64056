#!/bin/bash
# kernel timeline of one fused call at batch 2^K (default 13): rocprofv3 --kernel-trace, last step printed
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PY=$(readlink -f "$(which python3)")
K=${1:-13}; TAG=${2:-tl13}
rm -rf gpurun_out/$TAG
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$TAG -o tl -- $PY bench.py --steps 2 --warmup 1 --batch-log2 $K --no-cpu-baseline --no-limb-split --check 0 > gpurun_out/$TAG.json 2> gpurun_out/$TAG.err
python3 tools/print_timeline.py gpurun_out/$TAG/tl_kernel_trace.csv > gpurun_out/$TAG.txt
rm -rf gpurun_out/$TAG
cat gpurun_out/$TAG.txt
