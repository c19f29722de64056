#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PY=$(readlink -f "$(which python3)")
P=${1:-6}; export P2E_FIXED_PIECES=${2:-1}
rm -rf gpurun_out/tl
P2E_MSM_PIECES=$P rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o tl -- $PY bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/tl.err
python3 tools/print_timeline.py gpurun_out/tl/tl_kernel_trace.csv
