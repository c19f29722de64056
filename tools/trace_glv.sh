#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PY=$(readlink -f "$(which python3)")
rm -rf gpurun_out/tlg
REPS=2 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tlg -o tl -- $PY tools/bench_kernels.py glv > gpurun_out/tlg.out 2> gpurun_out/tlg.err
cat gpurun_out/tlg.out
python3 tools/print_timeline.py gpurun_out/tlg/tl_kernel_trace.csv
