#!/bin/bash
# curve programs on the GPU for the first time: their tests, then the whole GPU suite, then the headline bench
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_curve_programs.py -m gpu -x -q > gpurun_out/pytest_curves.log 2>&1; echo "curves_exit=$?" >> gpurun_out/pytest_curves.log; tail -15 gpurun_out/pytest_curves.log
grep -q "curves_exit=0" gpurun_out/pytest_curves.log || exit 1
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_r02b.log 2>&1; echo "pytest_exit=$?" >> gpurun_out/pytest_gpu_r02b.log; tail -5 gpurun_out/pytest_gpu_r02b.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_r02b.json 2> gpurun_out/bench_r02b.err; echo "bench_exit=$?"; cat gpurun_out/bench_r02b.json
