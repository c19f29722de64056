#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_r02d.log 2>&1; echo "pytest_exit=$?" >> gpurun_out/pytest_gpu_r02d.log; tail -12 gpurun_out/pytest_gpu_r02d.log
