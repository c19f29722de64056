#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
{
for n in 8192 12288 16384 20480 24576 28672 32768 40960 48896 57344; do
  python tools/mid_sweep.py $n
  P2E_QUAD_MAX_N=1 python tools/mid_sweep.py $n
  P2E_QUAD_MAX_N=1 P2E_RUNS_MIN_N=1 P2E_MSM_PIECES=5 P2E_RUN_ITERS=9 python tools/mid_sweep.py $n
  P2E_QUAD_MAX_N=1 P2E_RUNS_MIN_N=1 P2E_MSM_PIECES=5 P2E_RUN_ITERS=12 python tools/mid_sweep.py $n
  P2E_QUAD_MAX_N=1 P2E_RUNS_MIN_N=1 P2E_MSM_PIECES=8 P2E_RUN_ITERS=9 python tools/mid_sweep.py $n
done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/mid_sweep2.log
cat gpurun_out/mid_sweep2.log
