#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
{
for n in 32768 26000 40000 49151; do
  python tools/mid_sweep.py $n
  for p in 4 5 6; do for r in 9 12; do
    P2E_RUNS_MIN_N=1 P2E_MSM_PIECES=$p P2E_RUN_ITERS=$r python tools/mid_sweep.py $n
  done; done
done
for p in 5 6 7 8; do P2E_MSM_PIECES=$p python tools/mid_sweep.py 65536; done
P2E_RUNS_MIN_N=1 P2E_QUAD_MAX_N=1 P2E_MSM_PIECES=5 python tools/mid_sweep.py 16384
P2E_RUNS_MIN_N=1 P2E_QUAD_MAX_N=1 P2E_MSM_PIECES=5 python tools/mid_sweep.py 24576
python tools/mid_sweep.py 24576
} 2>&1 | grep -v amdgpu.ids > gpurun_out/mid_sweep.log
cat gpurun_out/mid_sweep.log
