#!/bin/bash
# 2^15: run expansion switched on below its default threshold, run lengths / pieces
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run() {
  label=$1; shift
  env "$@" timeout -k 10 120 python bench.py --steps 12 --warmup 3 --batch-log2 15 --no-cpu-baseline --no-limb-split --no-p256 --check 16 > gpurun_out/r15_$label.json 2> gpurun_out/r15_$label.err
  python -c "
import json; d=json.load(open('gpurun_out/r15_$label.json')); print('$label:', d['value'], d['ms_per_step'], d['median_step_ms'], d.get('checked_vs_oracle'), d['phase_ms_per_step'])"
}
for rep in 1 2; do
run default_$rep P2E_X=0
run runs9_$rep P2E_RUNS_MIN_N=32768
run runs9_p5_$rep P2E_RUNS_MIN_N=32768 P2E_MSM_PIECES=5
run runs18_p5_$rep P2E_RUNS_MIN_N=32768 P2E_RUN_ITERS=18 P2E_MSM_PIECES=5
run runs5_$rep P2E_RUNS_MIN_N=32768 P2E_RUN_ITERS=5 P2E_MSM_PIECES=8
done
