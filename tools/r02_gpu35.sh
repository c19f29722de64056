#!/bin/bash
# final library: GPU suite, then old thresholds against new ones in alternating processes (same box), then the proxies
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_r02.log 2>&1; echo "pytest_exit=$?" >> gpurun_out/pytest_gpu_r02.log; tail -4 gpurun_out/pytest_gpu_r02.log
grep -q "pytest_exit=0" gpurun_out/pytest_gpu_r02.log || exit 1
{
echo "old = P2E_QUAD_MAX_N=24576 P2E_RUNS_MIN_N=49152 P2E_BINV_ALT_MAX_N=0 (the plan thresholds before this change), new = defaults"
for n in 8192 16384 24576 32768 40960 48896 65536; do
  P2E_QUAD_MAX_N=24576 P2E_RUNS_MIN_N=49152 P2E_BINV_ALT_MAX_N=0 python tools/mid_sweep.py $n 2>&1 | grep "^n=" | sed 's/{.*}/old/; s/valid.*//'
  python tools/mid_sweep.py $n 2>&1 | grep "^n=" | sed 's/{}/new/; s/valid.*//'
done
} > gpurun_out/plan_thresholds_ab.log
cat gpurun_out/plan_thresholds_ab.log
for lg in 13 14 15; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --batch-log2 $lg --no-limb-split --no-p256 --no-cpu-baseline > gpurun_out/proxy_$lg.json 2> gpurun_out/proxy_$lg.err
  python -c "
import json; d=json.load(open('gpurun_out/proxy_$lg.json')); print('2^$lg:', d['value'], d['ms_per_step'], d['median_step_ms'], d.get('checked_vs_oracle'))"
done
