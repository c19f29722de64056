#!/bin/bash
# round 2, first GPU call: the whole GPU suite, the default bench line, one-GPU proxies of the strong-scaled shape
set -o pipefail
mkdir -p gpurun_out/r02e
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02e/pytest.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/r02e/pytest.log
tail -5 gpurun_out/r02e/pytest.log
timeout -k 10 300 python bench.py --steps 10 --warmup 2 > gpurun_out/r02e/bench.json 2> gpurun_out/r02e/bench.err; echo "bench rc $?"
for k in 13 14 15; do
  timeout -k 10 120 python bench.py --steps 20 --warmup 3 --batch-log2 $k --no-cpu-baseline --no-limb-split > gpurun_out/r02e/bench_log2_$k.json 2>> gpurun_out/r02e/bench.err; echo "bench $k rc $?"
done
cat gpurun_out/r02e/bench*.json | cut -c1-600
