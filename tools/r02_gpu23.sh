#!/bin/bash
# mid-size batches (2^15 and 40 000): phase B alternating between two streams, optionally split
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run() {  # label, batch-log2, env...
  label=$1; lg=$2; shift; shift
  env "$@" timeout -k 10 120 python bench.py --steps 12 --warmup 3 --batch-log2 $lg --no-cpu-baseline --no-limb-split --no-p256 --check 16 > gpurun_out/mid_$label.json 2> gpurun_out/mid_$label.err
  python -c "
import json; d=json.load(open('gpurun_out/mid_$label.json')); print('$label:', d['value'], d['ms_per_step'], d['median_step_ms'], d.get('checked_vs_oracle'), d['phase_ms_per_step'])"
}
for rep in 1 2; do
run off15_$rep 15 P2E_BINV_ALT_MAX_N=0
run alt15_$rep 15 P2E_X=0
run alt15_split1_$rep 15 P2E_BINV_MID_SPLIT_LOG2=1
run alt15_split2_$rep 15 P2E_BINV_MID_SPLIT_LOG2=2
done
run off16 16 P2E_BINV_ALT_MAX_N=0
run alt16 16 P2E_BINV_ALT_MAX_N=100000
run alt16_split1 16 P2E_BINV_ALT_MAX_N=100000 P2E_BINV_MID_SPLIT_LOG2=1
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_size or ragged or run_expansion" 2>&1 | tail -3
