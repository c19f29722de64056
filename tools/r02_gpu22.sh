#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run() {  # label, env...
  label=$1; shift
  env "$@" timeout -k 10 120 python bench.py --steps 12 --warmup 3 --batch-log2 15 --no-cpu-baseline --no-limb-split --no-p256 --check 16 > gpurun_out/q15x_$label.json 2> gpurun_out/q15x_$label.err
  python -c "
import json; d=json.load(open('gpurun_out/q15x_$label.json')); print('$label:', d['value'], d['ms_per_step'], d['median_step_ms'], d.get('checked_vs_oracle'), d['phase_ms_per_step'])"
}
for rep in 1 2; do
run default_$rep P2E_X=0
run quad_nolds_$rep P2E_QUAD_MAX_N=32768 P2E_EXPAND_LDS_SMALL=0
run quad_lds64k_$rep P2E_QUAD_MAX_N=32768 P2E_EXPAND_LDS_SMALL=65536
run quad2w_nolds_$rep P2E_LIB=$GRAFT_REPO_ROOT/tools/ab_build/libp2e_quad2w.so P2E_QUAD_MAX_N=32768 P2E_EXPAND_LDS_SMALL=0
run quad_nolds_split1_$rep P2E_QUAD_MAX_N=32768 P2E_EXPAND_LDS_SMALL=0 P2E_BINV_SPLIT_LOG2=1 P2E_BINV_SPLIT_LOG2_LAST=2
run quad_nolds_p8_$rep P2E_QUAD_MAX_N=32768 P2E_EXPAND_LDS_SMALL=0 P2E_MSM_PIECES_SMALL=8 P2E_FIXED_PIECES_SMALL=2
done
