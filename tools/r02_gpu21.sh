#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for rep in 1 2; do
for v in default quad2w; do
  if [ $v = quad2w ]; then export P2E_LIB=$GRAFT_REPO_ROOT/tools/ab_build/libp2e_quad2w.so P2E_QUAD_MAX_N=32768; else unset P2E_LIB P2E_QUAD_MAX_N; fi
  timeout -k 10 120 python bench.py --steps 12 --warmup 3 --batch-log2 15 --no-cpu-baseline --no-limb-split --no-p256 --check 16 > gpurun_out/q2w_${v}_$rep.json 2> gpurun_out/q2w_${v}_$rep.err
  python - <<PY
import json
d = json.load(open("gpurun_out/q2w_${v}_$rep.json"))
print("$v rep $rep:", d["value"], d["ms_per_step"], d["median_step_ms"], d.get("checked_vs_oracle"), d["phase_ms_per_step"])
PY
done
done
unset P2E_LIB P2E_QUAD_MAX_N
# the same variant at 2^13 / 2^14 (one wave per SIMD there either way: the cap must not cost anything)
for lg in 13 14; do
for v in default quad2w; do
  if [ $v = quad2w ]; then export P2E_LIB=$GRAFT_REPO_ROOT/tools/ab_build/libp2e_quad2w.so; else unset P2E_LIB; fi
  timeout -k 10 120 python bench.py --steps 20 --warmup 5 --batch-log2 $lg --no-cpu-baseline --no-limb-split --no-p256 --check 0 > gpurun_out/q2w_${v}_lg$lg.json 2> /dev/null
  python -c "
import json; d=json.load(open('gpurun_out/q2w_${v}_lg$lg.json')); print('$v 2^$lg:', d['value'], d['ms_per_step'], d['median_step_ms'])"
done
done
