#!/bin/bash
# the four-lane plan above its threshold: HEAD against a build whose k_chains_quad is capped at 256 VGPRs (gpurun_exp_cap256.so), against the mid-size plan
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03ah_cap_ab.txt; : > $OUT
for rep in 1 2 3; do
 for n in 24576 32768; do
  P2E_QUAD_MAX_N=0 timeout -k 10 120 python tools/stream_order.py $n torch_first 15 2>&1 | grep "^n=" | sed "s/^/plan=mid lib=HEAD /" >> $OUT
  P2E_QUAD_MAX_N=1000000 timeout -k 10 120 python tools/stream_order.py $n torch_first 15 2>&1 | grep "^n=" | sed "s/^/plan=quad lib=HEAD /" >> $OUT
  P2E_LIB=$GRAFT_REPO_ROOT/gpurun_exp_cap256.so P2E_QUAD_MAX_N=1000000 timeout -k 10 120 python tools/stream_order.py $n torch_first 15 2>&1 | grep "^n=" | sed "s/^/plan=quad lib=cap256 /" >> $OUT
 done
done
python3 - <<PY
import re,collections,statistics
d=collections.defaultdict(list)
for l in open("$OUT"):
    m=re.match(r"(plan=\S+ lib=\S+) n=(\d+) .*median ([\d.]+)",l)
    if m: d[(int(m.group(2)),m.group(1))].append(float(m.group(3)))
for k in sorted(d): print(k, d[k], round(statistics.mean(d[k]),3))
PY
