#!/bin/bash
# four-lane plan after the chains moved to lazy limbs: run length of the expansion runs x loop pieces (equal pieces unless
# the run length is 4 and P2E_SMALL_TAKES applies).  Usage (gpurun): RUNS="0 2 4" PIECES="5 6" NS="8192" REPS=2 tools/quad_sweep3.sh TAG
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${TAG}_quad_sweep3.txt
: > $OUT
for rep in $(seq ${REPS:-2}); do
  for n in ${NS:-8192}; do
    for r in ${RUNS:-4}; do
      for pc in ${PIECES:-6}; do
        P2E_RUN_ITERS_SMALL=$r P2E_MSM_PIECES_SMALL=$pc timeout -k 10 120 python tools/stream_order.py $n torch_first 15 2>&1 | grep "^n=" | sed "s/^/run_iters=$r pieces=$pc /" >> $OUT
      done
    done
  done
done
python3 - <<PY
import re,collections,statistics
d=collections.defaultdict(list)
for l in open("$OUT"):
    m=re.match(r"(run_iters=\S+ pieces=\S+) n=(\d+) .*median ([\d.]+)",l)
    if m: d[(int(m.group(2)),m.group(1))].append(float(m.group(3)))
for k in sorted(d, key=lambda k:(k[0],statistics.mean(d[k]))): print(k[0], k[1], d[k], round(statistics.mean(d[k]),3))
PY
