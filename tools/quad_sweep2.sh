#!/bin/bash
# four-lane plan after the chains moved to lazy limbs: LDS cap of the expansion kernels x runs per loop piece x inversion
# split, one process per setting.  Usage (gpurun): LDS="54000 40000" TAKES="default 6,6,5" SPLITS="2" NS="8192" REPS=2 tools/quad_sweep2.sh TAG
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${TAG}_quad_sweep2.txt
: > $OUT
for rep in $(seq ${REPS:-2}); do
  for n in ${NS:-8192}; do
    for lds in ${LDS:-54000}; do
      for split in ${SPLITS:-2}; do
        for takes in ${TAKES:-default}; do
          if [ "$takes" = default ]; then unset P2E_SMALL_TAKES; else export P2E_SMALL_TAKES=$takes; fi
          P2E_EXPAND_LDS_SMALL=$lds P2E_BINV_SPLIT_LOG2=$split timeout -k 10 120 python tools/stream_order.py $n torch_first 15 2>&1 | grep "^n=" | sed "s/^/lds=$lds split=$split takes=$takes /" >> $OUT
        done
      done
    done
  done
done
python3 - <<PY
import re,collections,statistics
d=collections.defaultdict(list)
for l in open("$OUT"):
    m=re.match(r"(lds=\S+ split=\S+ takes=\S+) n=(\d+) .*median ([\d.]+)",l)
    if m: d[(int(m.group(2)),m.group(1))].append(float(m.group(3)))
for k in sorted(d, key=lambda k:(k[0],statistics.mean(d[k]))): print(k[0], k[1], d[k], round(statistics.mean(d[k]),3))
PY
