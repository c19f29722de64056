#!/bin/bash
# one environment knob, on / off, alternating processes: tools/env_ab.sh TAG "NAME=A NAME=B ..." "N ..." [REPS]
TAG=${1:-r03}; SETTINGS=$2; NS=${3:-65536}; REPS=${4:-3}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${TAG}_env_ab.txt
: > $OUT
for rep in $(seq $REPS); do
  for n in $NS; do
    for kv in $SETTINGS; do
      env ${kv//,/ } timeout -k 10 120 python tools/stream_order.py $n torch_first 15 2>&1 | grep "^n=" | sed "s/^/$kv /" >> $OUT
    done
  done
done
python3 - <<PY
import re,collections,statistics
d=collections.defaultdict(list)
for l in open("$OUT"):
    m=re.match(r"(\S+) n=(\d+) .*median ([\d.]+)",l)
    if m: d[(int(m.group(2)),m.group(1))].append(float(m.group(3)))
for k in sorted(d, key=lambda k:(k[0],statistics.mean(d[k]))): print(k[0], k[1], d[k], round(statistics.mean(d[k]),3))
PY
