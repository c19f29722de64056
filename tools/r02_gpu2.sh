#!/bin/bash
# quad plan: parity on the GPU, timeline and proxies at 2^13..2^15, sweeps of the split and the threshold
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02b; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "ragged or golden or edge or tiny or replay or cfg3 or wide_and_tail" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-limb-split"
for k in 13 14 15; do
  timeout -k 10 120 $B --batch-log2 $k > $O/quad_$k.json 2>> $O/err.log
  P2E_QUAD_MAX_N=0 timeout -k 10 120 $B --batch-log2 $k > $O/lane_$k.json 2>> $O/err.log
  P2E_QUAD_MAX_N=100000 timeout -k 10 120 $B --batch-log2 $k > $O/quadforce_$k.json 2>> $O/err.log
done
for sp in 0 1 3; do P2E_BINV_SPLIT_LOG2=$sp timeout -k 10 120 $B --batch-log2 13 > $O/quad_13_split$sp.json 2>> $O/err.log; done
for pc in 5 12; do P2E_MSM_PIECES=$pc timeout -k 10 120 $B --batch-log2 13 > $O/quad_13_pieces$pc.json 2>> $O/err.log; done
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02b/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d["value"], d["ms_per_step"], d["phase_ms_per_step"])
    except Exception as e: print(f, "ERR", e)
P
bash tools/r02_timeline.sh 13 tl13q | head -60
