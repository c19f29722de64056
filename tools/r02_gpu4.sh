#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02d; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "ragged or golden or edge or tiny or replay or cfg3 or wide_and_tail" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
B="python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-limb-split"
for rep in 1 2; do for k in 13 14; do timeout -k 10 120 $B --batch-log2 $k > $O/quad_${k}_rep$rep.json 2>> $O/err.log; done; done
for pc in 3 4 6 8; do P2E_MSM_PIECES_SMALL=$pc timeout -k 10 120 $B --batch-log2 13 > $O/quad_13_pieces$pc.json 2>> $O/err.log; done
for fp in 2; do P2E_FIXED_PIECES_SMALL=$fp timeout -k 10 120 $B --batch-log2 13 > $O/quad_13_fixed$fp.json 2>> $O/err.log; done
for sp in 1 3; do P2E_BINV_SPLIT_LOG2=$sp timeout -k 10 120 $B --batch-log2 13 > $O/quad_13_split$sp.json 2>> $O/err.log; done
P2E_MSM_PIECES_SMALL=4 timeout -k 10 120 $B --batch-log2 14 > $O/quad_14_pieces4.json 2>> $O/err.log
P2E_MSM_PIECES_SMALL=8 timeout -k 10 120 $B --batch-log2 14 > $O/quad_14_pieces8.json 2>> $O/err.log
timeout -k 10 120 $B --batch-log2 16 > $O/big_16.json 2>> $O/err.log
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02d/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d["value"], d["ms_per_step"], d.get("median_step_ms"), d["phase_ms_per_step"])
    except Exception as e: print(f, "ERR", e)
P
bash tools/r02_timeline.sh 13 tl13q3 | head -60
