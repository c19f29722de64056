#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_curve_programs.py -m gpu -x -q > gpurun_out/pytest_curves2.log 2>&1; echo "curves_exit=$?" >> gpurun_out/pytest_curves2.log; tail -4 gpurun_out/pytest_curves2.log
grep -q "curves_exit=0" gpurun_out/pytest_curves2.log || exit 1
timeout -k 10 400 python tools/bench_curve_programs.py 13 16 > gpurun_out/curve_programs_solinas.jsonl 2> gpurun_out/curve_programs_solinas.err; echo "exit=$?"
python - <<'PY'
import json
for l in open("gpurun_out/curve_programs_solinas.jsonl"):
    d = json.loads(l)
    print(d["program"], d["curve"], d["n"], d["ms"], d["fills_per_s"], d["whole_fill_frac_hbm_peak"], d["roofline"]["frac"], {k: v["frac"] for k, v in d["roofline"]["per_kernel"].items()})
PY
bash tools/curve_timeline.sh 3 1 16 > gpurun_out/ctl_p256_verify_16b.txt 2>&1; head -30 gpurun_out/ctl_p256_verify_16b.txt
