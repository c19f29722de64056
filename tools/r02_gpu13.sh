#!/bin/bash
# fixed-base windows as one run (k_expand_fb_run) + run plan of the curve programs: parity, then A/B in separate processes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_r02c.log 2>&1; echo "pytest_exit=$?" >> gpurun_out/pytest_gpu_r02c.log; tail -5 gpurun_out/pytest_gpu_r02c.log
grep -q "pytest_exit=0" gpurun_out/pytest_gpu_r02c.log || exit 1
for rep in 1 2; do
  for v in fb nofb; do
    if [ $v = fb ]; then export P2E_FB_RUN=1; else unset P2E_FB_RUN; fi
    timeout -k 10 120 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-limb-split --check 0 > gpurun_out/ab_fbrun_${v}_$rep.json 2> gpurun_out/ab_fbrun_${v}_$rep.err
    python - <<PY
import json
d = json.load(open("gpurun_out/ab_fbrun_${v}_$rep.json"))
print("$v $rep", d["value"], d["ms_per_step"], d["median_step_ms"], d["roofline"]["frac"], (d.get("roofline_k_expand") or {}).get("frac"), (d.get("roofline_k_expand_fb_run") or {}).get("frac"), d["phase_ms_per_step"])
PY
  done
done
unset P2E_FB_RUN
timeout -k 10 400 python tools/bench_curve_programs.py 16 > gpurun_out/curve_programs_runs.jsonl 2> gpurun_out/curve_programs_runs.err; echo "exit=$?"
python - <<'PY'
import json
for l in open("gpurun_out/curve_programs_runs.jsonl"):
    d = json.loads(l)
    print(d["program"], d["curve"], d["n"], d["ms"], d["fills_per_s"], d["whole_fill_frac_hbm_peak"], d["roofline"]["frac"], {k: v["frac"] for k, v in d["roofline"]["per_kernel"].items()})
PY
