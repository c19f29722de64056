#!/bin/bash
# SQ counters of every pipeline kernel in the DEFAULT launch plan (chains, inversions and expansions running beside each
# other), summed over the launches of one bench step: who is busy, who waits.  Counters only (no trace domains).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PY=$(readlink -f "$(which python3)")
rm -rf gpurun_out/pmcp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d gpurun_out/pmcp -o p -- $PY bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-limb-split --check 0 ${BENCH_ARGS} > /dev/null 2> gpurun_out/pmcp.err
python3 - <<'PYEOF'
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/pmcp/p_counter_collection.csv")))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if k.startswith("k_"):
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]) / 2      # warm-up + 1 step
print("per bench step, SQ_* cycle counters in quad-cycles; share of the kernel's SQ_WAVE_CYCLES in brackets")
for k, c in sorted(acc.items()):
    w = c["SQ_WAVE_CYCLES"] or 1
    print(f"{k:22s} waves {c['SQ_WAVES']:.3e}  wave_cycles {w:.3e}  active_any {c['SQ_ACTIVE_INST_ANY']:.3e} [{c['SQ_ACTIVE_INST_ANY']/w:.0%}]"
          f"  active_valu {c['SQ_ACTIVE_INST_VALU']:.3e} [{c['SQ_ACTIVE_INST_VALU']/w:.0%}]  wait_inst_any {c['SQ_WAIT_INST_ANY']:.3e} [{c['SQ_WAIT_INST_ANY']/w:.0%}]"
          f"  wait_any {c['SQ_WAIT_ANY']:.3e} [{c['SQ_WAIT_ANY']/w:.0%}]  insts_valu {c['SQ_INSTS_VALU']:.3e}")
PYEOF
tail -2 gpurun_out/pmcp.err
