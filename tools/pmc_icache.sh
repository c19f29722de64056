#!/bin/bash
# instruction-cache and issue counters of the pipeline kernels (counters only)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PY=$(readlink -f "$(which python3)")
for SET in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES" "SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_IFETCH SQ_INSTS_BRANCH"; do
rm -rf gpurun_out/pmci
rocprofv3 --pmc $SET --output-format csv -d gpurun_out/pmci -o p -- $PY bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-limb-split --check 0 ${BENCH_ARGS} > /dev/null 2> gpurun_out/pmci.err || { echo "set failed: $SET"; tail -3 gpurun_out/pmci.err; continue; }
python3 - <<'PYEOF'
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/pmci/p_counter_collection.csv")))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if k.startswith("k_"):
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]) / 2
for k, c in sorted(acc.items()):
    print(f"{k:22s} " + "  ".join(f"{n} {v:.3e}" for n, v in sorted(c.items())))
PYEOF
done
