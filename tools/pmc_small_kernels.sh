#!/bin/bash
# per-DISPATCH instruction-fetch counters of the short latency-bound kernels of one 2^13 step (counters only, no trace domains)
# Usage (gpurun): tools/pmc_small_kernels.sh TAG [N]
TAG=${1:-r03}; N=${2:-8192}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PY=$(readlink -f "$(which python3)")
OUT=gpurun_out/${TAG}_pmc_small_kernels.txt
: > $OUT
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_IFETCH SQ_IFETCH SQ_INSTS_BRANCH" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
rm -rf gpurun_out/pmcs
rocprofv3 --pmc $SET --output-format csv -d gpurun_out/pmcs -o p -- $PY tools/stream_order.py $N torch_first 2 > /dev/null 2> gpurun_out/pmcs.err || { echo "set failed: $SET" | tee -a $OUT; tail -3 gpurun_out/pmcs.err; continue; }
python3 - >> $OUT <<'PYEOF'
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/pmcs/p_counter_collection.csv")))
disp = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if k.startswith("k_"):
        d = disp.setdefault(int(r["Dispatch_Id"]), {"name": k, "grid": r["Grid_Size"]})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
ids = list(disp)
# the last step = the dispatches after the last k_finalize but one
fin = [i for i in ids if disp[i]["name"] == "k_finalize"]
start = fin[-2] if len(fin) > 1 else ids[0]
for i in ids:
    if i > start:
        d = disp[i]
        print(f"{d['name']:20s} grid {d['grid']:>8s}  " + "  ".join(f"{n} {v:.4g}" for n, v in d.items() if n not in ("name", "grid")))
print()
PYEOF
done
cat $OUT
