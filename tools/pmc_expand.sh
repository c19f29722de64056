#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PY=$(readlink -f "$(which python3)")
export P2E_MSM_PIECES=1
rm -rf gpurun_out/pmc1 gpurun_out/pmc2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d gpurun_out/pmc1 -o p -- $PY bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/pmc1.err
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc2 -o p -- $PY bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/pmc2.err
python3 - <<'PYEOF'
import csv, collections
for d in ("pmc1", "pmc2"):
    try:
        rows = list(csv.DictReader(open(f"gpurun_out/{d}/p_counter_collection.csv")))
    except Exception as e:
        print(d, "missing", e); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("k_"):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(k, {c: f"{max(v):.3e}" for c, v in cs.items()})
PYEOF
tail -3 gpurun_out/pmc1.err
