#!/bin/bash
# kernel timeline of one curve-program call: tools/curve_timeline.sh <kind 1..3> <curve 0..1> <log2 n>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PY=$(readlink -f "$(which python3)")
rm -rf gpurun_out/ctl
cat > /tmp/ctl_run.py <<PYEOF
import sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import torch, plonky2_ecdsa_amd as p2e
kind, curve, lg = $1, $2, $3
n = 1 << lg
ctx = p2e.Context(device=0)
b = p2e.synth_signatures_curve(curve, seed=777, n=1)
blind = (int.from_bytes(bytes(b[3][0]), "little"), int.from_bytes(bytes(b[4][0]), "little"))
prog = p2e.CurveProgram(ctx, kind, curve, blind)
sig = [torch.from_numpy(a).cuda() for a in p2e.synth_signatures_curve(curve, seed=5, n=n)]
cols = torch.empty((prog.num_cols, n + 16), dtype=torch.int64, device="cuda")
for _ in range(3):
    if kind == 3:
        prog.verify_witness_batch(*sig, cols=cols[:, :n], ld=n + 16)
    else:
        prog.mul_witness_batch(sig[3], sig[4], sig[0], cols=cols[:, :n], ld=n + 16)
    torch.cuda.synchronize()
PYEOF
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ctl -o ctl -- $PY /tmp/ctl_run.py > /dev/null 2> gpurun_out/ctl.err
python3 - <<'PYEOF'
import csv
rows = list(csv.DictReader(open("gpurun_out/ctl/ctl_kernel_trace.csv")))
rows = [r for r in rows if "kc_" in r["Kernel_Name"] or "k_finalize" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if "kc_scalar" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    nm = r["Kernel_Name"].replace("void ", "").split("(")[0]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    print(f"{nm:34s} q={r['Queue_Id']:>3s} start={s:8.3f} end={e:8.3f} dur={e - s:7.3f} grid={r['Grid_Size_X']}x{r['Grid_Size_Y']}")
PYEOF
