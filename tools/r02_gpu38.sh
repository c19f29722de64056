#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cat > /tmp/side.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
import plonky2_ecdsa_amd as p2e
n = int(sys.argv[1]); mode = sys.argv[2]
if mode == "own_first":
    ctx = p2e.Context(device=0)
sig = [torch.from_numpy(a).cuda() for a in p2e.synth_signatures(seed=4, n=n)]
ld = n + 16
cols = torch.empty((p2e.VERIFY_COLS, ld), dtype=torch.int64, device="cuda")
err = torch.empty(n, dtype=torch.uint8, device="cuda"); valid = torch.empty(n, dtype=torch.uint8, device="cuda")
if mode == "own_last":
    ctx = p2e.Context(device=0)
if mode == "side":
    st = torch.cuda.Stream()
    ctx = p2e.Context(device=0, stream=st.cuda_stream)
if mode == "side_hi":
    st = torch.cuda.Stream(priority=-1)
    ctx = p2e.Context(device=0, stream=st.cuda_stream)
torch.cuda.synchronize()
call = lambda: ctx.ecdsa_verify_witness_batch(*sig, cols=cols[:, :n], err=err, valid=valid, ld=ld)[3]
for _ in range(4): call()
torch.cuda.synchronize()
ts = []
for _ in range(15):
    t = time.perf_counter(); bad = call(); ts.append((time.perf_counter() - t) * 1e3)
ts.sort()
print(f"n={n} {mode}: median {ts[7]:.3f} ms min {ts[0]:.3f} valid {int(valid.sum())}", flush=True)
PY
for n in 8192 16384 32768 65536; do for mode in own_first own_last side side_hi; do python /tmp/side.py $n $mode 2>&1 | grep "^n="; done; done
