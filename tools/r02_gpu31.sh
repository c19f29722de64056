#!/bin/bash
# does the creation order (torch's first CUDA work vs the context's streams) change the step time?  mid_sweep with the
# context created first / last
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cat > /tmp/order.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
import plonky2_ecdsa_amd as p2e
n = int(sys.argv[1]); first = sys.argv[2]
if first == "ctx":
    ctx = p2e.Context(device=0)
sig_h = p2e.synth_signatures(seed=4, n=n)
sig = [torch.from_numpy(a).cuda() for a in sig_h]
ld = n + 16
cols = torch.empty((p2e.VERIFY_COLS, ld), dtype=torch.int64, device="cuda")
err = torch.empty(n, dtype=torch.uint8, device="cuda"); valid = torch.empty(n, dtype=torch.uint8, device="cuda")
if first != "ctx":
    ctx = p2e.Context(device=0)
call = lambda: ctx.ecdsa_verify_witness_batch(*sig, cols=cols[:, :n], err=err, valid=valid, ld=ld)[3]
for _ in range(4): call()
torch.cuda.synchronize()
ts = []
for _ in range(15):
    t = time.perf_counter(); call(); ts.append((time.perf_counter() - t) * 1e3)
ts.sort()
print(f"n={n} created first: {first}: median {ts[7]:.3f} ms min {ts[0]:.3f}", flush=True)
PY
for n in 8192 16384 32768 65536; do for rep in 1 2; do for first in ctx torch; do python /tmp/order.py $n $first 2>&1 | grep "^n="; done; done; done
