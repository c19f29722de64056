#!/bin/bash
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
print(f"n={n} first={first} lazy={os.environ.get('P2E_LAZY_STREAMS','0')}{os.environ.get('P2E_LAZY_OWN_STREAM','0')}: median {ts[7]:.3f} ms min {ts[0]:.3f}", flush=True)
PY
for n in 8192 16384 32768; do for first in ctx torch; do for lazy in 0 1 2; do
  unset P2E_LAZY_STREAMS P2E_LAZY_OWN_STREAM; if [ $lazy = 1 ]; then export P2E_LAZY_STREAMS=1; fi; if [ $lazy = 2 ]; then export P2E_LAZY_STREAMS=1 P2E_LAZY_OWN_STREAM=1; fi
  python /tmp/order.py $n $first 2>&1 | grep "^n="
done; done; done
