"""Fused verify call at an arbitrary batch size with the context's knobs taken from the environment (one process per
setting): median ms of STEPS synchronous calls.  Usage: python tools/mid_sweep.py N [STEPS]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import plonky2_ecdsa_amd as p2e
n = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 15
sig = [torch.from_numpy(a).cuda() for a in p2e.synth_signatures(seed=4, n=n)]
ctx = p2e.Context(device=0)
ld = n + 16
cols = torch.empty((p2e.VERIFY_COLS, ld), dtype=torch.int64, device="cuda")
err = torch.empty(n, dtype=torch.uint8, device="cuda"); valid = torch.empty(n, dtype=torch.uint8, device="cuda")
call = lambda: ctx.ecdsa_verify_witness_batch(*sig, cols=cols[:, :n], err=err, valid=valid, ld=ld)[3]
for _ in range(3): call()
torch.cuda.synchronize()
ts = []
for _ in range(steps):
    t = time.perf_counter(); bad = call(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
ts.sort()
knobs = {k: v for k, v in os.environ.items() if k.startswith("P2E_")}
print(f"n={n} {knobs}: median {ts[len(ts)//2]:.3f} ms  min {ts[0]:.3f}  ({n/ts[len(ts)//2]*1e3/1e6:.2f} M fills/s) valid {int(valid.sum())} bad {bad}", flush=True)
