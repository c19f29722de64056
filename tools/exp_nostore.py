"""Experiment behind DESIGN.md section 5 (iii): time the pipeline with an alternative build of libp2e_hip.so
(argv[1]), e.g. one whose column stores are predicated off (PairEmit::store_* guarded by an impossible value):
ALU-only time of every kernel.  Build such a variant from a scratch copy of csrc/, never commit it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import plonky2_ecdsa_amd as p2e
if len(sys.argv) > 1:
    p2e.LIB_PATH = sys.argv[1]
n = 1 << 16
sigs = p2e.synth_signatures(seed=4, n=n)
ctx = p2e.Context(device=0, phase_timing=True)
dev = [torch.from_numpy(a).cuda() for a in sigs]
ld = n + 16
big = torch.empty((p2e.VERIFY_COLS, ld), dtype=torch.int64, device="cuda")
err = torch.empty(n, dtype=torch.uint8, device="cuda"); valid = torch.empty(n, dtype=torch.uint8, device="cuda")
def step():
    return ctx.ecdsa_verify_witness_batch(*dev, cols=big[:, :n], err=err, valid=valid, ld=ld)
step(); step(); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 5
print(f"lib={p2e.LIB_PATH} ms/step={dt*1e3:.3f} phases={ctx.last_phase_ms()}")
