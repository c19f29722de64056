"""How long does the HOST need to enqueue one fused call (asynchronous context: the call returns when everything is in the
queues), against the time the GPU needs to run it?  python tools/host_enqueue_time.py N [STEPS]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import plonky2_ecdsa_amd as p2e
n = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 21
sig = [torch.from_numpy(a).cuda() for a in p2e.synth_signatures(seed=4, n=n)]
ld = n + 16
cols = torch.empty((p2e.VERIFY_COLS, ld), dtype=torch.int64, device="cuda")
err = torch.empty(n, dtype=torch.uint8, device="cuda"); valid = torch.empty(n, dtype=torch.uint8, device="cuda")
st = torch.cuda.Stream()
ctx = p2e.Context(device=0, stream=st.cuda_stream, asynchronous=True)
torch.cuda.synchronize()
enq, tot = [], []
for k in range(steps + 4):
    t0 = time.perf_counter()
    ctx.ecdsa_verify_witness_batch(*sig, cols=cols[:, :n], err=err, valid=valid, ld=ld)
    t1 = time.perf_counter()
    assert ctx.sync() == 0
    t2 = time.perf_counter()
    if k >= 4:
        enq.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
enq.sort(); tot.sort()
print(f"n={n}: host enqueue median {enq[len(enq)//2]:.3f} ms (min {enq[0]:.3f}), whole call median {tot[len(tot)//2]:.3f} ms (min {tot[0]:.3f}), valid {int(valid.sum())}")
