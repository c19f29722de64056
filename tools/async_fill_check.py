"""Step time of the fused verifier through an ASYNCHRONOUS context on a caller-created side stream (issue + p2e_sync),
against a synchronous context, same process.  Usage: python tools/async_fill_check.py N"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import plonky2_ecdsa_amd as p2e
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
sig = [torch.from_numpy(a).cuda() for a in p2e.synth_signatures(seed=4, n=n)]
ld = n + 16
cols = torch.empty((p2e.VERIFY_COLS, ld), dtype=torch.int64, device="cuda")
err = torch.empty(n, dtype=torch.uint8, device="cuda"); valid = torch.empty(n, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
def bench(ctx, label, sync):
    call = lambda: ctx.ecdsa_verify_witness_batch(*sig, cols=cols[:, :n], err=err, valid=valid, ld=ld)[3]
    ts = []
    for k in range(12):
        t = time.perf_counter(); call()
        if sync: ctx.sync()
        ts.append((time.perf_counter() - t) * 1e3)
    print(label, " ".join(f"{x:.2f}" for x in ts), flush=True)
bench(p2e.Context(device=0), "sync ctx, library stream:", False)
st = torch.cuda.Stream()
bench(p2e.Context(device=0, stream=st.cuda_stream), "sync ctx, side stream:   ", False)
bench(p2e.Context(device=0, stream=st.cuda_stream, asynchronous=True), "async ctx, side stream:  ", True)
st2 = torch.cuda.Stream()
c2 = p2e.Context(device=0, stream=st2.cuda_stream, asynchronous=True)
bench(c2, "async ctx, fresh stream: ", True)
print("segments:", c2.segments())
