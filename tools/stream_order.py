"""Does the step time depend on WHO touched the GPU first?  One process per setting.
  python tools/stream_order.py N MODE [STEPS]     MODE: ctx_first | torch_first | side_stream
ctx_first: p2e.Context() before torch's first CUDA work; torch_first: after it; side_stream: the context runs on a
caller-created side stream.  Prints the median / min step in ms (synchronous calls) and, with P2E_ORDER_TRACE=1, runs a
single extra step at the end (for a rocprofv3 kernel trace of exactly that order)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import plonky2_ecdsa_amd as p2e
n = int(sys.argv[1]); mode = sys.argv[2]; steps = int(sys.argv[3]) if len(sys.argv) > 3 else 21
if mode == "ctx_first":
    ctx = p2e.Context(device=0)
sig_h = p2e.synth_signatures(seed=4, n=n)
sig = [torch.from_numpy(a).cuda() for a in sig_h]
ld = n + 16
cols = torch.empty((p2e.VERIFY_COLS, ld), dtype=torch.int64, device="cuda")
err = torch.empty(n, dtype=torch.uint8, device="cuda"); valid = torch.empty(n, dtype=torch.uint8, device="cuda")
if mode == "torch_first":
    ctx = p2e.Context(device=0)
elif mode == "side_stream":
    st = torch.cuda.Stream()
    ctx = p2e.Context(device=0, stream=st.cuda_stream)
torch.cuda.synchronize()
call = lambda: ctx.ecdsa_verify_witness_batch(*sig, cols=cols[:, :n], err=err, valid=valid, ld=ld)[3]
for _ in range(4): call()
torch.cuda.synchronize()
ts = []
for _ in range(steps):
    t = time.perf_counter(); call(); ts.append((time.perf_counter() - t) * 1e3)
ts.sort()
touch = os.environ.get("P2E_TOUCH_STREAMS", "1")
print(f"n={n} order={mode} touch={touch}: median {ts[len(ts) // 2]:.3f} ms min {ts[0]:.3f} valid {int(valid.sum())}", flush=True)
