import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import plonky2_ecdsa_amd as p2e
print("queues env", os.environ.get("GPU_MAX_HW_QUEUES"))
ctx = p2e.Context(device=0)
for n in (1024, 65536):
    sig = p2e.synth_signatures(seed=3, n=n)
    px, py, k = [torch.from_numpy(a).cuda() for a in (sig[3], sig[4], sig[0])]
    ld = n + 16
    cols = torch.empty((p2e.GLV_MUL_COLS, ld), dtype=torch.int64, device="cuda")
    err = torch.empty(n, dtype=torch.uint8, device="cuda"); valid = torch.empty(n, dtype=torch.uint8, device="cuda")
    for mode in ("prealloc", "default", "events"):
        ts = []
        for it in range(8):
            torch.cuda.synchronize(); t = time.perf_counter()
            if mode == "prealloc":
                ctx.glv_mul_witness_batch(px, py, k, cols=cols[:, :n], err=err, valid=valid, ld=ld)
            elif mode == "default":
                ctx.glv_mul_witness_batch(px, py, k, cols=cols[:, :n], ld=ld)
            else:
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); ctx.glv_mul_witness_batch(px, py, k, cols=cols[:, :n], err=err, valid=valid, ld=ld); e.record()
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
        print(n, mode, [round(x, 2) for x in ts], "phases", {k: round(v, 2) for k, v in ctx.last_phase_ms().items() if k in ("scalar", "total")})
