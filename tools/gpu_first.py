"""First GPU contact: smoke (parity vs oracle) + phase timings at a few batch sizes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as ge
import plonky2_ecdsa_amd as p2e

ge.smoke()
ctx = p2e.Context(device=0)
for n in [int(x) for x in (sys.argv[1:] or ["4096", "16384", "65536"])]:
    sigs = p2e.synth_signatures(seed=4, n=n)
    dev = [torch.from_numpy(a).cuda() for a in sigs]
    cols = torch.empty((p2e.VERIFY_COLS, n), dtype=torch.int64, device="cuda")
    err = torch.empty(n, dtype=torch.uint8, device="cuda")
    valid = torch.empty(n, dtype=torch.uint8, device="cuda")
    for it in range(3):
        torch.cuda.synchronize()
        t = time.time()
        _, _, _, bad = ctx.ecdsa_verify_witness_batch(*dev, cols=cols, err=err, valid=valid)
        torch.cuda.synchronize()
        dt = time.time() - t
        ph = ctx.last_phase_ms()
        print(f"n={n} it={it} bad={bad} valid={int(valid.sum())} wall={dt*1e3:.2f} ms fills/s={n/dt:.0f} "
              f"GB/s={n*661080/dt/1e9:.1f} phases={ph}", flush=True)
    del cols
    torch.cuda.empty_cache()
