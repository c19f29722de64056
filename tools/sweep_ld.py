"""Does the power-of-two column stride (ld = 2^16 elements = 512 KiB) camp on HBM channels?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import plonky2_ecdsa_amd as p2e
n = 1 << 16
sigs = p2e.synth_signatures(seed=4, n=n)
ctx = p2e.Context(device=0, phase_timing=True)
dev = [torch.from_numpy(a).cuda() for a in sigs]
err = torch.empty(n, dtype=torch.uint8, device="cuda"); valid = torch.empty(n, dtype=torch.uint8, device="cuda")
for pad in [0, 16, 32, 64, 96, 128, 256, 272, 1040, 4112]:
    ld = n + pad
    big = torch.empty((p2e.VERIFY_COLS, ld), dtype=torch.int64, device="cuda")
    view = big[:, :n]
    def step():
        return ctx._check(ctx._L.p2e_ecdsa_verify_witness_batch(ctx._h, *[p2e._ptr(d) for d in dev], p2e.C.c_void_p(view.data_ptr()),
                          p2e.C.c_size_t(n), p2e.C.c_size_t(ld), p2e._ptr(err), p2e._ptr(valid)))
    step(); step(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
    ph = ctx.last_phase_ms()
    print(f"pad={pad:5d} ms/step={dt*1e3:.3f} fills/s={n/dt:.0f} expand_ms={ph['expand']:.3f} expand_TBps={ph['expand_cols']*8*n/ph['expand']/1e9:.3f}", flush=True)
    del big, view
    torch.cuda.empty_cache()
