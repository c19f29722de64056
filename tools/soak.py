"""Soak run: the same batch through the same entry point ITERS times, the whole output matrix reduced to two checksums
on the GPU each time -- every repetition must reproduce the first.  Catches missing stream / event dependencies in the
launch plans (they would show as sporadic differences), which single parity runs can miss.
Usage: python tools/soak.py [ITERS]   (default 25)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import plonky2_ecdsa_amd as p2e

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 25
ctx = p2e.Context(device=0)


def digest(cols, n):
    v = cols[:, :n]
    return int(v.sum().item()), int((v ^ (v >> 7)).sum().item())


def soak(label, n, ncols, call):
    ld = n + 16
    cols = torch.zeros((ncols, ld), dtype=torch.int64, device="cuda")
    err = torch.empty(n, dtype=torch.uint8, device="cuda")
    valid = torch.empty(n, dtype=torch.uint8, device="cuda")
    first, t0 = None, time.time()
    for it in range(iters):
        cols.fill_(-1 if it & 1 else 0)                     # stale values from the previous repetition cannot pass
        bad = call(cols[:, :n], err, valid, ld)
        torch.cuda.synchronize()
        d = digest(cols, n) + (int(valid.sum().item()), bad)
        if first is None:
            first = d
        assert d == first, f"{label}: repetition {it} differs: {d} vs {first}"
    print(f"{label}: {iters} repetitions identical (valid {first[2]} of {n}, flagged {first[3]}), {time.time() - t0:.1f} s", flush=True)
    del cols


for n in (8192, 33000, 65536):
    sig = [torch.from_numpy(a).cuda() for a in p2e.synth_signatures(seed=4, n=n)]
    soak(f"verify_secp256k1 n={n}", n, p2e.VERIFY_COLS,
         lambda c, e, v, ld: ctx.ecdsa_verify_witness_batch(*sig, cols=c, err=e, valid=v, ld=ld)[3])
    if n != 65536:
        soak(f"glv_mul n={n}", n, p2e.GLV_MUL_COLS,
             lambda c, e, v, ld: ctx.glv_mul_witness_batch(sig[3], sig[4], sig[0], cols=c, err=e, valid=v, ld=ld)[3])
for curve, cname in ((p2e.CURVE_SECP256K1, "secp256k1"), (p2e.CURVE_P256, "p256")):
    b = p2e.synth_signatures_curve(curve, seed=777, n=1)
    blind = (int.from_bytes(bytes(b[3][0]), "little"), int.from_bytes(bytes(b[4][0]), "little"))
    base = p2e.synth_signatures_curve(curve, seed=5, n=2048)
    for kind, kname, sizes in ((p2e.CP_WINDOWED_MUL, "windowed", (8192, 65536)), (p2e.CP_SCALAR_MUL, "scalar_mul", (16384,)),
                               (p2e.CP_VERIFY, "verify", (8192, 65536))):
        if kind == p2e.CP_VERIFY and curve != p2e.CURVE_P256:
            continue
        prog = p2e.CurveProgram(ctx, kind, curve, blind)
        for n in sizes:
            import numpy as np
            sig = [torch.from_numpy(np.tile(a, (n // 2048, 1))).cuda() for a in base]
            if kind == p2e.CP_VERIFY:
                call = lambda c, e, v, ld: prog.verify_witness_batch(*sig, cols=c, err=e, valid=v, ld=ld)[3]
            else:
                call = lambda c, e, v, ld: prog.mul_witness_batch(sig[3], sig[4], sig[0], cols=c, err=e, valid=v, ld=ld)[3]
            soak(f"{kname}_{cname} n={n}", n, prog.num_cols, call)
        prog.close()
print("soak ok")
