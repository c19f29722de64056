import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import plonky2_ecdsa_amd as p2e, oracle_c
for n in (4096, 1 << 15, 49152, 1 << 16):
    sigs = p2e.synth_signatures(seed=4, n=n)
    ctx = p2e.Context(device=0)
    dev = [torch.from_numpy(a).cuda() for a in sigs]
    cols, err, valid, bad = ctx.ecdsa_verify_witness_batch(*dev)
    torch.cuda.synchronize()
    sample = np.linspace(0, n - 1, 48).astype(np.int64)
    want, want_aux, _, _ = oracle_c.verify_witness_aux(*[a[sample] for a in sigs])
    got = cols[:, torch.from_numpy(sample).cuda()].cpu().numpy().view(np.uint64)
    print(n, "main equal", np.array_equal(got, want), "ld", cols.stride(0))
    for pad in (0, 16):
        aux_full = torch.zeros((p2e.VERIFY_AUX_COLS, n + pad), dtype=torch.int64, device="cuda")
        aux, aerr, abad = ctx.aux_witness_batch(0, dev[4], cols, n=n, ld=cols.stride(0), aux=aux_full[:, :n], ld_aux=n + pad)
        torch.cuda.synchronize()
        ga = aux_full[:, torch.from_numpy(sample).cuda()].cpu().numpy().view(np.uint64)
        d = np.argwhere(ga != want_aux)
        print("  pad", pad, "aux mismatches", len(d), d[:6].tolist(), "rows", sorted(set(d[:, 0].tolist()))[:10], "sigs", sorted(set(d[:,1].tolist()))[:10])
        if len(d):
            r, c = d[0]; print("   got", ga[r, c], "want", want_aux[r, c])
