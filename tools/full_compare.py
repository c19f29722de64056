"""One-off deep check: a 2^16 batch through the production path (run expansion, paired stores), the first COUNT signatures
compared with the oracle on EVERY column, for the u64 matrix, the compact container and the built-in-generator columns."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import plonky2_ecdsa_amd as p2e, oracle_c
count = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
n = 1 << 16
sigs = p2e.synth_signatures(seed=20260, n=n)
ctx = p2e.Context(device=0)
dev = [torch.from_numpy(a).cuda() for a in sigs]
cols, err, valid, bad = ctx.ecdsa_verify_witness_batch(*dev)
aux, aerr, abad = ctx.aux_witness_batch(0, dev[4], cols, n=n, ld=cols.stride(0))
nar, wid, cerr, cvalid, cbad = ctx.ecdsa_verify_witness_compact_batch(*dev)
torch.cuda.synchronize()
t = time.time()
want, want_aux, werr, wflags = oracle_c.verify_witness_aux(*[a[:count] for a in sigs])
dt = time.time() - t
got = cols[:, :count].cpu().numpy().view(np.uint64)
got_aux = aux[:, :count].cpu().numpy().view(np.uint64)
got_c = p2e.compact_expand(0, nar[:, :count].cpu().numpy().view(np.uint32), wid[:, :count].cpu().numpy())
res = {"signatures_compared": count, "columns": int(want.shape[0]), "aux_columns": int(want_aux.shape[0]),
       "u64_matrix_equal": bool(np.array_equal(got, want)), "aux_equal": bool(np.array_equal(got_aux, want_aux)),
       "compact_container_equal": bool(np.array_equal(got_c, want)), "flagged": int(bad + abad + cbad),
       "all_valid": bool(int(valid.sum()) == n and int(cvalid.sum()) == n and wflags.all()),
       "oracle_seconds": round(dt, 1), "elements_compared": int(count * (2 * want.shape[0] + want_aux.shape[0]))}
print(json.dumps(res))
assert res["u64_matrix_equal"] and res["aux_equal"] and res["compact_container_equal"] and res["all_valid"] and res["flagged"] == 0
