import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import plonky2_ecdsa_amd as p2e
import oracle_c
n = int(os.environ.get("DBG_N", "8"))
sigs = p2e.synth_signatures(seed=5, n=n)
want, werr, wf = oracle_c.verify_witness(*sigs)
ctx = p2e.Context(device=0, host_pointers=True)
got, err, valid, bad = ctx.ecdsa_verify_witness_batch(*sigs)
got = np.asarray(got).view(np.uint64)
mm = np.argwhere(got != want)
print("env", {k: v for k, v in os.environ.items() if k.startswith("P2E_")}, "err", err[:16], "valid", valid[:16], "mismatches", len(mm),
      "first", mm[:3].tolist(), "cols with mismatch (first 5 distinct)", sorted(set(mm[:, 0].tolist()))[:5])
