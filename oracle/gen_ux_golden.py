"""Pins the constraint-block (U29 gate) model of oracle/check_circuit.py: SHA-256 of the 8(f) rank 2 vector
(249 385 values per verify, 194 361 per glv_mul, little-endian u32) of every golden signature, written to
tests/golden/ux_digest.json.  Run from the repo root:  python oracle/gen_ux_golden.py
(plonky2_ux is not available offline: these digests pin OUR model of its gates, "parity unpinned".)"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import check_circuit as CC  # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")


def digest(ux):
    return hashlib.sha256(np.asarray(ux, dtype="<u4").tobytes()).hexdigest()


def main():
    out = {"verify": [], "glv_mul": []}
    z = np.load(os.path.join(GOLD, "verify_golden.npz"))
    for i in range(z["cols"].shape[1]):
        if not z["valid"][i]:
            out["verify"].append(None)      # the invalid golden fails its last connect: no replay, no vector
            continue
        c = CC.check_verify(z["cols"][:, i], *CC.unpack_inputs([z["inputs"][:, k, :] for k in range(5)], i))
        out["verify"].append({"len": len(c.ux), "sha256": digest(c.ux), "head": [int(v) for v in c.ux[:8]]})
    g = np.load(os.path.join(GOLD, "glv_mul_golden.npz"))
    for i in range(g["cols"].shape[1]):
        c = CC.check_glv_mul(g["cols"][:, i], *CC.unpack_inputs([g["inputs"][:, k, :] for k in range(3)], i))
        out["glv_mul"].append({"len": len(c.ux), "sha256": digest(c.ux), "head": [int(v) for v in c.ux[:8]]})
    json.dump(out, open(os.path.join(GOLD, "ux_digest.json"), "w"), indent=1)
    print(json.dumps(out)[:300])


if __name__ == "__main__":
    main()
