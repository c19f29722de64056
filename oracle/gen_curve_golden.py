"""Golden vectors of the curve programs, written by the big-int gadget walk of p2e_ref.py:
tests/golden/curve_programs.npz (inputs, blinding points, one full column vector per program) and
tests/golden/curve_programs.json (SHA-256 of every signature's column / built-in-generator vector, generator counts).
Run from the repo root:  python oracle/gen_curve_golden.py
("parity unpinned" as for every other golden of this repo: the reference holds no vectors and cannot be built here;
the constraint replay of oracle/check_circuit.py is what ties these numbers to the reference's gadgets.)"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import p2e_ref as R  # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")
PROGRAMS = [("windowed_secp256k1", 1, 0), ("scalar_mul_secp256k1", 2, 0), ("windowed_p256", 1, 1), ("scalar_mul_p256", 2, 1),
            ("verify_p256", 3, 1)]
CURVES = [R.SECP256K1, R.P256]
N_CASES = 4


def b32(v):
    return np.frombuffer(int(v).to_bytes(32, "little"), np.uint8).copy()


def digest(vals):
    return hashlib.sha256(np.asarray(vals, dtype="<u8").tobytes()).hexdigest()


def cases(name, kind, curve):
    """(blind, [inputs...]) of program `name`: deterministic from the name"""
    cv = CURVES[curve]
    rng = R.SplitMix64(int.from_bytes(hashlib.sha256(name.encode()).digest()[:8], "little"))
    blind = cv.mul(rng.below(cv.n), cv.g)
    out = []
    for i in range(N_CASES):
        if kind == 3:
            sig = list(R.synth_signature_curve(cv, rng))
            if i == N_CASES - 1:
                sig[0] ^= 1 << 77        # a signature that does not verify: every column still defined, valid = 0
            out.append(tuple(sig))
        else:
            pt = cv.mul(rng.below(cv.n), cv.g)
            k = rng.below(cv.n)
            if i == 1:
                k = (1 << 255) + (1 << 200) + 0xF0F0   # zero windows / long zero-bit runs; above 2^255
                k %= 1 << 256
            if i == 2:
                k = 0x1111111111111111                 # small scalar: the high windows never add
            out.append((pt[0], pt[1], k))
    return blind, out


def walk(kind, curve, blind, case):
    cv = CURVES[curve]
    if kind == 1:
        cols, aux, ops, _pt = R.windowed_mul_witness(cv, *case, blind)
        return cols, aux, ops, 1
    if kind == 2:
        cols, aux, ops, _pt = R.scalar_mul_witness(cv, *case, blind)
        return cols, aux, ops, 1
    cols, aux, ok, ops = R.verify_p256_witness(*case, blind)
    return cols, aux, ops, int(ok)


def main():
    arrays, meta = {}, {}
    for name, kind, curve in PROGRAMS:
        blind, cs = cases(name, kind, curve)
        arrays[name + "_blind"] = np.stack([b32(blind[0]), b32(blind[1])])
        arrays[name + "_inputs"] = np.stack([np.stack([b32(v) for v in c]) for c in cs])   # (cases, args, 32)
        m = {"kind": kind, "curve": curve, "cases": []}
        for i, c in enumerate(cs):
            cols, aux, ops, ok = walk(kind, curve, blind, c)
            if i == 0:
                arrays[name + "_cols0"] = np.asarray(cols, dtype=np.uint64)
                m["num_cols"], m["num_aux"], m["num_gens"] = len(cols), len(aux), len(ops)
                m["gens_sha256"] = hashlib.sha256(json.dumps([[o[0], o[1], o[2], o[3]] for o in ops]).encode()).hexdigest()
            m["cases"].append({"cols_sha256": digest(cols), "aux_sha256": digest(aux), "valid": ok})
        meta[name] = m
        print(name, m["num_cols"], m["num_aux"], m["num_gens"], [c["valid"] for c in m["cases"]])
    np.savez_compressed(os.path.join(GOLD, "curve_programs.npz"), **arrays)
    json.dump(meta, open(os.path.join(GOLD, "curve_programs.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
