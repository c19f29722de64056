"""Generates the committed fixtures under tests/golden/ from the Python big-int restatement
(oracle/p2e_ref.py).  Run from the repo root:  python oracle/gen_golden.py

The reference (Rust) cannot be built or imported here, so these vectors are outputs of the
restatement, not of the reference ("parity unpinned" vs literal reference outputs; see p2e_ref.py).
They pin the C oracle, the CPU emulation of the kernels and the HIP kernels to ONE set of numbers.
"""
import gzip
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import p2e_ref as R  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)
M29 = (1 << 29) - 1


def limbs(v, n=9):
    return [(v >> (29 * i)) & M29 for i in range(n)]


def run(fn):
    try:
        return fn(), 0
    except R.RefPanic as e:
        return None, e.code


def kat_add_sub(field):
    m = R.MODULI[field]
    rng = R.SplitMix64(100 + field)
    cases = [(0, 0), (1, m - 1), (m - 1, 1), (m - 1, m - 1), (m - 2, 1), (0, m - 1), (5, 7), (7, 5),
             (m, 0), (m + 1, 3), (2**256 - 1, 2**256 - 1), (m // 2, m - m // 2), (m // 2 + 1, m - m // 2)]
    cases += [(rng.below(m), rng.below(m)) for _ in range(12)]
    out = []
    for a, b in cases:
        for kind, fn in (("add", R.gen_add), ("sub", R.gen_sub)):
            res, err = run(lambda: fn(limbs(a), limbs(b), m))
            out.append({"kind": kind, "field": field, "a": limbs(a), "b": limbs(b),
                        "out": res[0] if res else None, "ov": res[1] if res else None, "err": err})
    # limbs that are arbitrary Goldilocks elements: value >= 2^256 -> panic
    big = [M29] * 8 + [1 << 40]
    for kind, fn in (("add", R.gen_add), ("sub", R.gen_sub)):
        res, err = run(lambda: fn(big, limbs(1), m))
        out.append({"kind": kind, "field": field, "a": big, "b": limbs(1), "out": None, "ov": None, "err": err})
    # un-normalised limbs whose value still fits (get_biguint_target just sums limb << 29 i)
    odd = [(1 << 35) + 5] + [3] * 7 + [1]
    for kind, fn in (("add", R.gen_add), ("sub", R.gen_sub)):
        res, err = run(lambda: fn(odd, limbs(9), m))
        out.append({"kind": kind, "field": field, "a": odd, "b": limbs(9),
                    "out": res[0] if res else None, "ov": res[1] if res else None, "err": err})
    return out


def kat_mul(field):
    m = R.MODULI[field]
    rng = R.SplitMix64(200 + field)
    cases = [(0, 0), (1, 1), (1, m - 1), (m - 1, m - 1), (2**256 - 1, 2**256 - 1), (m, m), (m + 1, 2),
             (2**261 - 1, 1), (2**261 - 1, 2**261 - 1), (2**260, 2**257), (0, 2**256 - 1)]
    cases += [(rng.below(m), rng.below(m)) for _ in range(16)]
    # Rare branches of a fold-style reduction (second/third carry, final "r >= m" correction) fire only when
    # x*y mod m is tiny -- probability ~2^-125 on random operands: force them with y = r * x^-1 for small r,
    # canonical and non-canonical (y + m < 2^256) operands.  (p: x = 2^256 - 1 case constructed analytically.)
    for bits in (0, 1, 8, 32, 33, 64, 100, 128, 129, 130, 133):
        for rep in range(4):
            x = rng.below(m)
            r = rng.next() | (rng.next() << 64) | (rng.next() << 128)
            r &= (1 << bits) - 1
            y = r * pow(x, -1, m) % m
            if rep & 1 and y + m < 2**256:
                y += m
            cases.append((x, y))
    if field == 0:
        cases.append((2**256 - 1, 0xfffffc2c000e983fc85b8cd422f7173ab1f7703980a424c58e33ced0da7b7ff4))
    else:
        # third carry of the n-reduction (hi counts 8,5,1,1): search the small-remainder family until it fires
        c = 2**256 - m
        for _ in range(200000):
            x = rng.below(m)
            r = (rng.next() | (rng.next() << 64) | (rng.next() << 128)) & ((1 << 133) - 1)
            y = r * pow(x, -1, m) % m
            hi, lo = (x * y) >> 256, (x * y) & (2**256 - 1)
            path = []
            for _ in range(3):
                t = lo + hi * c
                hi, lo = t >> 256, t & (2**256 - 1)
                path.append(hi)
            if path[1] and path[2]:
                cases.append((x, y))
                break
        else:
            raise AssertionError("no third-carry case found")
    out = []
    for x, y in cases:
        def f():
            r, q, cs = R.gen_mul(limbs(x), limbs(y), m)
            b = R.gen_checksum(cs)
            assert R.check_mul_gate(limbs(x), limbs(y), r, q, cs, m) and R.check_checksum_gate(cs, b)
            return r, q, cs, b
        res, err = run(f)
        out.append({"field": field, "x": limbs(x), "y": limbs(y), "r": res[0] if res else None,
                    "q": res[1] if res else None, "cs": res[2] if res else None, "b": res[3] if res else None,
                    "err": err})
    bad = limbs(5)
    bad[3] = 1 << 29
    res, err = run(lambda: R.gen_mul(bad, limbs(1), m))
    out.append({"field": field, "x": bad, "y": limbs(1), "r": None, "q": None, "cs": None, "b": None, "err": err})
    return out


def kat_inv(field):
    m = R.MODULI[field]
    rng = R.SplitMix64(300 + field)
    cases = [1, 2, m - 1, m - 2, 0, m, m + 1, 2**256 - 1, (m + 1) // 2] + [rng.below(m) for _ in range(10)]
    out = []
    for x in cases:
        res, err = run(lambda: R.gen_inv(limbs(x), m))
        out.append({"field": field, "x": limbs(x), "inv": res[0] if res else None, "div": res[1] if res else None,
                    "err": err})
    return out


def kat_div_rem():
    """BigUintDivRemGenerator (gadgets/biguint.rs:508-518) at the shapes rem_biguint / reduce produce and a few odd ones.
    Inputs are plain 29-bit limb splits (NOT limbs_of: that has the set_biguint_target quirk on the input side)."""
    def split(v, k):
        return [(v >> (R.BITS * i)) & ((1 << R.BITS) - 1) for i in range(k)]
    rng = R.SplitMix64(700)

    def rnd(bits):
        v = 0
        for _ in range((bits + 63) // 64):
            v = (v << 64) | rng.next()
        return v >> (64 * ((bits + 63) // 64) - bits) if bits else 0
    out = []
    for na, nb in ((18, 9), (9, 9), (10, 9), (18, 1), (5, 9), (9, 5), (1, 1), (12, 7)):
        nd = 0 if nb > na + 1 else na - nb + 1
        cases = [(rnd(R.BITS * na), R.P if nb == 9 else max(1, rnd(R.BITS * nb))),          # reduce: b = field order
                 (rnd(R.BITS * na), R.N if nb == 9 else max(1, rnd(R.BITS * nb - 3))),
                 ((1 << (R.BITS * na)) - 1, 1), ((1 << (R.BITS * na)) - 1, (1 << (R.BITS * nb)) - 1),
                 (0, 5), (7, 0),                                                                # zero dividend, zero divisor
                 (rnd(min(R.BITS * na, R.BITS * nb)), max(1, rnd(R.BITS * nb)))]
        for bits_short in (0, 1, 2, 3, 4):     # quotients around the top of their limbs: the convert_base quirk zone
            b = max(1, rnd(R.BITS * nb - 5))
            qv = rnd(max(1, R.BITS * nd - bits_short)) | (1 << max(0, R.BITS * nd - bits_short - 1)) if nd else 0
            cases.append((min((1 << (R.BITS * na)) - 1, qv * b + (b >> 1)), b))
        for a, b in cases:
            al, bl = split(a, na), split(b & ((1 << (R.BITS * nb)) - 1), nb)
            res, err = run(lambda: R.gen_div_rem(al, bl))
            out.append({"na": na, "nb": nb, "a": al, "b": bl, "div": res[0] if res else None, "rem": res[1] if res else None,
                        "err": err})
    bad = split(12345, 9)
    bad[2] = 1 << 29                                                                            # a limb out of range
    res, err = run(lambda: R.gen_div_rem(bad, split(77, 9)))
    out.append({"na": 9, "nb": 9, "a": bad, "b": split(77, 9), "div": None, "rem": None, "err": err})
    return out


def kat_add_many(field):
    m = R.MODULI[field]
    rng = R.SplitMix64(400 + field)
    out = []
    for k in (1, 2, 4, 8):
        for trial in range(4):
            xs = [rng.below(m) for _ in range(k)]
            if trial == 0:
                xs = [m - 1] * k
            if trial == 1:
                xs = [0] * k
            if trial == 2 and k == 4:
                xs = [xs[0]] * 3 + [0]  # the curve_double pattern [xx, xx, xx, A=0]
            if k == 1:
                continue  # add_many_nonnative returns the operand itself (gadgets/nonnative.rs:315-317)
            res, err = run(lambda: R.gen_add_many([limbs(x) for x in xs], m))
            out.append({"field": field, "xs": [limbs(x) for x in xs], "out": res[0] if res else None,
                        "ov": res[1] if res else None, "err": err})
    return out


def kat_glv():
    rng = R.SplitMix64(500)
    n = R.N
    cases = [0, 1, 2, n - 1, n - 2, (n - 1) // 2, (n + 1) // 2, n, n + 5, 2**256 - 1, 2**128, 2**128 - 1,
             R.GLV_S, n - R.GLV_S] + [rng.below(n) for _ in range(24)]
    out = []
    for k in cases:
        res, err = run(lambda: R.gen_glv(limbs(k)))
        if res:
            k1, k2, n1, n2 = res
            kk = R.canon(k, n)
            v1 = R.value_of(k1) * (-1 if n1 else 1)
            v2 = R.value_of(k2) * (-1 if n2 else 1)
            assert (v1 + R.GLV_S * v2 - kk) % n == 0  # curve/glv.rs:114-125
        out.append({"k": limbs(k), "k1": res[0] if res else None, "k2": res[1] if res else None,
                    "n1": res[2] if res else None, "n2": res[3] if res else None, "err": err})
    return out


def kat_checksum():
    rng = R.SplitMix64(600)
    out = []
    for t in range(6):
        if t < 3:
            x, y = rng.below(R.P), rng.below(R.P)
            _, _, a = R.gen_mul(limbs(x), limbs(y), R.P)
        else:
            a = [rng.next() % R.P_GL for _ in range(17)]
        res, err = run(lambda: R.gen_checksum(a))
        out.append({"a": a, "b": res, "err": err})
    return out


def write_gz(path, text):
    """deterministic gzip (no timestamp) so regenerating the goldens does not dirty the tree"""
    with open(path, "wb") as raw:
        with gzip.GzipFile(filename="", mode="wb", fileobj=raw, mtime=0) as f:
            f.write(text.encode())


def main():
    kats = {"add_sub": kat_add_sub(0) + kat_add_sub(1), "mul": kat_mul(0) + kat_mul(1),
            "inv": kat_inv(0) + kat_inv(1), "add_many": kat_add_many(0) + kat_add_many(1), "glv": kat_glv(),
            "checksum": kat_checksum(), "div_rem": kat_div_rem()}
    with open(os.path.join(OUT, "prim_kats.json"), "w") as f:
        json.dump(kats, f, separators=(",", ":"))

    rp = R.rando_point()
    nr146 = rp
    for _ in range(146):
        nr146 = R.ec_double(nr146)
    consts = {
        "keccak256_of_8_zero_bytes": R.keccak256(bytes(8)).hex(),
        "hash_0_scalar": hex(int.from_bytes(R.keccak256(bytes(8)), "little")),
        "rando": [hex(rp[0]), hex(rp[1])],
        "neg_rando_146": [hex(nr146[0]), hex((-nr146[1]) % R.P)],
        "p_limbs": R.const_limbs(R.P), "n_limbs": R.const_limbs(R.N),
        "inv_2_29_goldilocks": hex(pow(1 << 29, -1, R.P_GL)),
        "fb_table_sha256": hashlib.sha256(b"".join(
            c.to_bytes(32, "little") for w in range(66) for pt in R.fixed_base_window(
                R.ec_mul(16**w, R.G)) for c in pt)).hexdigest(),
    }
    with open(os.path.join(OUT, "constants.json"), "w") as f:
        json.dump(consts, f, indent=1)

    # full-verify column dumps: one valid signature, one with a wrong message (constraints fail, witness
    # still fully defined), one with non-canonical inputs (pk.x + p, r + n if they fit 256 bits: no)
    sigs = [R.synth_signature_at(1, 0), R.synth_signature_at(1, 1)]
    bad = list(R.synth_signature_at(1, 2))
    bad[0] = (bad[0] + 1) % R.N
    sigs.append(tuple(bad))
    cols = []
    oks = []
    ops = None
    for s in sigs:
        c, ok, ops = R.verify_witness(*s)
        cols.append(c)
        oks.append(bool(ok))
    arr = np.array(cols, dtype=np.uint64).T.copy()  # (82615, 3)
    np.savez_compressed(os.path.join(OUT, "verify_golden.npz"), cols=arr,
                        inputs=np.frombuffer(b"".join(v.to_bytes(32, "little") for s in sigs for v in s),
                                             dtype=np.uint8).reshape(len(sigs), 5, 32),
                        valid=np.array(oks, dtype=np.uint8))
    write_gz(os.path.join(OUT, "schedule_verify.json.gz"), json.dumps([list(o) for o in ops], separators=(",", ":")))

    # glv_mul alone (BASELINE config 3)
    g_in = [(R.synth_signature_at(3, i)[3], R.synth_signature_at(3, i)[4], R.SplitMix64(33 + i).below(R.N))
            for i in range(2)]
    gcols = []
    gops = None
    for px, py, k in g_in:
        c, ok, gops = R.glv_mul_witness(px, py, k)
        assert ok
        gcols.append(c)
    np.savez_compressed(os.path.join(OUT, "glv_mul_golden.npz"), cols=np.array(gcols, dtype=np.uint64).T.copy(),
                        inputs=np.frombuffer(b"".join(v.to_bytes(32, "little") for s in g_in for v in s),
                                             dtype=np.uint8).reshape(len(g_in), 3, 32))
    write_gz(os.path.join(OUT, "schedule_glv_mul.json.gz"), json.dumps([list(o) for o in gops], separators=(",", ":")))

    # built-in-generator values of the same golden inputs, and their column map
    aux, aops = [], None
    for s in sigs:
        _c, a, _ok, aops = R.verify_witness_aux(*s)
        aux.append(a)
    gaux, gaops = [], None
    for px, py, k in g_in:
        _c, a, _ok, gaops = R.glv_mul_witness_aux(px, py, k)
        gaux.append(a)
    np.savez_compressed(os.path.join(OUT, "aux_golden.npz"), verify=np.array(aux, dtype=np.uint64).T.copy(),
                        glv_mul=np.array(gaux, dtype=np.uint64).T.copy())
    write_gz(os.path.join(OUT, "aux_schedule.json.gz"),
             json.dumps({"verify": [list(o) for o in aops], "glv_mul": [list(o) for o in gaops]}, separators=(",", ":")))
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
