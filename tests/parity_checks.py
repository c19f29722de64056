"""Backend-independent parity checks against the committed golden fixtures (tests/golden/, generated
by oracle/gen_golden.py from the Python big-int restatement).  Bit-exact: integer work."""
import gzip
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KATS = json.load(open(os.path.join(GOLD, "prim_kats.json")))


def cols_of(list_of_limb_lists):
    """[[l0..lk], ...] -> (k, n) uint64"""
    return np.array(list_of_limb_lists, dtype=np.uint64).T.copy()


def _check(name, got, want, errs, want_errs):
    assert np.array_equal(errs != 0, want_errs != 0), f"{name}: error flags differ: {errs} vs {want_errs}"
    assert np.array_equal(errs[want_errs != 0] & want_errs[want_errs != 0], want_errs[want_errs != 0]), name
    ok = want_errs == 0
    for g, w, label in zip(got, want, range(len(got))):
        g = np.asarray(g)
        assert np.array_equal(g[..., ok], w[..., ok]), f"{name}: output {label} differs"


def check_add_sub(be):
    for kind in ("add", "sub"):
        for field in (0, 1):
            ks = [k for k in KATS["add_sub"] if k["kind"] == kind and k["field"] == field]
            a, b = cols_of([k["a"] for k in ks]), cols_of([k["b"] for k in ks])
            want_err = np.array([k["err"] for k in ks], dtype=np.uint8)
            want_out = cols_of([k["out"] or [0] * 9 for k in ks])
            want_ov = np.array([k["ov"] or 0 for k in ks], dtype=np.uint64)
            out, ov, err = getattr(be, kind)(field, a, b)
            _check(f"{kind}/{field}", (out, ov), (want_out, want_ov), np.asarray(err), want_err)


def check_mul(be):
    for field in (0, 1):
        ks = [k for k in KATS["mul"] if k["field"] == field]
        x, y = cols_of([k["x"] for k in ks]), cols_of([k["y"] for k in ks])
        want_err = np.array([k["err"] for k in ks], dtype=np.uint8)
        want = [cols_of([k[f] or [0] * n for k in ks]) for f, n in (("r", 9), ("q", 9), ("cs", 17), ("b", 16))]
        r, q, cs, b, err = be.mul(field, x, y)
        _check(f"mul/{field}", (r, q, cs, b), want, np.asarray(err), want_err)


def check_inv(be):
    for field in (0, 1):
        ks = [k for k in KATS["inv"] if k["field"] == field]
        x = cols_of([k["x"] for k in ks])
        want_err = np.array([k["err"] for k in ks], dtype=np.uint8)
        want = [cols_of([k[f] or [0] * 9 for k in ks]) for f in ("inv", "div")]
        inv, div, err = be.inv(field, x)
        _check(f"inv/{field}", (inv, div), want, np.asarray(err), want_err)


def check_div_rem(be):
    """BigUintDivRemGenerator KATs, one batch per (na, nb) shape."""
    shapes = sorted({(k["na"], k["nb"]) for k in KATS["div_rem"]})
    for na, nb in shapes:
        ks = [k for k in KATS["div_rem"] if (k["na"], k["nb"]) == (na, nb)]
        nd = 0 if nb > na + 1 else na - nb + 1
        a = np.array([k["a"] for k in ks], dtype=np.uint64).T.copy()
        b = np.array([k["b"] for k in ks], dtype=np.uint64).T.copy()
        want_err = np.array([k["err"] for k in ks], dtype=np.uint8)
        want_div = np.array([k["div"] if k["div"] is not None else [0] * nd for k in ks], dtype=np.uint64).reshape(len(ks), nd).T
        want_rem = np.array([k["rem"] if k["rem"] is not None else [0] * nb for k in ks], dtype=np.uint64).T
        div, rem, err = be.div_rem(a, b)
        _check(f"div_rem/{na}x{nb}", (div, rem), (want_div, want_rem), np.asarray(err), want_err)


def check_add_many(be):
    for field in (0, 1):
        for kk in (2, 4, 8):
            ks = [k for k in KATS["add_many"] if k["field"] == field and len(k["xs"]) == kk]
            s = np.array([[x for x in k["xs"]] for k in ks], dtype=np.uint64).transpose(1, 2, 0).copy()  # (k, 9, n)
            want_err = np.array([k["err"] for k in ks], dtype=np.uint8)
            want_out = cols_of([k["out"] or [0] * 9 for k in ks])
            want_ov = np.array([k["ov"] or 0 for k in ks], dtype=np.uint64)
            out, ov, err = be.add_many(field, s)
            _check(f"add_many/{field}/{kk}", (out, ov), (want_out, want_ov), np.asarray(err), want_err)


def check_glv(be):
    ks = KATS["glv"]
    k = cols_of([c["k"] for c in ks])
    want_err = np.array([c["err"] for c in ks], dtype=np.uint8)
    want = [cols_of([c["k1"] or [0] * 5 for c in ks]), cols_of([c["k2"] or [0] * 5 for c in ks]),
            np.array([c["n1"] or 0 for c in ks], dtype=np.uint64), np.array([c["n2"] or 0 for c in ks], dtype=np.uint64)]
    k1, k2, n1, n2, err = be.glv(k)
    _check("glv", (k1, k2, n1, n2), want, np.asarray(err), want_err)


def check_checksum(be):
    ks = KATS["checksum"]
    a = cols_of([c["a"] for c in ks])
    want_err = np.array([c["err"] for c in ks], dtype=np.uint8)
    want = cols_of([c["b"] or [0] * 16 for c in ks])
    b, err = be.checksum(a)
    _check("checksum", (b,), (want,), np.asarray(err), want_err)


def check_split_pack(be):
    rng = np.random.default_rng(5)
    packed = rng.integers(0, 256, size=(1001, 32), dtype=np.uint8)  # odd n: exercises the tail lane
    packed[0] = 0
    packed[1] = 255
    limbs = np.asarray(be.split(packed))
    vals = [int.from_bytes(bytes(r), "little") for r in packed]
    want = np.array([[(v >> (29 * k)) & ((1 << 29) - 1) for v in vals] for k in range(9)], dtype=np.uint64)
    assert np.array_equal(limbs, want)
    back, err = be.pack(limbs)
    assert np.array_equal(np.asarray(back), packed) and not np.asarray(err).any()
    bad = limbs.copy()
    bad[2, 5] = 1 << 29          # limb out of range
    bad[8, 7] = (1 << 29) - 1    # value >= 2^256
    _, err = be.pack(bad)
    err = np.asarray(err)
    assert err[5] & 1 and err[7] & 2 and not err[[0, 1, 2, 3, 4, 6]].any()


def load_verify_golden():
    g = np.load(os.path.join(GOLD, "verify_golden.npz"))
    return g["cols"], g["inputs"], g["valid"]


def check_verify_golden(be):
    cols, inputs, valid = load_verify_golden()
    args = [np.ascontiguousarray(inputs[:, k, :]) for k in range(5)]
    got, err, flags = be.verify(*args)
    assert not np.asarray(err).any()
    assert np.array_equal(np.asarray(flags), valid)
    got = np.asarray(got).view(np.uint64)
    bad = np.argwhere(got != cols)
    assert bad.size == 0, f"{len(bad)} mismatching (col, sig) entries, first {bad[:5].tolist()}"


def check_glv_mul_golden(be):
    g = np.load(os.path.join(GOLD, "glv_mul_golden.npz"))
    args = [np.ascontiguousarray(g["inputs"][:, k, :]) for k in range(3)]
    got, err, flags = be.glv_mul(*args)
    assert not np.asarray(err).any() and np.asarray(flags).all()
    assert np.array_equal(np.asarray(got).view(np.uint64), g["cols"])


def check_aux_golden(be):
    """Built-in-generator columns of the golden inputs, both programs."""
    g = np.load(os.path.join(GOLD, "aux_golden.npz"))
    _cols, inputs, _valid = load_verify_golden()
    _c, aux, err = be.aux(0, [np.ascontiguousarray(inputs[:, k, :]) for k in range(5)])
    assert not np.asarray(err).any()
    bad = np.argwhere(np.asarray(aux) != g["verify"])
    assert bad.size == 0, f"{len(bad)} mismatching (aux col, sig) entries, first {bad[:5].tolist()}"
    gi = np.load(os.path.join(GOLD, "glv_mul_golden.npz"))["inputs"]
    _c, aux, err = be.aux(1, [np.ascontiguousarray(gi[:, k, :]) for k in range(3)])
    assert not np.asarray(err).any()
    assert np.array_equal(np.asarray(aux), g["glv_mul"])


def golden_aux_schedule(name):
    with gzip.open(os.path.join(GOLD, "aux_schedule.json.gz"), "rt") as f:
        return [tuple(x) for x in json.load(f)[name]]


def golden_schedule(name):
    with gzip.open(os.path.join(GOLD, f"schedule_{name}.json.gz"), "rt") as f:
        return [tuple(x) for x in json.load(f)]


ALL_PRIM_CHECKS = (check_add_sub, check_mul, check_inv, check_add_many, check_glv, check_checksum, check_split_pack,
                   check_div_rem)


def structured_values(seed, count, bound_bits=256):
    """Operands made of words with long runs of ones / zeros / alternating bits: the inputs on which
    carry chains and fold-style reductions actually take their rare paths."""
    rng = np.random.default_rng(seed)
    words = np.array([0, 1, 2, 3, 0xFFFFFFFF, 0xFFFFFFFE, 0x80000000, 0x7FFFFFFF, 0x55555555, 0xAAAAAAAA,
                      0xFFFFFC2F, 0xD0364141], dtype=np.uint64)
    pick = rng.integers(0, len(words), size=(count, 8))
    rnd = rng.integers(0, 1 << 32, size=(count, 8), dtype=np.uint64)
    use_rnd = rng.random((count, 8)) < 0.2
    w = np.where(use_rnd, rnd, words[pick])
    vals = []
    for row in w:
        v = 0
        for k in range(8):
            v |= int(row[k]) << (32 * k)
        vals.append(v & ((1 << bound_bits) - 1))
    return vals


def check_structured_mul_add_sub(be, ora, count=4000):
    import oracle_c
    for field in (0, 1):
        x = oracle_c.limbs_cols(structured_values(10 + field, count))
        y = oracle_c.limbs_cols(structured_values(20 + field, count))
        for g, w in zip(be.mul(field, x, y), ora.mul(field, x, y)):
            assert np.array_equal(np.asarray(g), w), f"mul field {field}"
        for name in ("add", "sub"):
            for g, w in zip(getattr(be, name)(field, x, y), getattr(ora, name)(field, x, y)):
                assert np.array_equal(np.asarray(g), w), f"{name} field {field}"
        k = x[:, : max(64, count // 16)]
        got, want = be.inv(field, k), ora.inv(field, k)
        assert np.array_equal(np.asarray(got[2]) != 0, want[2] != 0)
        ok = want[2] == 0
        for g, w in zip(got[:2], want[:2]):
            assert np.array_equal(np.asarray(g)[:, ok], w[:, ok]), f"inv field {field}"
    kk = oracle_c.limbs_cols(structured_values(30, count))
    for g, w in zip(be.glv(kk), ora.glv(kk)):
        assert np.array_equal(np.asarray(g), w), "glv"
