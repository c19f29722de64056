"""CPU: the C oracle against the committed golden fixtures (made by the Python big-int restatement) and
the restatement against the reference's constraint equations.  -m "not gpu"."""
import json
import os

import numpy as np
import pytest

import oracle_c
import p2e_ref as R
import parity_checks as pc
from backends import OracleBackend


@pytest.mark.parametrize("check", pc.ALL_PRIM_CHECKS, ids=lambda f: f.__name__)
def test_oracle_generators_match_golden(check):
    check(OracleBackend())


def test_oracle_verify_matches_golden():
    pc.check_verify_golden(OracleBackend())


def test_oracle_glv_mul_matches_golden():
    pc.check_glv_mul_golden(OracleBackend())


def test_oracle_aux_matches_golden():
    pc.check_aux_golden(OracleBackend())


def test_aux_values_satisfy_their_defining_relations():
    """The built-in generators' targets are tied to the hot-path columns by plain Goldilocks constraints
    (mul by bool, not = 1 - b, digit = sum of bits): re-evaluate them on the Python restatement's output."""
    sig = R.synth_signature_at(21, 3)
    cols, aux, ok, ops = R.verify_witness_aux(*sig)
    assert ok and len(aux) == R.NUM_VERIFY_AUX == 8959
    kinds = [o[0] for o in ops]
    assert kinds.count("random_access") == 66 + 73 and kinds.count("mul_by_bool") == 4 * 139 + 2 * 4
    # 4-bit split of u1: bits recompose the limbs, digits recompose the scalar
    bits = aux[0:261]
    u1_limbs = [sum(b << j for j, b in enumerate(bits[29 * k:29 * k + 29])) for k in range(9)]
    comb = aux[261:261 + 198]
    digits = comb[2::3]
    assert all(c == a + 4 * b for a, b, c in zip(comb[0::3], comb[1::3], digits))
    u1 = sum(l << (29 * k) for k, l in enumerate(u1_limbs))
    assert u1 == sum(d << (4 * t) for t, d in enumerate(digits)) and u1 < R.N
    assert u1 == sig[0] * pow(sig[2], -1, R.N) % R.N                                  # gadgets/ecdsa.rs:40-41
    # every conditional add: not_b = 1 - b, products are limb * bool, and exactly one side is non-zero
    for i, (kind, first, n, _label) in enumerate(ops):
        if kind == "not" and i + 4 < len(ops) and [o[0] for o in ops[i + 1:i + 5]] == ["mul_by_bool"] * 4:
            not_b = aux[first]
            xt, yt, xf, yf = [aux[o[1]:o[1] + o[2]] for o in ops[i + 1:i + 5]]
            assert not_b in (0, 1)
            assert (not any(xf) and not any(yf)) if not_b == 0 else (not any(xt) and not any(yt))


def test_constants():
    c = json.load(open(os.path.join(pc.GOLD, "constants.json")))
    # Keccak digest quoted in SURVEY.md 8c and hard-coded in oracle/p2e_oracle.c + csrc/consts.hpp
    assert R.keccak256(bytes(8)).hex() == "011b4d03dd8c01f1049143cf9c4c817e4b167f1d1b83e5c6f0f10d89ba1e7bce"
    assert c["keccak256_of_8_zero_bytes"] == R.keccak256(bytes(8)).hex()
    assert R.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"  # known answer
    rx, ry = oracle_c.rando()
    assert [hex(rx), hex(ry)] == c["rando"]
    assert (ry * ry - rx**3 - 7) % R.P == 0
    assert pow(R.GLV_BETA, 3, R.P) == 1 and pow(R.GLV_S, 3, R.N) == 1           # curve/glv.rs constants
    assert (R.GY**2 - R.GX**3 - 7) % R.P == 0                                    # curve/secp256k1.rs:49-60
    # lambda * G == (beta * Gx, Gy): the endomorphism the GLV gadget relies on (curve/glv.rs:84-102)
    assert R.ec_mul(R.GLV_S, R.G) == (R.GLV_BETA * R.GX % R.P, R.GY)


def test_reference_to_digits_kat():
    """The one literal KAT in the reference (curve/curve_msm.rs:199-233, native to_digits w=17): restated
    here to pin the LE bit-order convention the window-digit split shares."""
    x_canonical = [0b10101010101010101010101010101010, 0b10101010101010101010101010101010,
                   0b11001100110011001100110011001100, 0b11001100110011001100110011001100,
                   0b11110000111100001111000011110000, 0b11110000111100001111000011110000,
                   0b00001111111111111111111111111111, 0b11111111111111111111111111111111]
    x = sum(w << (32 * i) for i, w in enumerate(x_canonical))
    digits = [(x >> (17 * i)) & ((1 << 17) - 1) for i in range(16)]
    assert digits == [0b01010101010101010, 0b10101010101010101, 0b01010101010101010, 0b11001010101010101,
                      0b01100110011001100, 0b00110011001100110, 0b10011001100110011, 0b11110000110011001,
                      0b01111000011110000, 0b00111100001111000, 0b00011110000111100, 0b11111111111111110,
                      0b01111111111111111, 0b11111111111111000, 0b11111111111111111, 0b1]
    # and the gadget's split (per-29-bit-limb LE bits, regrouped) is the same plain base-2^w expansion
    limbs = R.limbs_of(x % R.N, 9)
    v = R.value_of(limbs)
    assert R.Walker().split_4(limbs) == [(v >> (4 * i)) & 15 for i in range(66)]
    assert R.Walker().split_2(limbs[:5]) == [(R.value_of(limbs[:5]) >> (2 * i)) & 3 for i in range(73)]


def test_python_ref_vs_c_oracle_random_signatures():
    sigs = [R.synth_signature_at(11, i) for i in range(3)]
    arrs = [oracle_c.pack256([s[k] for s in sigs]) for k in range(5)]
    cols, err, flags = oracle_c.verify_witness(*arrs)
    assert not err.any() and flags.all()
    for i, s in enumerate(sigs):
        ref, ok, _ = R.verify_witness(*s)
        assert ok and np.array_equal(np.array(ref, dtype=np.uint64), cols[:, i])


def test_self_contained_ranges_on_every_generator_of_a_golden_witness():
    """Cheap, operand-free sanity on every generator's own columns (ranges, the CheckSumGate carry chain).  The
    reference's constraint equations WITH their operands (add/sub/inv/add_many relations, MulNonnativeGate, GLV,
    the three connects) are replayed by oracle/check_circuit.py: tests/test_check_circuit.py."""
    cols, inputs, valid = pc.load_verify_golden()
    ops = pc.golden_schedule("verify")
    col = cols[:, 0]
    for kind, field, c0, nc, label in ops:
        m = R.MODULI[field]
        seg = [int(v) for v in col[c0:c0 + nc]]
        if kind == "mul":
            r, q, cs, b = seg[:9], seg[9:18], seg[18:35], seg[35:51]
            assert R.check_checksum_gate(cs, b)
            assert R.value_of(r) < m and all(l < 1 << 29 for l in r + q)
        elif kind == "inv":
            inv, div = R.value_of(seg[:9]), R.value_of(seg[9:18])
            assert 0 < inv < m and div < m
        elif kind in ("add", "sub"):
            assert seg[9] in (0, 1) and R.value_of(seg[:9]) <= m
        elif kind == "add_many":
            assert seg[9] < 4 and R.value_of(seg[:9]) < m


def test_error_semantics():
    be = OracleBackend()
    x = oracle_c.limbs_cols([0, R.P, 5])
    inv, div, err = be.inv(0, x)
    assert list(err) == [R.ERR_INVERSE_OF_ZERO, R.ERR_INVERSE_OF_ZERO, 0]


def test_optimised_cpu_variant_is_bit_identical():
    """bench.py's cpu_baseline_optimised (lock-step groups, Montgomery batch inversion across signatures) writes the
    same columns as the faithful-cost walk, also when an element drops out of lock step (s = 0: its inverse generator
    flags instead of asking for an inversion) and for a group size that does not divide the batch."""
    import plonky2_ecdsa_amd as p2e
    arrs = [a.copy() for a in p2e.synth_signatures(seed=12, n=41)]
    arrs[2][5] = 0            # s = 0
    arrs[1][7, 0] ^= 1        # does not verify
    a = oracle_c.verify_witness(*arrs)
    for group in (1, 16, 64):
        b = oracle_c.verify_witness_lockstep(*arrs, nthreads=3, group=group)
        ok = a[1] == 0
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[0][:, ok], b[0][:, ok])
    assert a[1][5] == R.ERR_INVERSE_OF_ZERO and a[2][7] == 0
