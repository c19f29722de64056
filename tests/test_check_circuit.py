"""CPU: the full constraint replay (oracle/check_circuit.py) on the committed goldens and on the C oracle's
output, and the mutation test that shows the replay has teeth: a witness with ANY single corrupted column of any
generator kind is rejected.  This is the strongest pin of the (parity-unpinned) oracle this environment allows:
every generator output is pushed, with its operands, through the reference's own constraint equations
(gadgets/nonnative.rs:262-273,330-351,373-386,518-530; gates/mul_nonnative.rs:101-130,411-427; range checks
nonnative.rs:453-460; gadgets/glv.rs:76-82; connects curve.rs:134, glv.rs:82, ecdsa.rs:48-52)."""
import numpy as np
import pytest

import check_circuit as CC
import oracle_c
import p2e_ref as R
import parity_checks as pc


def _golden(i):
    cols, inputs, valid = pc.load_verify_golden()
    aux = np.load(pc.GOLD + "/aux_golden.npz")["verify"]
    return cols[:, i], CC.unpack_inputs([inputs[:, k, :] for k in range(5)], i), aux[:, i], int(valid[i])


def test_replay_accepts_valid_goldens_and_counts_every_generator():
    for i in (0, 1):
        cols, ins, aux, valid = _golden(i)
        assert valid == 1
        c = CC.check_verify(cols, *ins, aux=aux)
        assert len(c.gens) == 3555 and c.cur == 82615 and len(c.aux) == 8959
        assert c.counts == {"mul": 1087, "inv": 312, "sub": 1267, "add": 742, "add_many": 146, "glv": 1}
        # the replay's own registration order == the golden schedule (an independent second walk of the gadgets)
        sched = pc.golden_schedule("verify")
        assert [(g[0], g[1], g[2], g[3], g[4]) for g in c.gens] == [tuple(s) for s in sched]


def test_replay_rejects_the_invalid_golden_only_at_the_final_connect():
    cols, ins, aux, valid = _golden(2)
    assert valid == 0
    with pytest.raises(CC.ConstraintViolation) as e:
        CC.check_verify(cols, *ins, aux=aux)
    assert "connect_nonnative(r, point.x)" in str(e.value)


def test_replay_accepts_glv_mul_goldens():
    g = np.load(pc.GOLD + "/glv_mul_golden.npz")
    aux = np.load(pc.GOLD + "/aux_golden.npz")["glv_mul"]
    for i in range(g["cols"].shape[1]):
        ins = CC.unpack_inputs([g["inputs"][:, k, :] for k in range(3)], i)
        c = CC.check_glv_mul(g["cols"][:, i], *ins, aux=aux[:, i])
        assert c.cur == 65243 and len(c.aux) == 4738


def test_replay_accepts_c_oracle_output_on_fresh_signatures():
    """columns that did NOT come from the Python restatement: the C oracle on fresh synthetic signatures"""
    sigs = [R.synth_signature_at(77, i) for i in range(2)]
    arrs = [oracle_c.pack256([s[k] for s in sigs]) for k in range(5)]
    cols, aux, err, flags = oracle_c.verify_witness_aux(*arrs)
    assert not err.any() and flags.all()
    for i, s in enumerate(sigs):
        CC.check_verify(cols[:, i], *s, aux=aux[:, i])


# every output class of every generator kind: (kind, offset inside the generator's columns)
_MUTATIONS = [("add", 0), ("add", 8), ("add", 9), ("sub", 0), ("sub", 8), ("sub", 9), ("add_many", 3), ("add_many", 9),
              ("mul", 0), ("mul", 8), ("mul", 9), ("mul", 17), ("mul", 18), ("mul", 34), ("mul", 35), ("mul", 50),
              ("inv", 0), ("inv", 8), ("inv", 9), ("inv", 17), ("glv", 0), ("glv", 4), ("glv", 5), ("glv", 9), ("glv", 10),
              ("glv", 11)]


def test_any_single_corrupted_column_fails_the_replay():
    cols, ins, aux, _ = _golden(0)
    sched = pc.golden_schedule("verify")
    rng = np.random.default_rng(1)
    tried = 0
    for kind, off in _MUTATIONS:
        gens = [g for g in sched if g[0] == kind]
        # one early instance (cheap: the replay stops at the first violation) and one anywhere in the circuit
        picks = [gens[int(rng.integers(0, min(len(gens), 12)))], gens[int(rng.integers(0, len(gens)))]]
        for g in picks[:1 if kind == "glv" else 2]:
            bad = cols.copy()
            c = g[2] + off
            # +1, or a bit flip for the boolean columns (still a well-formed Goldilocks element either way)
            bad[c] = (int(bad[c]) ^ 1) if int(bad[c]) in (0, 1) else (int(bad[c]) + 1) % R.P_GL
            with pytest.raises(CC.ConstraintViolation):
                CC.check_verify(bad, *ins)
            tried += 1
    assert tried == 2 * 20 + 6
    # the unused results of the window table (cur_p / cur_q of i = 3, gadgets/curve_msm.rs:45-50) are pinned too
    unused = [g for g in sched if g[4] == "glv_mul/msm/table"][70:80]
    bad = cols.copy()
    c = unused[-1][2] + 3                      # y3 = sub(pr, y1) of the 8th add: consumed by no later gadget
    bad[c] = (int(bad[c]) + 1) % R.P_GL
    with pytest.raises(CC.ConstraintViolation):
        CC.check_verify(bad, *ins)


def test_corrupted_built_in_generator_column_fails_the_replay():
    cols, ins, aux, _ = _golden(0)
    rng = np.random.default_rng(2)
    for c in [0, 260, 300, 470, 8958] + [int(x) for x in rng.integers(0, 8959, size=5)]:
        bad = aux.copy()
        bad[c] = int(bad[c]) ^ 1
        with pytest.raises(CC.ConstraintViolation):
            CC.check_verify(cols, *ins, aux=bad)


def test_wrong_inputs_fail_the_replay():
    cols, ins, aux, _ = _golden(0)
    for k in range(5):
        wrong = list(ins)
        wrong[k] ^= 1 << 17
        with pytest.raises(CC.ConstraintViolation):
            CC.check_verify(cols, *wrong)


@pytest.mark.parametrize("program", [0, 1])
def test_library_wiring_equals_the_replayed_gadget_wiring(program):
    """include/p2e.h p2e_schedule_wiring (C++ ScheduleBuilder: where every generator's operands live) against the
    operand sources the constraint replay records while it walks the reference's gadgets: same source for every
    operand of all 3 555 generators, same limb counts (constants keep convert_base's count, quirk Q5), same
    range_check flags.  This is the table the 8(f) rank 2 / rank 3 kernels are driven by."""
    import plonky2_ecdsa_amd as p2e
    if program == 0:
        cols, ins, aux, _ = _golden(0)
        c = CC.check_verify(cols, *ins, aux=aux)
    else:
        g = np.load(pc.GOLD + "/glv_mul_golden.npz")
        c = CC.check_glv_mul(g["cols"][:, 0], *CC.unpack_inputs([g["inputs"][:, k, :] for k in range(3)], 0))
    wiring = p2e.schedule_wiring(program)
    consts = {cid: p2e.wiring_const(cid) for cid in range(10)}
    slot = {"pky": 0, "pkx": 1, "msg": 2, "k": 2, "r": 3, "s": 4}
    assert len(wiring) == len(c.gens)
    n_rc = 0
    for (kind, field, c0, nc, label, operands), (w_ops, w_rc) in zip(c.gens, wiring):
        assert len(operands) == len(w_ops), (label, kind)
        for limbs, (src, nl) in zip(operands, w_ops):
            limbs = list(limbs)
            assert len(limbs) == nl, (label, kind, limbs, nl)
            if nl == 0:
                assert src == p2e.SRC_CONST | 6                     # the zero constant: no limbs
                continue
            first = limbs[0]
            if first[0] == "col":
                assert src == first[1] and [s for s in limbs] == [("col", first[1] + k) for k in range(nl)]
            elif first[0] == "aux":
                assert src == p2e.SRC_AUX | first[1] and limbs == [("aux", first[1] + k) for k in range(nl)]
            elif first[0] == "in":
                assert src == p2e.SRC_INPUT | slot[first[1]] and limbs == [("in", first[1], k) for k in range(nl)]
            else:
                assert src & p2e.SRC_CONST and (src & 0xFFFF) in consts
                value, cnl = consts[src & 0xFFFF]
                assert cnl == nl and [v for _, v in limbs] == R.const_limbs(value)
        n_rc += w_rc
    # range_check flags: replay them from the reference's call sites through the U29 block sizes the replay recorded
    # (a range-checked gadget has exactly one more value, the cmp_biguint result, in its constraint block)
    base = {"add": 45, "sub": 47, "inv": 434, "mul": 0, "glv": None}
    for (kind, field, c0, nc, label, operands), (w_ops, w_rc), (u0, un, ulabel) in zip(
            [g for g in c.gens if g[0] != "glv"], [w for g, w in zip(c.gens, wiring) if g[0] != "glv"], c.ux_ops):
        if kind in base:
            # operands shorter than 9 limbs do not change the count: add_biguint runs over max(len) limbs
            assert un - base[kind] == int(w_rc), (label, kind, un, w_rc)
    assert n_rc > 0
