"""Curve programs: curve_scalar_mul_windowed (gadgets/curve_windowed_mul.rs:131-173),
curve_scalar_mul (gadgets/curve.rs:245-285) on secp256k1 and P-256, verify_p256_message_circuit (gadgets/ecdsa.rs:55-78).

CPU tests: the big-int gadget walk against the committed goldens, the goldens through the constraint replay, the
kernel bodies (compiled for the CPU, tests/emu) against both, the schedule builder's operand wiring against the
replay's.  GPU tests: the HIP kernels through the C ABI against the goldens, the emulation and the big-int walk, and
their output through the constraint replay with no oracle value involved.

"parity unpinned": the reference holds no vectors for these gadgets and cannot be built here."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import check_circuit as CC
import p2e_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
META = json.load(open(os.path.join(GOLD, "curve_programs.json")))
PROGRAMS = list(META)
CURVES = [R.SECP256K1, R.P256]
SRC_AUX, SRC_INPUT, SRC_CONST = 0x20000000, 0x40000000, 0x80000000


def _npz():
    return np.load(os.path.join(GOLD, "curve_programs.npz"))


def _int(a):
    return int.from_bytes(bytes(bytearray(a)), "little")


def _case_ints(name):
    z = _npz()
    blind = (_int(z[name + "_blind"][0]), _int(z[name + "_blind"][1]))
    ins = z[name + "_inputs"]
    return blind, [[_int(ins[i, k]) for k in range(ins.shape[1])] for i in range(ins.shape[0])], z[name + "_cols0"]


def _digest(v):
    return hashlib.sha256(np.asarray(v, dtype="<u8").tobytes()).hexdigest()


def _walk(name, blind, case):
    kind, cv = META[name]["kind"], CURVES[META[name]["curve"]]
    if kind == 1:
        cols, aux, ops, _ = R.windowed_mul_witness(cv, *case, blind)
        return cols, aux, ops, 1
    if kind == 2:
        cols, aux, ops, _ = R.scalar_mul_witness(cv, *case, blind)
        return cols, aux, ops, 1
    cols, aux, ok, ops = R.verify_p256_witness(*case, blind)
    return cols, aux, ops, int(ok)


def _replay(name, cols, blind, case, aux=None):
    kind, cv = META[name]["kind"], CURVES[META[name]["curve"]]
    if kind == 1:
        return CC.check_windowed_mul(cv, cols, *case, blind, aux=aux)[0]
    if kind == 2:
        return CC.check_scalar_mul(cv, cols, *case, blind, aux=aux)[0]
    return CC.check_verify_p256(cols, *case, blind, aux=aux)


# ---- the kernel bodies compiled for the CPU (tests/emu) -----------------------------------------------------------
class Emu:
    def __init__(self):
        self.L = C.CDLL(os.path.join(ROOT, "tests", "emu", "libp2e_emu.so"))
        self.L.emu_curve_program.restype = C.c_long
        self.L.emu_curve_program_gens.restype = C.c_long

    @staticmethod
    def _p(a):
        return a.ctypes.data_as(C.c_void_p) if a is not None else None

    def sizes(self, kind, curve, blind):
        nc, ng, na = C.c_long(), C.c_long(), C.c_long()
        rc = self.L.emu_curve_program(kind, curve, self._p(blind[0]), self._p(blind[1]), None, None, None, None, None, None,
                                      C.c_size_t(0), C.c_size_t(0), None, None, 32, C.byref(nc), C.byref(ng), C.byref(na))
        assert rc == 0
        return nc.value, ng.value, na.value

    def run(self, kind, curve, blind, args, piece=32):
        """args: (px, py, k) or (msg, r, s, pkx, pky), each (n, 32) uint8"""
        if len(args) == 3:
            px, py, k = args
            msg = r = s = k
        else:
            msg, r, s, px, py = args
        n = px.shape[0]
        nc, _, _ = self.sizes(kind, curve, blind)
        cols, err, valid = np.zeros((nc, n), np.uint64), np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        bad = self.L.emu_curve_program(kind, curve, self._p(blind[0]), self._p(blind[1]), self._p(msg), self._p(r), self._p(s),
                                       self._p(px), self._p(py), self._p(cols), C.c_size_t(n), C.c_size_t(n), self._p(err),
                                       self._p(valid), piece, None, None, None)
        return cols, err, valid, bad

    def gens(self, kind, curve, blind):
        _, ng, _ = self.sizes(kind, curve, blind)
        kinds, fields = np.zeros(ng, np.int32), np.zeros(ng, np.int32)
        first, ncols = np.zeros(ng, np.uint32), np.zeros(ng, np.uint32)
        src, nl = np.zeros((ng, 4), np.uint32), np.zeros((ng, 4), np.uint8)
        assert self.L.emu_curve_program_gens(kind, curve, self._p(blind[0]), self._p(blind[1]), self._p(kinds), self._p(fields),
                                             self._p(first), self._p(ncols), self._p(src), self._p(nl), C.c_size_t(ng)) == ng
        return kinds, fields, first, ncols, src, nl

    def _args5(self, args):
        if len(args) == 3:
            px, py, k = args
            return k, None, None, px, py
        return tuple(args)

    def aux(self, kind, curve, blind, args, cols):
        """built-in-generator values from a finished witness matrix (the body kc_aux runs): (num_aux_cols, n), err"""
        b, P, z = blind, self._p, C.c_size_t(0)
        self.L.emu_curve_aux.restype = C.c_long
        na = self.L.emu_curve_aux(kind, curve, P(b[0]), P(b[1]), None, None, None, None, None, None, z, None, z, z, None)
        msg, r, s, px, py = self._args5(args)
        n = px.shape[0]
        cols = np.ascontiguousarray(cols)
        aux, err = np.zeros((na, n), np.uint64), np.zeros(n, np.uint8)
        self.L.emu_curve_aux(kind, curve, P(b[0]), P(b[1]), P(msg), P(r), P(s), P(px), P(py), P(cols), C.c_size_t(n), P(aux), C.c_size_t(n),
                             C.c_size_t(n), P(err))
        return aux, err

    def gate(self, kind, curve, blind, aux):
        b, P, z = blind, self._p, C.c_size_t(0)
        self.L.emu_curve_gate.restype = C.c_long
        ng = self.L.emu_curve_gate(kind, curve, P(b[0]), P(b[1]), None, z, None, z, z)
        n = aux.shape[1]
        gate = np.zeros((ng, n), np.uint64)
        if ng:
            self.L.emu_curve_gate(kind, curve, P(b[0]), P(b[1]), P(np.ascontiguousarray(aux)), C.c_size_t(n), P(gate), C.c_size_t(n), C.c_size_t(n))
        return gate

    def ux(self, kind, curve, blind, args, cols, aux):
        b, P, z = blind, self._p, C.c_size_t(0)
        self.L.emu_curve_ux.restype = C.c_long
        nu = self.L.emu_curve_ux(kind, curve, P(b[0]), P(b[1]), None, None, None, None, None, None, z, None, z, None, z, z, None)
        msg, r, s, px, py = self._args5(args)
        n = px.shape[0]
        ux, err = np.zeros((nu, n), np.uint64), np.zeros(n, np.uint8)
        self.L.emu_curve_ux(kind, curve, P(b[0]), P(b[1]), P(msg), P(r), P(s), P(px), P(py), P(np.ascontiguousarray(cols)), C.c_size_t(n),
                            P(np.ascontiguousarray(aux)), C.c_size_t(n), P(ux), C.c_size_t(n), C.c_size_t(n), P(err))
        return ux, err

    def const(self, kind, curve, blind, cid):
        out = np.zeros(32, np.uint8)
        assert self.L.emu_curve_program_const(kind, curve, self._p(blind[0]), self._p(blind[1]), C.c_uint32(cid), self._p(out)) == 0
        return _int(out)


@pytest.fixture(scope="module")
def emu():
    return Emu()


def _golden_arrays(name):
    z = _npz()
    blind = (np.ascontiguousarray(z[name + "_blind"][0]), np.ascontiguousarray(z[name + "_blind"][1]))
    ins = z[name + "_inputs"]
    return blind, [np.ascontiguousarray(ins[:, k, :]) for k in range(ins.shape[1])]


# ---- CPU ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", PROGRAMS)
def test_gadget_walk_reproduces_the_goldens(name):
    blind, cases, cols0 = _case_ints(name)
    m = META[name]
    for i, case in enumerate(cases):
        cols, aux, ops, ok = _walk(name, blind, case)
        assert (len(cols), len(aux), len(ops)) == (m["num_cols"], m["num_aux"], m["num_gens"])
        assert _digest(cols) == m["cases"][i]["cols_sha256"] and _digest(aux) == m["cases"][i]["aux_sha256"]
        assert ok == m["cases"][i]["valid"]
        if i == 0:
            assert np.array_equal(np.asarray(cols, np.uint64), cols0)


@pytest.mark.parametrize("name", PROGRAMS)
def test_c_oracle_reproduces_the_goldens_and_the_walk(name):
    """oracle/p2e_oracle.c's curve programs (fixed-width C, Knuth D, one Fermat ladder per inverse, the reference's affine
    formulas) pinned to the committed goldens -- written by the big-int walk -- on hot-path AND built-in-generator
    columns, to the walk itself on fresh random inputs, and its lock-step variant to the faithful one."""
    import oracle_c
    m = META[name]
    blind, args = _golden_arrays(name)
    _, _, cols0 = _case_ints(name)
    assert oracle_c.curve_program_num_cols(m["kind"], m["curve"]) == (m["num_cols"], m["num_aux"])
    cols, aux, err, flags = oracle_c.curve_program(m["kind"], m["curve"], blind, args)
    assert not err.any() and np.array_equal(cols[:, 0], cols0)
    for i, cs in enumerate(m["cases"]):
        assert _digest(cols[:, i]) == cs["cols_sha256"] and _digest(aux[:, i]) == cs["aux_sha256"], (name, i)
        assert int(flags[i]) == cs["valid"]
    cols2, aux2, err2, flags2 = oracle_c.curve_program(m["kind"], m["curve"], blind, args, lockstep=3)
    assert np.array_equal(cols, cols2) and np.array_equal(aux, aux2) and np.array_equal(flags, flags2) and not err2.any()
    # fresh inputs, a fresh blinding point: two elements against the big-int walk
    cv = CURVES[m["curve"]]
    rng = R.SplitMix64(1000 + 7 * m["kind"] + m["curve"])
    blind_i = cv.mul(rng.below(cv.n), cv.g)
    arr = lambda vs: np.stack([np.frombuffer(int(v).to_bytes(32, "little"), np.uint8).copy() for v in vs])
    if m["kind"] == 3:
        cases = [R.synth_signature_curve(cv, rng) for _ in range(2)]
    else:
        cases = []
        for _ in range(2):
            pt = cv.mul(rng.below(cv.n), cv.g)
            cases.append((pt[0], pt[1], rng.below(1 << 256)))       # scalars above the order are legal inputs
    cols, aux, err, flags = oracle_c.curve_program(m["kind"], m["curve"], blind_i, [arr([c[k] for c in cases]) for k in range(len(cases[0]))])
    assert not err.any()
    for i, case in enumerate(cases):
        wc, wa, _, ok = _walk(name, blind_i, case)
        assert np.array_equal(cols[:, i], np.asarray(wc, np.uint64)) and np.array_equal(aux[:, i], np.asarray(wa, np.uint64))
        assert int(flags[i]) == ok


def test_c_oracle_edge_inputs_of_the_curve_programs():
    """Where the reference panics (scalar 0: result == 2^264 * start, the unblinding add inverts zero; the caller's point
    equal to the blinding point) the C oracle flags INVERSE_OF_ZERO like the kernels do; a non-canonical point coordinate
    and scalars above the order agree with the walk."""
    import oracle_c
    rng = R.SplitMix64(5)
    arr = lambda vs: np.stack([np.frombuffer(int(v).to_bytes(32, "little"), np.uint8).copy() for v in vs])
    for ci, cv in enumerate(CURVES):
        blind = cv.mul(rng.below(cv.n), cv.g)
        pts = [cv.mul(rng.below(cv.n), cv.g) for _ in range(3)] + [blind]
        ks = [0, cv.n + 5, (1 << 256) - 1, 12345]
        xs = [p[0] for p in pts]
        if xs[2] + cv.p < 1 << 256:
            xs[2] += cv.p
        for kind, f in ((1, R.windowed_mul_witness), (2, R.scalar_mul_witness)):
            cols, aux, err, flags = oracle_c.curve_program(kind, ci, blind, (arr(xs), arr([p[1] for p in pts]), arr(ks)))
            assert err[0] & R.ERR_INVERSE_OF_ZERO
            for i in (1, 2):
                ref = f(cv, xs[i], pts[i][1], ks[i], blind)[0]
                assert err[i] == 0 and np.array_equal(cols[:, i], np.asarray(ref, np.uint64)), (cv.name, kind, i)
            try:
                f(cv, xs[3], pts[3][1], ks[3], blind)
                assert err[3] == 0
            except R.RefPanic as e:
                assert e.code == R.ERR_INVERSE_OF_ZERO and err[3] & R.ERR_INVERSE_OF_ZERO


def test_native_results_of_the_gadgets():
    """the gadgets compute k * P (curve/curve_multiplication.rs): the result limbs of the last curve_add"""
    rng = R.SplitMix64(99)
    for cv in CURVES:
        blind = cv.mul(rng.below(cv.n), cv.g)
        p = cv.mul(rng.below(cv.n), cv.g)
        k = rng.below(cv.n)
        assert R.windowed_mul_witness(cv, p[0], p[1], k, blind)[3] == cv.mul(k, p)
        assert R.scalar_mul_witness(cv, p[0], p[1], k, blind)[3] == cv.mul(k, p)
    # P-256: generator on the curve, order of the group (curve/p256.rs:67-81 test_generator_is_on_curve)
    assert R.P256.on_curve(R.P256.g) and R.P256.mul(R.P256.n, R.P256.g) is None


@pytest.mark.parametrize("name", PROGRAMS)
def test_goldens_pass_the_constraint_replay_and_mutations_do_not(name):
    blind, cases, cols0 = _case_ints(name)
    cols, aux, ops, ok = _walk(name, blind, cases[0])
    c = _replay(name, cols0, blind, cases[0], aux=aux)
    assert len(c.gens) == META[name]["num_gens"] and len(c.aux) == META[name]["num_aux"]
    # one corrupted column in each generator kind: rejected
    seen = {}
    for kind, field, c0, nc, label, _ in c.gens:
        seen.setdefault((kind, field), []).append((c0, nc))
    for (kind, field), blocks in seen.items():
        c0, nc = blocks[len(blocks) // 2]
        for off in (0, nc - 1):
            bad = cols0.copy()
            bad[c0 + off] = int(bad[c0 + off]) ^ 1
            with pytest.raises(CC.ConstraintViolation):
                _replay(name, bad, blind, cases[0])
    if META[name]["kind"] == 3:   # the tampered signature fails exactly its last connect
        colsx, _, _, okx = _walk(name, blind, cases[-1])
        assert okx == 0
        with pytest.raises(CC.ConstraintViolation) as e:
            _replay(name, colsx, blind, cases[-1])
        assert "connect_nonnative(r, point.x)" in str(e.value)


@pytest.mark.parametrize("name", PROGRAMS)
def test_kernel_bodies_match_the_goldens(name, emu):
    m = META[name]
    blind, args = _golden_arrays(name)
    _, _, cols0 = _case_ints(name)
    assert emu.sizes(m["kind"], m["curve"], blind) == (m["num_cols"], m["num_gens"], m["num_aux"])
    cols, err, valid, bad = emu.run(m["kind"], m["curve"], blind, args)
    assert bad == 0 and not err.any()
    assert np.array_equal(cols[:, 0], cols0)
    for i, cs in enumerate(m["cases"]):
        assert _digest(cols[:, i]) == cs["cols_sha256"], (name, i)
        assert int(valid[i]) == cs["valid"]
    # the cut into pieces (inversion batches) must not change any output
    cols2, _, valid2, _ = emu.run(m["kind"], m["curve"], blind, args, piece=7)
    assert np.array_equal(cols, cols2) and np.array_equal(valid, valid2)
    if m["kind"] != 2:
        # the large-batch plan: the windowed loop walked as runs (results inside a run keep no affine form in memory),
        # the verifier's fixed-base windows as one run per signature
        for windows_per_run in (6, 11):
            cols3, _, valid3, _ = emu.run(m["kind"], m["curve"], blind, args, piece=-windows_per_run)
            assert np.array_equal(cols, cols3) and np.array_equal(valid, valid3)


@pytest.mark.parametrize("name", PROGRAMS)
def test_other_targets_of_the_circuits(name, emu):
    """SURVEY 8(f) ranks 1 and 2 for the rank-4 gadgets: the built-in-generator values (split bits, digits, is_equal /
    not, random-access selections, bool products) against the gadget walk's own record, the gate-internal values and
    the constraint-block (U29 gate) values against the constraint replay's -- derived from the finished matrices by the
    bodies kc_aux / k_gate / kc_ux run."""
    m = META[name]
    blind, args = _golden_arrays(name)
    blind_i, cases, _ = _case_ints(name)
    cols, _, _, _ = emu.run(m["kind"], m["curve"], blind, args)
    aux, aerr = emu.aux(m["kind"], m["curve"], blind, args, cols)
    gate = emu.gate(m["kind"], m["curve"], blind, aux)
    ux, uerr = emu.ux(m["kind"], m["curve"], blind, args, cols, aux)
    assert aux.shape[0] == m["num_aux"] and not aerr.any() and not uerr.any()
    for i, cs in enumerate(m["cases"]):
        assert _digest(aux[:, i]) == cs["aux_sha256"]
        if not cs["valid"] or i > 1:
            continue                                        # (the replay of an invalid signature stops at its last connect)
        c = _replay(name, cols[:, i], blind_i, cases[i], aux=aux[:, i])
        assert np.array_equal(gate[:, i], np.asarray(c.gate, np.uint64)) and gate.shape[0] == len(c.gate)
        assert np.array_equal(ux[:, i], np.asarray(c.ux, np.uint64)) and ux.shape[0] == len(c.ux)
        assert int(ux[:, i].max()) < 1 << 29


def test_kernel_bodies_edge_inputs(emu):
    """scalar 0 ends in result == -(2^264 starting point): the reference panics in inverse() (gadgets/nonnative.rs:863),
    the kernels flag the element; scalars above the group order and a raw (non-canonical) point coordinate are
    legal inputs of the gadgets and must agree with the walk (mul_nonnative reads raw limbs, quirk Q3)."""
    rng = R.SplitMix64(5)
    for ci, cv in enumerate(CURVES):
        blind = cv.mul(rng.below(cv.n), cv.g)
        b = (np.frombuffer(blind[0].to_bytes(32, "little"), np.uint8).copy(), np.frombuffer(blind[1].to_bytes(32, "little"), np.uint8).copy())
        pts = [cv.mul(rng.below(cv.n), cv.g) for _ in range(3)]
        ks = [0, cv.n + 5, (1 << 256) - 1]
        xs = [pts[0][0], pts[1][0], pts[2][0]]
        if pts[2][0] + cv.p < 1 << 256:
            xs[2] = pts[2][0] + cv.p          # same field element, raw limbs differ
        arr = lambda vs: np.stack([np.frombuffer(int(v).to_bytes(32, "little"), np.uint8).copy() for v in vs])
        for kind, f in ((1, R.windowed_mul_witness), (2, R.scalar_mul_witness)):
            cols, err, valid, bad = emu.run(kind, ci, b, (arr(xs), arr([p[1] for p in pts]), arr(ks)))
            assert err[0] & R.ERR_INVERSE_OF_ZERO and bad == 1
            with pytest.raises(R.RefPanic):
                f(cv, xs[0], pts[0][1], ks[0], blind)
            for i in (1, 2):
                ref = f(cv, xs[i], pts[i][1], ks[i], blind)[0]
                assert err[i] == 0 and np.array_equal(cols[:, i], np.asarray(ref, np.uint64)), (cv.name, kind, i)


def test_degenerate_points_are_flagged_where_the_reference_panics(emu):
    """the gadgets use the incomplete affine formulas: the caller's point equal to the circuit's blinding point (or its
    negative) makes precompute_window add a point to itself / to its negative -> x2 - x1 = 0 -> the inverse generator
    panics in the reference (gadgets/nonnative.rs:863); every such element carries P2E_ERR_INVERSE_OF_ZERO, its
    neighbours in the batch are untouched"""
    rng = R.SplitMix64(321)
    arr = lambda vs: np.stack([np.frombuffer(int(v).to_bytes(32, "little"), np.uint8).copy() for v in vs])
    for ci, cv in enumerate(CURVES):
        blind = cv.mul(rng.below(cv.n), cv.g)
        b = (arr([blind[0]])[0], arr([blind[1]])[0])
        good = cv.mul(rng.below(cv.n), cv.g)
        pts = [good, blind, cv.neg(blind), good]
        ks = [rng.below(cv.n) for _ in pts]
        for kind, f in ((1, R.windowed_mul_witness), (2, R.scalar_mul_witness)):
            cols, err, valid, bad = emu.run(kind, ci, b, (arr([p[0] for p in pts]), arr([p[1] for p in pts]), arr(ks)))
            for i, pt in enumerate(pts):
                try:
                    ref = f(cv, pt[0], pt[1], ks[i], blind)[0]
                except R.RefPanic as e:
                    assert e.code == R.ERR_INVERSE_OF_ZERO and err[i] & R.ERR_INVERSE_OF_ZERO, (cv.name, kind, i)
                    continue
                assert err[i] == 0 and np.array_equal(cols[:, i], np.asarray(ref, np.uint64)), (cv.name, kind, i)
            assert err[0] == 0 and err[3] == 0 and (kind == 2 or (err[1] and err[2]))


def _tampered_p256_batch(n, seed):
    """valid signatures with every kind of defect mixed in: tampered msg / r / s / pk.x, r >= p, s = 0 (inverse of zero)"""
    import plonky2_ecdsa_amd as p2e  # host-side synthesis only (no GPU)
    cv = R.P256
    sig = [np.ascontiguousarray(a) for a in _synth_p256(n, seed)]
    sig[0][3, 5] ^= 1
    sig[1][4, 0] ^= 2
    sig[2][5, 31] ^= 0x40
    sig[3][6, 1] ^= 1                       # pk off the curve: curve_assert_valid's connect fails
    sig[1][7] = np.frombuffer(((1 << 256) - 1).to_bytes(32, "little"), np.uint8)   # r >= p: can never equal x
    sig[2][8] = 0                           # s = 0: the reference panics in inverse()
    return sig


def _synth_p256(n, seed):
    lib = C.CDLL(os.path.join(ROOT, "tests", "emu", "libp2e_emu.so"))
    out = [np.zeros((n, 32), np.uint8) for _ in range(5)]
    assert lib.emu_synth_signatures_curve(1, C.c_uint64(seed), C.c_size_t(0), C.c_size_t(n), *[a.ctypes.data_as(C.c_void_p) for a in out]) == 0
    return out


def test_p256_verdict_only_equals_the_full_fill(emu):
    """p2e_p256_verify_batch (no witness): same valid / err per signature as the full fill of the same inputs"""
    cv = R.P256
    blind_i = cv.mul(424242, cv.g)
    blind = (np.frombuffer(blind_i[0].to_bytes(32, "little"), np.uint8).copy(), np.frombuffer(blind_i[1].to_bytes(32, "little"), np.uint8).copy())
    sig = _tampered_p256_batch(12, 6)
    _, err, valid, bad = emu.run(3, 1, blind, sig)
    emu.L.emu_p256_verify_only.restype = C.c_long
    err2, valid2 = np.zeros(12, np.uint8), np.zeros(12, np.uint8)
    bad2 = emu.L.emu_p256_verify_only(emu._p(blind[0]), emu._p(blind[1]), *[emu._p(a) for a in sig], C.c_size_t(12), emu._p(err2), emu._p(valid2))
    assert bad == bad2 and np.array_equal(err != 0, err2 != 0) and np.array_equal(valid, valid2)
    assert list(valid) == [1, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1, 1] and err[8] & R.ERR_INVERSE_OF_ZERO


@pytest.mark.parametrize("name", PROGRAMS)
def test_builder_wiring_equals_the_replayed_gadget_wiring(name, emu):
    """the C++ schedule builder's generator table and operand wiring (p2e_curve_program_describe / _wiring) against the
    sources the constraint replay records while it walks the reference's gadgets"""
    m = META[name]
    blind_i, cases, cols0 = _case_ints(name)
    blind, _ = _golden_arrays(name)
    c = _replay(name, cols0, blind_i, cases[0])
    kinds, fields, first, ncols, src, nl = emu.gens(m["kind"], m["curve"], blind)
    names = ("add", "sub", "add_many", "mul", "inv", "glv")
    slot = {"pky": 0, "py": 0, "pkx": 1, "px": 1, "msg": 2, "k": 2, "r": 3, "s": 4}
    assert len(c.gens) == len(kinds)
    consts = {}
    for g, (kind, field, c0, nc, label, operands) in enumerate(c.gens):
        assert (names[kinds[g]], int(fields[g]), int(first[g]), int(ncols[g])) == (kind, field, c0, nc), (g, label)
        for k, limbs in enumerate(operands):
            limbs, s, n_l = list(limbs), int(src[g, k]), int(nl[g, k])
            assert len(limbs) == n_l, (label, kind, k)
            if s & SRC_CONST:
                cid = s & 0xFFFF
                if cid not in consts:
                    consts[cid] = emu.const(m["kind"], m["curve"], blind, cid)
                want = R.const_limbs(consts[cid])
                if n_l == len(want):
                    assert [v for *_, v in limbs] == want or [x[0] for x in limbs] == ["virt"] * n_l
                else:   # curve_scalar_mul's virtual `result` carries the constant in 9 limbs
                    assert limbs[0][0] == "virt" and n_l == R.NL
                continue
            if n_l == 0:
                continue
            f0 = limbs[0]
            if f0[0] == "col":
                assert s == f0[1] and limbs == [("col", f0[1] + j) for j in range(n_l)]
            elif f0[0] == "aux":
                assert s == SRC_AUX | f0[1] and limbs == [("aux", f0[1] + j) for j in range(n_l)], (label, kind, k)
            else:
                assert f0[0] == "in" and s == SRC_INPUT | slot[f0[1]], (label, f0)


def test_p256_field_arithmetic_against_bigints(emu):
    """Barrett reduction with exact quotient (csrc/fe.hpp reduce_barrett) through the mul / inv generators of the
    verifier's scalar phase: u1 = msg * s^-1, u2 = r * s^-1 (mod n) and their quotient limbs, on structured operands"""
    cv = R.P256
    rng = R.SplitMix64(77)
    blind = cv.mul(rng.below(cv.n), cv.g)
    b = (np.frombuffer(blind[0].to_bytes(32, "little"), np.uint8).copy(), np.frombuffer(blind[1].to_bytes(32, "little"), np.uint8).copy())
    pk = cv.mul(rng.below(cv.n), cv.g)
    vals = [1, 2, cv.n - 1, cv.n - 2, (1 << 255), (1 << 256) - 1, cv.n + 1, 0xFFFFFFFF00000000FFFFFFFF, rng.below(cv.n)]
    arr = lambda vs: np.stack([np.frombuffer(int(v).to_bytes(32, "little"), np.uint8).copy() for v in vs])
    n = len(vals)
    msg, r, s = arr(vals), arr(vals[::-1]), arr([vals[(i * 5 + 2) % n] for i in range(n)])
    cols, err, valid, bad = emu.run(3, 1, b, (msg, r, s, arr([pk[0]] * n), arr([pk[1]] * n)))
    for i in range(n):
        try:
            ref = R.verify_p256_witness(_int(msg[i]), _int(r[i]), _int(s[i]), pk[0], pk[1], blind)[0]
        except R.RefPanic:
            assert err[i] != 0
            continue
        assert np.array_equal(cols[:, i], np.asarray(ref, np.uint64)), i


def test_p256_multiplication_free_reduction_against_barrett(emu):
    """csrc/fe.hpp reduce_p256_solinas (the chains' and inversions' reduction: NIST fast reduction, no quotient) against
    reduce_barrett on 2 000 000 products of random / structured operands and raw 512-bit values"""
    emu.L.emu_p256_reduce_selfcheck.restype = C.c_long
    assert emu.L.emu_p256_reduce_selfcheck(C.c_long(2000000), C.c_uint64(2024)) == 0


# ---- GPU ------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def gpu():
    import torch
    import plonky2_ecdsa_amd as p2e
    assert torch.cuda.is_available(), "the -m gpu tests need the MI355X"
    return p2e, torch, p2e.Context(device=0)


def _gpu_run(gpu, name_or_kind, curve, blind_i, args):
    p2e, torch, ctx = gpu
    prog = p2e.CurveProgram(ctx, name_or_kind, curve, blind_i)
    dev = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in args]
    if len(args) == 3:
        cols, err, valid, bad = prog.mul_witness_batch(*dev)
    else:
        cols, err, valid, bad = prog.verify_witness_batch(*dev)
    torch.cuda.synchronize()
    out = cols.cpu().numpy().view(np.uint64), err.cpu().numpy(), valid.cpu().numpy(), bad
    return prog, out


@pytest.mark.gpu
@pytest.mark.parametrize("name", PROGRAMS)
def test_gpu_matches_the_goldens(name, gpu):
    m = META[name]
    blind_i, cases, cols0 = _case_ints(name)
    _, args = _golden_arrays(name)
    prog, (cols, err, valid, bad) = _gpu_run(gpu, m["kind"], m["curve"], blind_i, args)
    assert prog.num_cols == m["num_cols"] and prog.num_aux_cols == m["num_aux"] and bad == 0
    assert np.array_equal(cols[:, 0], cols0)
    for i, cs in enumerate(m["cases"]):
        assert _digest(cols[:, i]) == cs["cols_sha256"] and int(valid[i]) == cs["valid"]
    # the output of the HIP kernels through the constraint replay: no oracle value involved
    _replay(name, cols[:, 1], blind_i, cases[1])
    prog.close()


def _ragged_inputs(p2e, m, n, seed):
    sig = p2e.synth_signatures_curve(m["curve"], seed=seed, n=n)
    if m["kind"] == 3:
        sig[0][7, 3] ^= 0x10            # one tampered message
        return sig
    return (sig[3], sig[4], sig[0])       # the public keys as points, the messages as scalars


def _blind_of(cv, rng):
    blind_i = cv.mul(rng.below(cv.n), cv.g)
    return blind_i, (np.frombuffer(blind_i[0].to_bytes(32, "little"), np.uint8).copy(), np.frombuffer(blind_i[1].to_bytes(32, "little"), np.uint8).copy())


@pytest.mark.gpu
@pytest.mark.parametrize("plan", ["four_lanes", "lane_per_signature"])
@pytest.mark.parametrize("name", PROGRAMS)
def test_gpu_ragged_batch_against_the_c_oracle_and_the_walk(name, plan, gpu, emu, monkeypatch):
    """300 random inputs (one full workgroup of paired stores + a ragged tail): every column of every input against the
    independent C oracle (oracle/p2e_oracle.c: the reference's affine formulas, Knuth D, Fermat inverses -- no code shared
    with the kernels), three of them against the big-int walk as well.  Both small-batch forms of phases A / B: four lanes
    per signature with split inversion batches (what n <= 17 408 takes since round 3) and one lane per signature (what
    17 408 < n < 49 152 takes), each op expanded on its own."""
    import oracle_c
    p2e, torch, ctx = gpu
    if plan == "lane_per_signature":
        monkeypatch.setenv("P2E_QUAD_MAX_N", "0")
        ctx = p2e.Context(device=0)
        gpu = (p2e, torch, ctx)
    m = META[name]
    cv = CURVES[m["curve"]]
    blind_i, blind = _blind_of(cv, R.SplitMix64(1234 + m["kind"] + 10 * m["curve"]))
    n = 300
    args = _ragged_inputs(p2e, m, n, 42 + m["kind"])
    prog, (cols, err, valid, bad) = _gpu_run(gpu, m["kind"], m["curve"], blind_i, args)
    ocols, _oaux, oerr, oflags = oracle_c.curve_program(m["kind"], m["curve"], blind, args, want_aux=False)
    assert bad == 0 and not oerr.any() and np.array_equal(err, oerr) and np.array_equal(valid, oflags)
    assert np.array_equal(cols, ocols)
    if m["kind"] == 3:
        assert valid.sum() == n - 1 and valid[7] == 0
    for i in (0, 7, n - 1):
        ref = _walk(name, blind_i, [_int(a[i]) for a in args])[0]
        assert np.array_equal(cols[:, i], np.asarray(ref, np.uint64))
    # the library's tables against the emulation's (same builder, but through the C ABI)
    kinds, fields, first, ncols, src, nl = emu.gens(m["kind"], m["curve"], blind)
    names = ("add", "sub", "add_many", "mul", "inv", "glv")
    desc, wiring = prog.describe(), prog.wiring()
    assert len(desc) == len(kinds) == m["num_gens"]
    for g in (0, 1, len(desc) // 2, len(desc) - 1):
        assert desc[g][:4] == (names[kinds[g]], int(fields[g]), int(first[g]), int(ncols[g]))
        assert wiring[g][0] == [(int(src[g, k]), int(nl[g, k])) for k in range(len(wiring[g][0]))]
    assert prog.const(0) == blind_i[0] or m["kind"] == 3
    assert sum(d[2] for d in prog.aux_describe()) == m["num_aux"]
    prog.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", [n for n in PROGRAMS if META[n]["kind"] != 2])
def test_gpu_run_plan_of_the_curve_programs_on_a_ragged_batch(name, monkeypatch):
    """The LARGE-BATCH plan of the windowed programs (taken from 49 152 inputs per call: kc_expand_runs over pieces of 6
    windows, the d_ops_runs F_NO_AFFINE marks, the verifier's fixed-base chain on its own stream as one run per signature,
    the reordered phase C -- curve_api.inc run_curve_program), forced onto a ragged batch of 700 with P2E_CP_RUNS_MIN_N=1
    (read in p2e_ctx_create) and compared on every column of every input with the C oracle."""
    import oracle_c
    import torch
    import plonky2_ecdsa_amd as p2e
    monkeypatch.setenv("P2E_CP_RUNS_MIN_N", "1")
    ctx = p2e.Context(device=0)
    m = META[name]
    cv = CURVES[m["curve"]]
    blind_i, blind = _blind_of(cv, R.SplitMix64(555 + m["kind"] + 10 * m["curve"]))
    n = 700
    args = _ragged_inputs(p2e, m, n, 4242 + m["kind"])
    prog, (cols, err, valid, bad) = _gpu_run((p2e, torch, ctx), m["kind"], m["curve"], blind_i, args)
    ph = ctx.last_phase_ms()
    assert ph["runs_launches"] > 0, "the run plan was not taken"
    ocols, _oaux, oerr, oflags = oracle_c.curve_program(m["kind"], m["curve"], blind, args, want_aux=False)
    assert bad == 0 and np.array_equal(err, oerr) and np.array_equal(valid, oflags)
    assert np.array_equal(cols, ocols)
    if m["kind"] == 3:
        assert ph["fbrun_launches"] > 0 and valid.sum() == n - 1 and valid[7] == 0
    for i in (0, n - 1):
        ref = _walk(name, blind_i, [_int(a[i]) for a in args])[0]
        assert np.array_equal(cols[:, i], np.asarray(ref, np.uint64))
    prog.close()


@pytest.mark.gpu
@pytest.mark.parametrize("plan", ["op_by_op", "runs"])
@pytest.mark.parametrize("name", PROGRAMS)
def test_gpu_compact_container_of_the_curve_programs(name, plan, monkeypatch):
    """The curve programs writing the compact container directly (u32 narrow + u64 wide matrices, padded strides, a ragged
    batch: paired 8- / 16-byte stores + the tail kernels), both launch plans: expanded on the host it must be the C
    oracle's matrix on every column of every input; the pad columns stay untouched."""
    import oracle_c
    import torch
    import plonky2_ecdsa_amd as p2e
    m = META[name]
    if plan == "runs":
        if m["kind"] == 2:
            pytest.skip("curve_scalar_mul has no window loop: one plan only")
        monkeypatch.setenv("P2E_CP_RUNS_MIN_N", "1")
    ctx = p2e.Context(device=0)
    cv = CURVES[m["curve"]]
    blind_i, blind = _blind_of(cv, R.SplitMix64(777 + m["kind"] + 10 * m["curve"]))
    n = 600 + 37
    args = _ragged_inputs(p2e, m, n, 99 + m["kind"])
    prog = p2e.CurveProgram(ctx, m["kind"], m["curve"], blind_i)
    cmap, nn, nw = prog.compact_layout()
    assert nn + nw == m["num_cols"] and len(cmap) == m["num_cols"]
    nar = torch.full((nn, n + 3), -1, dtype=torch.int32, device="cuda")
    wid = torch.full((nw, n + 5), -1, dtype=torch.int64, device="cuda")
    dev = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in args]
    if m["kind"] == 3:
        _, _, err, valid, bad = prog.verify_witness_compact_batch(*dev, narrow=nar[:, :n], wide=wid[:, :n], ld_narrow=n + 3, ld_wide=n + 5)
    else:
        _, _, err, valid, bad = prog.mul_witness_compact_batch(*dev, narrow=nar[:, :n], wide=wid[:, :n], ld_narrow=n + 3, ld_wide=n + 5)
    torch.cuda.synchronize()
    ocols, _a, oerr, oflags = oracle_c.curve_program(m["kind"], m["curve"], blind, args, want_aux=False)
    got = prog.compact_expand(nar[:, :n].cpu().numpy(), wid[:, :n].cpu().numpy())
    assert bad == 0 and np.array_equal(err.cpu().numpy(), oerr) and np.array_equal(valid.cpu().numpy(), oflags)
    assert np.array_equal(got, ocols)
    assert bool((nar[:, n:] == -1).all()) and bool((wid[:, n:] == -1).all())
    # even strides: the paired-store kernels; same container
    nar2, wid2, _, _, bad2 = (prog.verify_witness_compact_batch(*dev) if m["kind"] == 3 else prog.mul_witness_compact_batch(*dev))
    torch.cuda.synchronize()
    assert bad2 == 0 and torch.equal(nar2, nar[:, :n]) and torch.equal(wid2, wid[:, :n])
    prog.close()


@pytest.mark.gpu
def test_gpu_p256_verifier_full_size_batch_every_signature():
    """verify_p256_message_circuit at 2^16 per call -- the plan bench.py's p256_verify leg reports (runs, fixed-base run,
    its own chain stream) -- compared with the C oracle's lock-step walk on EVERY signature and column: 2 048 distinct
    signatures (host-side P-256 signing is two generic scalar multiplications each) tiled over the batch, two of them
    tampered; each of the 32 tiles of the GPU matrix must equal the oracle's 2 048 columns."""
    import oracle_c
    import torch
    import plonky2_ecdsa_amd as p2e
    ctx = p2e.Context(device=0)
    cv = R.P256
    blind_i, blind = _blind_of(cv, R.SplitMix64(2468))
    n, distinct = 1 << 16, 2048
    base = p2e.synth_signatures_curve(p2e.CURVE_P256, seed=4, n=distinct)
    base[2][100, 3] ^= 0x08                 # a tampered s
    base[0][distinct - 1, 0] ^= 1           # a tampered message in the last lane of every tile
    sig = [torch.from_numpy(np.tile(a, (n // distinct, 1)).copy()).cuda() for a in base]
    prog = p2e.CurveProgram(ctx, p2e.CP_VERIFY, p2e.CURVE_P256, blind_i)
    cols, err, valid, bad = prog.verify_witness_batch(*sig)
    torch.cuda.synchronize()
    ph = ctx.last_phase_ms()
    assert ph["runs_launches"] > 0 and ph["fbrun_launches"] > 0, "2^16 per call must take the run plan"
    want, _, werr, wflags = oracle_c.curve_program(p2e.CP_VERIFY, p2e.CURVE_P256, blind, base, lockstep=64, want_aux=False)
    assert bad == 0 and not werr.any() and wflags.sum() == distinct - 2
    want_t = torch.from_numpy(want.view(np.int64)).cuda()
    e, v = err.cpu().numpy(), valid.cpu().numpy()
    for t in range(n // distinct):
        a, b = t * distinct, (t + 1) * distinct
        assert torch.equal(cols[:, a:b], want_t), f"tile {t} differs from the oracle"
        assert not e[a:b].any() and np.array_equal(v[a:b], wflags)
    prog.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", PROGRAMS)
def test_gpu_other_targets_of_the_circuits(name, gpu, emu):
    """aux / gate-internal / constraint-block matrices of a ragged batch from the library's own witness matrix: the
    built-in-generator values against the C oracle's own record of them on every column of every input; the gate-internal
    and U29-gate values (whose only independent model is the Python constraint replay) against the CPU-compiled bodies on
    every column and against the replay -- which recomputes them from the GPU's hot-path columns alone -- on six inputs"""
    import oracle_c
    p2e, torch, ctx = gpu
    m = META[name]
    cv = CURVES[m["curve"]]
    blind_i = cv.mul(0xBEEF + m["kind"], cv.g)
    blind = (np.frombuffer(blind_i[0].to_bytes(32, "little"), np.uint8).copy(), np.frombuffer(blind_i[1].to_bytes(32, "little"), np.uint8).copy())
    n = 300
    sig = p2e.synth_signatures_curve(m["curve"], seed=77 + m["kind"], n=n)
    args = sig if m["kind"] == 3 else (sig[3], sig[4], sig[0])
    prog = p2e.CurveProgram(ctx, m["kind"], m["curve"], blind_i)
    dev = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in args]
    cols, err, valid, bad = prog.mul_witness_batch(*dev) if m["kind"] != 3 else prog.verify_witness_batch(*dev)
    aux, aerr, abad = prog.aux_witness_batch(dev, cols)
    ux32, uerr, ubad = prog.ux_witness_batch(dev, cols, aux, u32=True)
    ux64, _, _ = prog.ux_witness_batch(dev, cols, aux, u32=False)
    gate = prog.gate_internal_batch(aux) if prog.num_gate_cols else None
    torch.cuda.synchronize()
    assert bad == abad == ubad == 0
    h_cols, h_aux = cols.cpu().numpy().view(np.uint64), aux.cpu().numpy().view(np.uint64)
    o_cols, o_aux, o_err, _ = oracle_c.curve_program(m["kind"], m["curve"], blind, args)
    assert not o_err.any() and np.array_equal(h_cols, o_cols) and np.array_equal(h_aux, o_aux)
    e_aux, _ = emu.aux(m["kind"], m["curve"], blind, args, h_cols)
    assert np.array_equal(h_aux, e_aux)
    e_ux, _ = emu.ux(m["kind"], m["curve"], blind, args, h_cols, h_aux)
    assert np.array_equal(ux64.cpu().numpy().view(np.uint64), e_ux)
    assert np.array_equal(ux32.cpu().numpy().view(np.uint32).astype(np.uint64), e_ux)
    assert (prog.num_aux_cols, prog.num_ux_cols) == (e_aux.shape[0], e_ux.shape[0])
    assert sum(nc for _, nc in prog.ux_describe()) == prog.num_ux_cols
    if gate is not None:
        assert np.array_equal(gate.cpu().numpy().view(np.uint64), emu.gate(m["kind"], m["curve"], blind, h_aux))
    for i in (0, 1, 63, 64, 255, n - 1):
        c = _replay(name, h_cols[:, i], blind_i, [_int(a[i]) for a in args], aux=h_aux[:, i])
        assert np.array_equal(e_ux[:, i], np.asarray(c.ux, np.uint64))
        if gate is not None:
            assert np.array_equal(gate.cpu().numpy().view(np.uint64)[:, i], np.asarray(c.gate, np.uint64))
    prog.close()


@pytest.mark.gpu
def test_gpu_wire_assembly_of_a_curve_program(gpu):
    """SURVEY 8(f) rank 3 for the P-256 verifier: its four matrices scattered into one wire matrix per signature through
    a random host-supplied map (p2e_curve_program_wire_map_create + p2e_assemble_wires), against numpy"""
    p2e, torch, ctx = gpu
    cv = R.P256
    prog = p2e.CurveProgram(ctx, p2e.CP_VERIFY, p2e.CURVE_P256, cv.mul(31337, cv.g))
    n = 80
    dev = [torch.from_numpy(a).cuda() for a in p2e.synth_signatures_curve(p2e.CURVE_P256, seed=8, n=n)]
    cols, _, valid, bad = prog.verify_witness_batch(*dev)
    aux, _, _ = prog.aux_witness_batch(dev, cols)
    ux, _, _ = prog.ux_witness_batch(dev, cols, aux, u32=True)
    gate = prog.gate_internal_batch(aux)
    torch.cuda.synchronize()
    assert bad == 0 and int(valid.sum()) == n
    mats = [cols.cpu().numpy().view(np.uint64), aux.cpu().numpy().view(np.uint64), ux.cpu().numpy().view(np.uint32), gate.cpu().numpy().view(np.uint64)]
    limits = [prog.num_cols, prog.num_aux_cols, prog.num_ux_cols, prog.num_gate_cols]
    rng = np.random.default_rng(9)
    cnt, cells = 60_003, 1 << 17
    kinds = rng.integers(0, 4, cnt).astype(np.uint32)
    colsel = np.array([rng.integers(0, limits[k]) for k in kinds], dtype=np.uint32)
    src = (kinds << 30) | colsel
    dst = rng.permutation(cells)[:cnt].astype(np.uint32)
    wm = prog.wire_map(src, dst, 128, cells // 128)
    wires = ctx.assemble_wires(wm, cols, aux, ux, gate)
    torch.cuda.synchronize()
    want = np.zeros((n, cells), np.uint64)
    for k in range(4):
        sel = kinds == k
        want[:, dst[sel]] = mats[k][colsel[sel]].T
    assert np.array_equal(wires.cpu().numpy().view(np.uint64), want)
    with pytest.raises(p2e.P2EError):                            # a source column beyond this program's witness matrix
        prog.wire_map(np.array([prog.num_cols], np.uint32), np.array([0], np.uint32), 4, 4)
    prog.close()


@pytest.mark.gpu
def test_gpu_p256_verdict_only(gpu):
    """the pre-filter (no witness) against the full fill on a batch with every kind of defect, ragged size"""
    p2e, torch, ctx = gpu
    cv = R.P256
    prog = p2e.CurveProgram(ctx, p2e.CP_VERIFY, p2e.CURVE_P256, cv.mul(424242, cv.g))
    n = 700
    sig = _tampered_p256_batch(n, 6)
    dev = [torch.from_numpy(a).cuda() for a in sig]
    _, err, valid, bad = prog.verify_witness_batch(*dev)
    err2, valid2, bad2 = prog.verify_batch(*dev)
    torch.cuda.synchronize()
    assert bad == bad2 == 1
    assert np.array_equal(err.cpu().numpy() != 0, err2.cpu().numpy() != 0) and np.array_equal(valid.cpu().numpy(), valid2.cpu().numpy())
    assert int(valid2.sum()) == n - 6
    prog.close()


@pytest.mark.gpu
def test_gpu_streamed_chunks_of_the_p256_verifier(emu):
    """the BASELINE config 5 shape for the P-256 verifier: 3 x 1 024 + 300 signatures streamed to pinned host memory in
    double-buffered chunks (plonky2_ecdsa_amd.stream.HostStreamer with a curve program), EVERY chunk checked on the host
    copy against the CPU-compiled kernel bodies, as columns and as per-signature rows"""
    import torch
    import plonky2_ecdsa_amd as p2e
    from plonky2_ecdsa_amd.stream import HostStreamer
    cv = R.P256
    blind_i = cv.mul(777, cv.g)
    blind = (np.frombuffer(blind_i[0].to_bytes(32, "little"), np.uint8).copy(), np.frombuffer(blind_i[1].to_bytes(32, "little"), np.uint8).copy())
    total, chunk = 3 * 1024 + 300, 1024
    sig = p2e.synth_signatures_curve(p2e.CURVE_P256, seed=21, n=total)
    sig[0][2000, 9] ^= 4
    want, werr, wvalid, _ = emu.run(p2e.CP_VERIFY, p2e.CURVE_P256, blind, sig)
    dev = [torch.from_numpy(a).cuda() for a in sig]
    for container in ("u64", "rows"):
        hs = HostStreamer(device=0, chunk=chunk, container=container, curve_program=(p2e.CP_VERIFY, p2e.CURVE_P256, blind_i))
        seen = []

        def consumer(ch):
            got = ch.cols.numpy().view(np.uint64) if container == "u64" else ch.rows.numpy().view(np.uint64).T
            assert np.array_equal(got, want[:, ch.first:ch.first + ch.n]), (container, ch.index)
            assert np.array_equal(ch.valid, wvalid[ch.first:ch.first + ch.n])
            seen.append(ch.n)
        res = hs.run(dev, consumer)
        assert seen == [1024, 1024, 1024, 300] and res["flagged"] == 0 and res["valid"] == total - 1
        hs.prog.close()


@pytest.mark.gpu
def test_gpu_p256_verify_batch_properties(gpu):
    """4 096 + 77 P-256 verifies: every valid signature verifies, every limb column is a 29-bit limb, tampering with
    any of the five inputs clears `valid`, sampled signatures pass the constraint replay"""
    p2e, torch, ctx = gpu
    cv = R.P256
    blind_i = cv.mul(0xC0FFEE, cv.g)
    n = 4096 + 77
    sig = p2e.synth_signatures_curve(p2e.CURVE_P256, seed=9, n=n)
    for k in range(5):
        sig[k][100 + k, 1] ^= 1
    prog, (cols, err, valid, bad) = _gpu_run(gpu, p2e.CP_VERIFY, p2e.CURVE_P256, blind_i, sig)
    assert bad == 0 and not err.any()
    expect = np.ones(n, np.uint8)
    expect[100:105] = 0
    assert np.array_equal(valid, expect)
    desc = prog.describe()
    for kind, field, c0, nc, label in desc[:: max(1, len(desc) // 97)]:
        limbs = 9 if kind in ("add", "sub", "add_many") else 18
        assert int(cols[c0:c0 + limbs].max()) < 1 << 29, (kind, label)
    for i in (0, n - 1):
        CC.check_verify_p256(cols[:, i], *[_int(a[i]) for a in sig], blind_i)
    prog.close()


@pytest.mark.gpu
@pytest.mark.parametrize("plan", ["four_lanes", "lane_per_signature"])
def test_gpu_edge_inputs_of_the_curve_programs(plan, monkeypatch):
    """Where the reference panics the kernels must flag the element and leave its neighbours alone, in both small-batch
    plans: scalar 0 (the unblinding add inverts zero), the caller's point equal to the circuit's blinding point or its
    negative (precompute_window adds a point to itself / its negative); scalars above the group order and a non-canonical
    point coordinate are legal.  Error bytes against the C oracle on every element, columns on every element it does not flag."""
    import oracle_c
    import torch
    import plonky2_ecdsa_amd as p2e
    if plan == "lane_per_signature":
        monkeypatch.setenv("P2E_QUAD_MAX_N", "0")
    ctx = p2e.Context(device=0)
    rng = R.SplitMix64(2718)
    arr = lambda vs: np.stack([np.frombuffer(int(v).to_bytes(32, "little"), np.uint8).copy() for v in vs])
    for ci, cv in enumerate(CURVES):
        blind_i, blind = _blind_of(cv, rng)
        good = [cv.mul(rng.below(cv.n), cv.g) for _ in range(70)]
        pts = list(good)
        ks = [rng.below(cv.n) for _ in pts]
        ks[3], ks[40], ks[69] = 0, cv.n + 5, (1 << 256) - 1
        pts[10], pts[64] = blind_i, cv.neg(blind_i)
        xs = [p[0] for p in pts]
        if xs[20] + cv.p < 1 << 256:
            xs[20] += cv.p                                      # same field element, raw limbs differ (quirk Q3)
        args = (arr(xs), arr([p[1] for p in pts]), arr(ks))
        for kind in (p2e.CP_WINDOWED_MUL, p2e.CP_SCALAR_MUL):
            prog, (cols, err, valid, bad) = _gpu_run((p2e, torch, ctx), kind, ci, blind_i, args)
            ocols, _a, oerr, _f = oracle_c.curve_program(kind, ci, blind, args, want_aux=False)
            assert np.array_equal(err, oerr) and bad == int((oerr != 0).sum()), (cv.name, kind)
            assert oerr[3] & R.ERR_INVERSE_OF_ZERO and oerr[0] == 0 and oerr[40] == 0 and oerr[69] == 0 and oerr[20] == 0
            if kind == p2e.CP_WINDOWED_MUL:
                assert oerr[10] and oerr[64]
            clean = oerr == 0
            assert np.array_equal(cols[:, clean], ocols[:, clean]), (cv.name, kind)
            prog.close()


@pytest.mark.gpu
def test_gpu_curve_program_misuse(gpu):
    p2e, torch, ctx = gpu
    with pytest.raises(p2e.P2EError):
        p2e.CurveProgram(ctx, p2e.CP_VERIFY, p2e.CURVE_SECP256K1, R.SECP256K1.g)       # the verifier program is P-256 only
    with pytest.raises(p2e.P2EError):
        p2e.CurveProgram(ctx, 9, p2e.CURVE_P256, R.P256.g)
    with pytest.raises(p2e.P2EError):
        p2e.CurveProgram(ctx, p2e.CP_WINDOWED_MUL, p2e.CURVE_P256, ((1 << 256) - 1, 5))   # not a canonical field element
    with pytest.raises(p2e.P2EError) as e:                                               # a point of the OTHER curve
        p2e.CurveProgram(ctx, p2e.CP_WINDOWED_MUL, p2e.CURVE_P256, R.SECP256K1.g)
    assert "not on the curve" in str(e.value)
    with pytest.raises(p2e.P2EError):
        p2e.CurveProgram(ctx, p2e.CP_SCALAR_MUL, p2e.CURVE_SECP256K1, (R.SECP256K1.g[0], R.SECP256K1.g[1] ^ 1))
    prog = p2e.CurveProgram(ctx, p2e.CP_WINDOWED_MUL, p2e.CURVE_P256, R.P256.g)
    sig = [torch.from_numpy(a).cuda() for a in p2e.synth_signatures_curve(p2e.CURVE_P256, seed=1, n=8)]
    with pytest.raises(p2e.P2EError):
        prog.verify_witness_batch(*sig)                                                 # a multiplication program
    cols, err, valid, bad = prog.mul_witness_batch(sig[3], sig[4], sig[0])              # a valid call still works
    torch.cuda.synchronize()
    assert bad == 0
    prog.close()


@pytest.mark.gpu
def test_gpu_curve_program_host_pointer_mode_and_failed_allocation(emu):
    """P2E_CTX_HOST_POINTERS: numpy in, numpy out through the library's staging buffers (the path a ctypes-only caller
    takes); an impossible batch size fails with a status (P2E_E_NOMEM = -4), frees what it staged, and the next call on
    the same context succeeds"""
    import plonky2_ecdsa_amd as p2e
    cv = R.P256
    blind_i = cv.mul(0xABCDEF, cv.g)
    blind = (np.frombuffer(blind_i[0].to_bytes(32, "little"), np.uint8).copy(), np.frombuffer(blind_i[1].to_bytes(32, "little"), np.uint8).copy())
    ctx = p2e.Context(device=0, host_pointers=True)
    prog = p2e.CurveProgram(ctx, p2e.CP_VERIFY, p2e.CURVE_P256, blind_i)
    n = 70
    sig = p2e.synth_signatures_curve(p2e.CURVE_P256, seed=3, n=n)
    cols, err, valid, bad = prog.verify_witness_batch(*sig)
    want, _, wvalid, _ = emu.run(p2e.CP_VERIFY, p2e.CURVE_P256, blind, sig)
    assert bad == 0 and np.array_equal(np.asarray(cols).view(np.uint64), want) and np.array_equal(np.asarray(valid), wvalid)
    aux, aerr, abad = prog.aux_witness_batch(sig, cols)
    assert abad == 0 and np.array_equal(np.asarray(aux).view(np.uint64), emu.aux(p2e.CP_VERIFY, p2e.CURVE_P256, blind, sig, want)[0])
    # a batch whose scratch cannot be allocated: claim 2^36 signatures with buffers the call never gets to touch
    huge = 1 << 36
    L = ctx._L
    L.p2e_p256_verify_witness_batch.restype = C.c_long
    rc = L.p2e_p256_verify_witness_batch(ctx._h, prog._h, *[a.ctypes.data_as(C.c_void_p) for a in sig],
                                         np.asarray(cols).ctypes.data_as(C.c_void_p), C.c_size_t(huge), C.c_size_t(huge),
                                         np.asarray(err).ctypes.data_as(C.c_void_p), np.asarray(valid).ctypes.data_as(C.c_void_p))
    assert rc == -4, (rc, L.p2e_last_error())
    cols2, _, valid2, bad2 = prog.verify_witness_batch(*sig)
    assert bad2 == 0 and np.array_equal(np.asarray(cols2).view(np.uint64), want)
    prog.close()

# ---- the crate's generators as stand-alone batch entry points over the P-256 fields (P2E_FIELD_P256_BASE / _SCALAR) ----
def _p256_generator_cases(field, count, seed):
    """limb-column operands: random canonical values, values around 0 / m / 2^256, raw 261-bit limb patterns"""
    m = R.MODULI[field]
    rng = R.SplitMix64(seed)
    edge = [0, 1, 2, m - 1, m, m + 1, (1 << 255), (1 << 256) - 1, m // 2, (m + 1) // 2]
    xs, ys = [], []
    for i in range(count):
        a = edge[i % len(edge)] if i % 3 == 0 else rng.below(m)
        b = edge[(i // len(edge)) % len(edge)] if i % 5 == 0 else rng.below(m)
        xs.append(R.limbs_of(a, R.NL))
        ys.append(R.limbs_of(b, R.NL))
    for i in range(count // 4):                                 # all nine limbs used: operands up to 2^261 - 1
        xs.append([rng.next() & R.MASK29 if (i + k) % 4 else R.MASK29 for k in range(R.NL)])
        ys.append([rng.next() & R.MASK29 for k in range(R.NL)])
    return xs, ys


def _check_p256_generators(be, field, count=160):
    m = R.MODULI[field]
    xs, ys = _p256_generator_cases(field, count, 1000 + field)
    cols = lambda ls: np.array(ls, dtype=np.uint64).T.copy()
    x, y = cols(xs), cols(ys)
    r, q, cs, b, err = [np.asarray(a) for a in be.mul(field, x, y)]
    for i, (xl, yl) in enumerate(zip(xs, ys)):
        try:
            wr, wq, wcs = R.gen_mul(xl, yl, m)
            wb = R.gen_checksum(wcs)
        except R.RefPanic:
            assert err[i] != 0, ("mul", i)
            continue
        assert err[i] == 0 and list(r[:, i]) == wr and list(q[:, i]) == wq and list(cs[:, i]) == wcs and list(b[:, i]) == wb, ("mul", i)
    for name, gen in (("add", R.gen_add), ("sub", R.gen_sub)):
        out, ov, err = [np.asarray(a) for a in getattr(be, name)(field, x, y)]
        for i, (xl, yl) in enumerate(zip(xs, ys)):
            try:
                wo, wov = gen(xl, yl, m)
            except R.RefPanic:
                assert err[i] != 0, (name, i)
                continue
            assert err[i] == 0 and list(out[:, i]) == wo and int(ov[i]) == wov, (name, i)
    inv, div, err = [np.asarray(a) for a in be.inv(field, x)]
    for i, xl in enumerate(xs):
        try:
            wi, wd = R.gen_inv(xl, m)
        except R.RefPanic:
            assert err[i] != 0, ("inv", i)
            continue
        assert err[i] == 0 and list(inv[:, i]) == wi and list(div[:, i]) == wd, ("inv", i)
    n = len(xs) // 4 * 4
    s4 = np.array([[xs[4 * j + k] for j in range(n // 4)] for k in range(4)], dtype=np.uint64).transpose(0, 2, 1).copy()   # (4, 9, n/4)
    out, ov, err = [np.asarray(a) for a in be.add_many(field, s4)]
    for j in range(n // 4):
        try:
            wo, wov = R.gen_add_many([xs[4 * j + k] for k in range(4)], m)
        except R.RefPanic:
            assert err[j] != 0, ("add_many", j)
            continue
        assert err[j] == 0 and list(out[:, j]) == wo and int(ov[j]) == wov, ("add_many", j)


@pytest.mark.parametrize("field", [R.FIELD_P256_BASE, R.FIELD_P256_SCALAR])
def test_generators_over_the_p256_fields_kernel_bodies(field):
    """MulNonnative + CheckSum, NonNativeAddition / Subtraction / MultipleAdds / Inverse (gates/mul_nonnative.rs:249-324,
    513-531; gadgets/nonnative.rs:626-645, 792-810, 696-728, 857-872) with FF = P256Base / P256Scalar, incl. the
    261-bit raw operands of the multiplication gate (reduce_barrett_wide) and every panic of the reference as a flag"""
    from backends import EmuBackend
    _check_p256_generators(EmuBackend(), field)


@pytest.mark.gpu
@pytest.mark.parametrize("field", [R.FIELD_P256_BASE, R.FIELD_P256_SCALAR])
def test_gpu_generators_over_the_p256_fields(field):
    from backends import GpuBackend
    _check_p256_generators(GpuBackend(), field, count=600)
