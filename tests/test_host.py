"""CPU: host logic of the product -- the C ABI library loads and exports every declared symbol, the
schedule builder reproduces the reference's generator order, the synthetic-input generator restates
sign_message, and the per-lane kernel bodies (compiled for the CPU by tests/emu, test-only) are
bit-exact against the golden fixtures.  No GPU compute is called.  -m "not gpu"."""
import os
import re

import numpy as np
import pytest

import oracle_c
import p2e_ref as R
import parity_checks as pc
import plonky2_ecdsa_amd as p2e
from backends import EmuBackend, OracleBackend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "p2e.h")).read()
    declared = set(re.findall(r"\b(p2e_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"p2e_ctx"}
    assert declared == set(p2e.EXPORTS), declared ^ set(p2e.EXPORTS)
    L = p2e.lib()
    for name in declared:
        assert hasattr(L, name), name


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(p2e.P2EError, match="no HIP device|NO_DEVICE|failed"):
        p2e.Context(device=0, host_pointers=True)


def test_schedule_matches_reference_generator_order():
    assert p2e.schedule_describe(p2e.PROGRAM_VERIFY) == pc.golden_schedule("verify")
    assert p2e.schedule_describe(p2e.PROGRAM_GLV_MUL) == pc.golden_schedule("glv_mul")
    assert p2e.schedule_num_cols(p2e.PROGRAM_VERIFY) == p2e.VERIFY_COLS == 82615
    assert p2e.schedule_num_cols(p2e.PROGRAM_GLV_MUL) == p2e.GLV_MUL_COLS == 65243
    from collections import Counter
    c = Counter(k for k, *_ in p2e.schedule_describe(0))
    # SURVEY.md section 8 op counts per verify
    assert (c["mul"], c["inv"], c["sub"], c["add"], c["add_many"], c["glv"]) == (1087, 312, 1267, 742, 146, 1)


def test_aux_column_map_matches_the_gadget_order():
    """p2e_aux_describe groups the oracle's per-builder-call records into one item per window / split /
    conditional negation: same order, same boundaries, same labels."""
    for prog, name, total in ((p2e.PROGRAM_VERIFY, "verify", p2e.VERIFY_AUX_COLS), (p2e.PROGRAM_GLV_MUL, "glv_mul", p2e.GLV_MUL_AUX_COLS)):
        items = p2e.aux_describe(prog)
        fine = pc.golden_aux_schedule(name)          # (kind, first, n, label) per builder call
        assert p2e.aux_num_cols(prog) == total == sum(n for _k, _f, n, _l in items) == sum(o[2] for o in fine)
        starts = {o[1]: o for o in fine}
        pos = 0
        for kind, first, n, label in items:
            assert first == pos and first in starts, (kind, first)
            assert starts[first][3] == label, (label, starts[first])
            pos += n
        assert pos == total
    from collections import Counter
    c = Counter(k for k, *_ in p2e.aux_describe(0))
    assert (c["split4"], c["split2"], c["fixed_base_window"], c["msm_digit"], c["conditional_neg"]) == (1, 2, 66, 73, 4)


def test_aux_bodies_match_golden():
    pc.check_aux_golden(EmuBackend())


def test_aux_bodies_random_batch_and_limb_range_flag():
    emu, ora = EmuBackend(), OracleBackend()
    sigs = p2e.synth_signatures(seed=808, n=70)
    _c, want, _e = ora.aux(0, sigs)
    cols, got, err = emu.aux(0, sigs)
    assert not err.any() and np.array_equal(got, want)
    # a scalar limb >= 2^29 has no split_le_base witness: flagged, nothing else disturbed
    cols = cols.copy()
    u1_col = [g for g in p2e.schedule_describe(0) if g[0] == "mul"][4][2]      # after the 4 muls of curve_assert_valid
    cols[u1_col + 3, 7] |= np.uint64(1 << 29)
    aux = np.zeros_like(got)
    aerr = np.zeros(70, np.uint8)
    import ctypes as C
    emu.L.emu_aux(C.c_int(0), sigs[4].ctypes.data_as(C.c_void_p), cols.ctypes.data_as(C.c_void_p), C.c_size_t(70),
                  aux.ctypes.data_as(C.c_void_p), C.c_size_t(70), C.c_size_t(70), aerr.ctypes.data_as(C.c_void_p))
    assert aerr[7] == 1 and not np.delete(aerr, 7).any()
    assert np.array_equal(np.delete(aux, 7, axis=1), np.delete(want, 7, axis=1))


def test_compact_layout_is_a_partition_of_the_columns():
    """Narrow = every column but check_sum[17] and b[16] of the mul generators; each matrix index used exactly once."""
    for prog, muls in ((p2e.PROGRAM_VERIFY, 1087), (p2e.PROGRAM_GLV_MUL, 877)):
        m, nn, nw = p2e.compact_layout(prog)
        wide = (m & p2e.COMPACT_WIDE) != 0
        assert nw == 33 * muls == int(wide.sum()) and nn == len(m) - nw
        assert sorted((m[wide] & 0x7FFFFFFF).tolist()) == list(range(nw)) and sorted(m[~wide].tolist()) == list(range(nn))
        for kind, _field, c0, nc, _label in p2e.schedule_describe(prog):
            assert wide[c0:c0 + nc].tolist() == ([False] * 18 + [True] * 33 if kind == "mul" else [False] * nc)
    assert p2e.compact_layout(0)[1:] == (46744, 35871)
    # the goldens really fit: every narrow column of the golden witnesses is < 2^32
    cols, _inputs, _valid = pc.load_verify_golden()
    m, _, _ = p2e.compact_layout(0)
    assert int((cols[(m & p2e.COMPACT_WIDE) == 0] >> np.uint64(32)).max()) == 0


@pytest.mark.parametrize("program", [0, 1])
def test_compact_emitter_bodies_match_the_oracle(program):
    """The fused walk with the compact emitter (u32 narrow / u64 wide cursors) expands back to the oracle's matrix."""
    emu, ora = EmuBackend(), OracleBackend()
    sigs = p2e.synth_signatures(seed=909, n=37)
    if program == 0:
        inputs, want = sigs, ora.verify(*sigs)[0]
    else:
        rng = R.SplitMix64(910)
        inputs = [sigs[3], sigs[4], oracle_c.pack256([rng.below(R.N) for _ in range(37)])]
        want = ora.glv_mul(*inputs)[0]
    _m, nn, nw = p2e.compact_layout(program)
    narrow, wide, err, _valid = emu.compact(program, inputs, nn, nw)
    assert not err.any()
    assert np.array_equal(p2e.compact_expand(program, narrow, wide), want)
    # the built-in-generator pass inside the container: narrow matrix in, u32 matrix out
    aux32, aerr = emu.aux_compact(program, inputs[4] if program == 0 else inputs[1], narrow)
    assert not aerr.any() and np.array_equal(aux32.astype(np.uint64), ora.aux(program, inputs)[1])


def _build_c_example(tmp_path, name="fill_batch"):
    import subprocess
    exe = str(tmp_path / name)
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", name + ".c"), "-L", os.path.join(ROOT, "plonky2-ecdsa_amd"),
                           "-lp2e_hip", "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe])
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "plonky2-ecdsa_amd") + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""),
               GPU_MAX_HW_QUEUES="8")
    return exe, env


def test_plain_c_client_builds_against_the_header_and_fails_loudly_without_a_gpu(tmp_path):
    """include/p2e.h is plain C; a C client links against the C ABI; with no GPU it gets an error, not a CPU path."""
    import subprocess
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c",
                           os.path.join(ROOT, "include", "p2e.h")])
    exe, env = _build_c_example(tmp_path)
    exe2, _ = _build_c_example(tmp_path, "fill_p256")
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: covered by the GPU suite")
    for e in (exe, exe2):
        r = subprocess.run([e, "8"], env=env, capture_output=True, text=True)
        assert r.returncode == 2 and "no CPU fallback" in r.stderr


def test_div_rem_bodies_random_shapes():
    """BigUintDivRemGenerator body (restoring division in registers) against the C oracle (Knuth D) on random operands of
    every supported shape, quotients pushed into the zone where the reference's set_biguint_target quirk bites."""
    import random
    emu, ora = EmuBackend(), OracleBackend()
    rnd = random.Random(4242)
    split = lambda v, k: [(v >> (29 * i)) & ((1 << 29) - 1) for i in range(k)]
    flagged = 0
    for na, nb in ((18, 9), (9, 9), (10, 9), (17, 8), (18, 1), (5, 9), (9, 5), (3, 3), (1, 1), (12, 7), (2, 9)):
        nd = 0 if nb > na + 1 else na - nb + 1
        A, Bv = [], []
        for t in range(120):
            a, b = rnd.getrandbits(rnd.randint(0, 29 * na)), rnd.getrandbits(rnd.randint(1, 29 * nb))
            if t % 3 == 0 and b and nd:
                a = min((1 << (29 * na)) - 1, b * rnd.getrandbits(max(1, 29 * nd - rnd.randint(0, 4))) + rnd.randrange(b))
            if t == 1:
                b = 0
            A.append(split(a, na)), Bv.append(split(b, nb))
        a, b = np.array(A, dtype=np.uint64).T.copy(), np.array(Bv, dtype=np.uint64).T.copy()
        want, got = ora.div_rem(a, b), emu.div_rem(a, b)
        assert np.array_equal(got[2], want[2]), (na, nb)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), (na, nb)
        flagged += int((want[2] != 0).sum())
    assert flagged > 30      # division by zero and the quirk zone are really exercised


def test_synth_signatures_restates_sign_message():
    arrs = p2e.synth_signatures(seed=9, n=4, first=2)
    for i in range(4):
        want = R.synth_signature_at(9, 2 + i)
        got = tuple(oracle_c.unpack256(a[i:i + 1])[0] for a in arrs)
        assert got == want
    # and they verify natively (curve/ecdsa.rs:42-62)
    msg, r, s, px, py = R.synth_signature_at(9, 2)
    c = pow(s, -1, R.N)
    pt = R.ec_add(R.ec_mul(msg * c % R.N, R.G), R.ec_mul(r * c % R.N, (px, py)))
    assert pt[0] % R.N == r


@pytest.mark.parametrize("check", pc.ALL_PRIM_CHECKS, ids=lambda f: f.__name__)
def test_kernel_bodies_match_golden(check):
    check(EmuBackend())


def test_pipeline_bodies_match_golden_verify():
    pc.check_verify_golden(EmuBackend())


def test_pipeline_bodies_match_golden_glv_mul():
    pc.check_glv_mul_golden(EmuBackend())


@pytest.mark.parametrize("chunk", [1, 7, 311])
def test_batch_inversion_chunking_is_output_invariant(chunk):
    emu, ora = EmuBackend(), OracleBackend()
    arrs = p2e.synth_signatures(seed=21, n=3)
    want, _, _ = ora.verify(*arrs)
    got, err, valid = emu.verify(*arrs, chunk=chunk)
    assert not err.any() and valid.all() and np.array_equal(got, want)


@pytest.mark.parametrize("field", [0, 1, 2, 3], ids=["secp256k1_p", "secp256k1_n", "p256_p", "p256_n"])
def test_safegcd_inversion_stress(field):
    """csrc/fe.hpp fe_inv_safegcd (Bernstein-Yang division steps on signed 30-bit limbs, what fe_inv runs): x * inv(x) == 1,
    the result canonical and identical to the binary-GCD inversion's, zero reported as not invertible -- random, short,
    word-patterned and m - small inputs over the crate's four moduli (field/*.rs: inverse = x^(m-2))."""
    import ctypes as C
    L = EmuBackend().L
    L.emu_safegcd_selfcheck.restype = C.c_long
    assert L.emu_safegcd_selfcheck(C.c_int(field), C.c_ulonglong(31 + field), C.c_size_t(500000)) == 0


def test_lazy_limb_curve_formulas_match_the_canonical_ones_in_every_variant():
    """csrc/ec29.hpp, quad29.hpp against csrc/ec.hpp, quad.hpp: the doubling and all four Z-one variants of the addition, lane per
    signature and four-lane form (every role, with and without the carried Z1^2, with and without the affine slot), on
    arbitrary coordinates incl. equal operands (Z3 = 0): X3, Y3, Z3, W, the prefix product, Z^2 values, the value each lane
    stores and the zero flag.  Runs under the limb-bound tracking of the emulation build."""
    import ctypes as C
    L = EmuBackend().L
    L.emu_ec29_selftest.restype = C.c_long
    assert L.emu_ec29_selftest(C.c_ulonglong(13), C.c_size_t(30000)) == 0


def test_lazy_limb_constants_are_what_the_header_says():
    """tools/f29_constants.py: 2^261 mod p and the borrowed-limb multiples of p that csrc/fe29.hpp subtracts with, re-derived."""
    import runpy
    runpy.run_path(os.path.join(ROOT, "tools", "f29_constants.py"), run_name="__main__")


def test_lazy_limb_field_arithmetic_matches_the_canonical_form():
    """csrc/fe29.hpp (secp256k1's p on unsaturated 29-bit limbs, what the four-lane chains compute with) against csrc/fe.hpp
    (canonical 32-bit words): every operation, operands pushed through the lazy forms the chains use, on random and
    patterned values (0, 1, p - 1, p - small, all-ones words).  The emulation build carries a worst-case bound beside
    every limb (-DP2E_F29_BOUNDS) and aborts the process if any operation's precondition could be violated."""
    import ctypes as C
    L = EmuBackend().L
    L.emu_f29_selftest.restype = C.c_long
    assert L.emu_f29_selftest(C.c_ulonglong(7), C.c_size_t(300000)) == 0


@pytest.mark.parametrize("split", [1, 4, 5])
def test_quad_lanes_and_split_inversion_batches_are_output_invariant(split):
    """csrc/quad.hpp (the small-batch plan): the chains walked by four lanes per signature (levels of four
    multiplications, host form), the window table as rows, every inversion batch cut into `split` sub-ranges --
    same columns, error flags and verdicts as the oracle, for both programs, with an inverse-of-zero element."""
    emu, ora = EmuBackend(), OracleBackend()
    arrs = [a.copy() for a in p2e.synth_signatures(seed=19, n=9)]
    arrs[1][3, 0] ^= 1
    rx, ry = R.rando_point()
    arrs[3][5] = np.frombuffer(int(rx).to_bytes(32, "little"), dtype=np.uint8)
    arrs[4][5] = np.frombuffer(int((-ry) % R.P).to_bytes(32, "little"), dtype=np.uint8)
    want, werr, wflags = ora.verify(*arrs)
    got, err, valid = emu.verify(*arrs, chunk=-split, run_iters=0)
    ok = werr == 0
    assert np.array_equal(err != 0, werr != 0) and err[5] & R.ERR_INVERSE_OF_ZERO
    assert np.array_equal(got[:, ok], want[:, ok]) and np.array_equal(valid[ok], wflags[ok])
    rng = R.SplitMix64(20)
    k = oracle_c.pack256([rng.below(R.N) for _ in range(9)])
    want, werr, _ = ora.glv_mul(arrs[3], arrs[4], k)
    got, err, _ = emu.glv_mul(arrs[3], arrs[4], k, chunk=-split, run_iters=4)
    ok = werr == 0
    assert np.array_equal(err != 0, werr != 0) and np.array_equal(got[:, ok], want[:, ok])


@pytest.mark.parametrize("run_iters", [0, 1, 3, 73])
def test_expansion_run_length_is_output_invariant(run_iters):
    """Phase C walks MSM-loop iterations in runs (phase B skips the affine conversion inside a run): any run
    length, and op-by-op expansion (0), must give the same columns."""
    emu, ora = EmuBackend(), OracleBackend()
    arrs = p2e.synth_signatures(seed=22, n=3)
    want, _, _ = ora.verify(*arrs)
    got, err, valid = emu.verify(*arrs, run_iters=run_iters)
    assert not err.any() and valid.all() and np.array_equal(got, want)


def test_pipeline_edge_inputs():
    """Inputs the reference accepts but a naive implementation gets wrong: non-canonical coordinates
    (>= p, < 2^256), a signature that does not verify, s with small inverse, pk = G."""
    emu, ora = EmuBackend(), OracleBackend()
    sig = list(R.synth_signature_at(5, 0))
    cases = [tuple(sig)]
    # pk.x + p if it still fits 256 bits is rare; use msg + n (non-canonical scalar) when it fits
    m2 = sig[0] + R.N
    if m2 < 2**256:
        cases.append((m2,) + tuple(sig[1:]))
    bad = list(sig)
    bad[1] = (bad[1] + 1) % R.N
    cases.append(tuple(bad))                                     # wrong r: constraints fail, witness defined
    cases.append((sig[0], sig[1], 1, sig[3], sig[4]))            # s = 1
    cases.append((sig[0], sig[1], sig[2], R.GX, R.GY))           # pk = G (table p == G multiples)
    arrs = [oracle_c.pack256([c[k] for c in cases]) for k in range(5)]
    want, werr, wflags = ora.verify(*arrs)
    got, err, valid = emu.verify(*arrs)
    assert np.array_equal(err != 0, werr != 0)
    ok = werr == 0
    assert np.array_equal(got[:, ok], want[:, ok]) and np.array_equal(valid[ok], wflags[ok])
    # the verdict-only walk (p2e_ecdsa_verify_batch bodies: no emission, no batch inversion) agrees on err and valid
    verr, vvalid = emu.verify_only(*arrs)
    assert np.array_equal(verr != 0, err != 0) and np.array_equal(vvalid, valid)


def test_verdict_only_bodies_on_mixed_batch():
    """Valid signatures, tampered r / s / msg / pk, r >= p, and an inverse-of-zero element: the verdict-only walk sets
    valid and err like the full witness walk (which the oracle checks)."""
    emu, ora = EmuBackend(), OracleBackend()
    n = 48
    sigs = [list(R.synth_signature_at(77, i)) for i in range(n)]
    rx, ry = R.rando_point()
    for i in range(0, n, 6):
        sigs[i][i % 3] = (sigs[i][i % 3] + 1 + i) % R.N           # tamper msg / r / s
    sigs[7][3], sigs[7][4] = R.GX, R.GY                            # wrong public key
    sigs[11][1] = R.P + 5                                          # r >= p can never equal a canonical x
    sigs[13][3], sigs[13][4] = rx, (-ry) % R.P                     # inverse of zero in the window table
    arrs = [oracle_c.pack256([c[k] for c in sigs]) for k in range(5)]
    _want, werr, wflags = ora.verify(*arrs)
    verr, vvalid = emu.verify_only(*arrs)
    assert np.array_equal(verr != 0, werr != 0) and verr[13] & R.ERR_INVERSE_OF_ZERO
    ok = werr == 0
    assert np.array_equal(vvalid[ok], wflags[ok]) and not vvalid[~ok].any()
    assert vvalid.sum() == n - len(range(0, n, 6)) - 3


def test_inverse_of_zero_is_flagged_not_fatal():
    """pk = -rando makes the very first table add hit x2 == x1 (reference: inverse() of zero panics,
    gadgets/nonnative.rs:863): the element is flagged, the rest of the batch is untouched."""
    emu, ora = EmuBackend(), OracleBackend()
    sig = R.synth_signature_at(6, 0)
    rx, ry = R.rando_point()
    cases = [sig, (sig[0], sig[1], sig[2], rx, (-ry) % R.P), R.synth_signature_at(6, 1)]
    arrs = [oracle_c.pack256([c[k] for c in cases]) for k in range(5)]
    want, werr, _ = ora.verify(*arrs)
    got, err, valid = emu.verify(*arrs)
    assert werr[1] & R.ERR_INVERSE_OF_ZERO and err[1] & R.ERR_INVERSE_OF_ZERO
    assert not err[0] and not err[2] and valid[0] and valid[2] and not valid[1]
    assert np.array_equal(got[:, [0, 2]], want[:, [0, 2]])


def test_kernel_bodies_structured_operands():
    """Carry chains / fold reductions on operands with long runs of ones and zeros (incl. values >= m)."""
    pc.check_structured_mul_add_sub(EmuBackend(), OracleBackend(), count=3000)


def test_rare_reduction_branches_are_in_the_goldens():
    """The KATs must contain multiplications whose product is congruent to something tiny: only those take
    the second-carry / final-correction paths of the fold reduction (csrc/fe.hpp reduce_p16, reduce_wide)."""
    hits = {0: set(), 1: set()}
    for k in pc.KATS["mul"]:
        if k["err"]:
            continue
        m = R.MODULI[k["field"]]
        x, y = R.value_of(k["x"]), R.value_of(k["y"])
        if x >> 256 or y >> 256:
            continue
        c = 2**256 - m
        hi, lo = (x * y) >> 256, (x * y) & (2**256 - 1)
        path = []
        for _ in range(4):
            t = lo + hi * c
            hi, lo = t >> 256, t & (2**256 - 1)
            path.append(min(hi, 1))
        hits[k["field"]].add((tuple(path[1:]), lo >= m))
    assert ((1, 0, 0), False) in hits[0] and any(h[1] for h in hits[0])          # p: second carry, final r >= p
    assert any(h[0][:2] == (1, 1) for h in hits[1]) and any(h[1] for h in hits[1])  # n: third carry, final r >= n


def test_binary_gcd_inversion_round_bound():
    """csrc/fe.hpp fe_inv_bingcd: 17 rounds of 31 approximate steps must end in gcd = 1 for every input
    (self-checking loop x * inv(x) == 1 over assorted bit lengths and word patterns; 10^8 inputs were run
    once offline, 4 * 10^5 here)."""
    import ctypes as C
    L = EmuBackend().L
    L.emu_bingcd_selfcheck.restype = C.c_long
    for field in (0, 1):
        assert L.emu_bingcd_selfcheck(C.c_int(field), C.c_ulonglong(99 + field), C.c_size_t(200000)) == 0


def _ux_digest(v):
    import hashlib
    return hashlib.sha256(np.asarray(v, dtype="<u4").tobytes()).hexdigest()


@pytest.mark.parametrize("program", [0, 1])
def test_constraint_block_columns_match_the_replay_model_and_the_golden_digests(program):
    """SURVEY 8(f) rank 2 (csrc/ux.hpp, the body k_ux runs, compiled for the CPU): the values the U29 gates hand back
    inside every add / sub / add_many / inv constraint block and range check, derived from the golden witness and aux
    matrices through the schedule's wiring table, against oracle/check_circuit.py's model -- which produces them as a
    by-product of replaying the reference's constraints on the same witness -- and against the committed digests."""
    import check_circuit as CC
    import json
    dig = json.load(open(os.path.join(pc.GOLD, "ux_digest.json")))["verify" if program == 0 else "glv_mul"]
    if program == 0:
        cols, inputs, valid = pc.load_verify_golden()
        aux = np.load(os.path.join(pc.GOLD, "aux_golden.npz"))["verify"]
        args = [np.ascontiguousarray(inputs[:, k, :]) for k in range(5)]
    else:
        g = np.load(os.path.join(pc.GOLD, "glv_mul_golden.npz"))
        cols, inputs, valid = g["cols"], g["inputs"], np.ones(g["cols"].shape[1], np.uint8)
        aux = np.load(os.path.join(pc.GOLD, "aux_golden.npz"))["glv_mul"]
        args = [np.ascontiguousarray(inputs[:, k, :]) for k in range(3)]
    ux, err = EmuBackend().ux(program, args, cols, aux)
    assert not err.any() and ux.shape[0] == p2e.ux_num_cols(program) == (p2e.VERIFY_UX_COLS if program == 0 else p2e.GLV_MUL_UX_COLS)
    assert int(ux.max()) < 1 << 29
    for i in range(cols.shape[1]):
        if not valid[i]:
            continue
        ins = CC.unpack_inputs([inputs[:, k, :] for k in range(inputs.shape[1])], i)
        c = (CC.check_verify if program == 0 else CC.check_glv_mul)(cols[:, i], *ins)
        assert np.array_equal(ux[:, i], np.array(c.ux, dtype=np.uint64))
        assert _ux_digest(c.ux) == dig[i]["sha256"] and len(c.ux) == dig[i]["len"]
        # the library's per-generator block map is the replay's
        blocks = [(u0, un) for (u0, un, _l) in c.ux_ops]
        lib = [b for (b, g) in zip(p2e.ux_describe(program), c.gens) if g[0] != "glv"]
        assert lib == blocks
    # feeding the derived vector back: the replay accepts it, and rejects a corrupted one
    c = (CC.check_verify if program == 0 else CC.check_glv_mul)(cols[:, 0], *CC.unpack_inputs([inputs[:, k, :] for k in range(inputs.shape[1])], 0),
                                                                 ux=ux[:, 0])
    bad = ux[:, 0].copy()
    bad[12345] ^= np.uint64(1)
    with pytest.raises(CC.ConstraintViolation):
        (CC.check_verify if program == 0 else CC.check_glv_mul)(cols[:, 0], *CC.unpack_inputs([inputs[:, k, :] for k in range(inputs.shape[1])], 0), ux=bad)


@pytest.mark.parametrize("program", [0, 1])
def test_gate_internal_values_match_the_replay_model(program):
    """The rest of SURVEY 8(f) rank 1 (csrc/aux.hpp body_gate, compiled for the CPU): per window the equality gadget's
    internals and the RandomAccessGate index bits, derived from the aux matrix, against the values the constraint replay
    records where the gadgets call is_equal / random_access ([upstream-from-memory] semantics, "parity unpinned")."""
    import check_circuit as CC
    if program == 0:
        cols, inputs, valid = pc.load_verify_golden()
        aux = np.load(os.path.join(pc.GOLD, "aux_golden.npz"))["verify"]
    else:
        g = np.load(os.path.join(pc.GOLD, "glv_mul_golden.npz"))
        cols, inputs, valid = g["cols"], g["inputs"], np.ones(g["cols"].shape[1], np.uint8)
        aux = np.load(os.path.join(pc.GOLD, "aux_golden.npz"))["glv_mul"]
    gate = EmuBackend().gate(program, aux)
    assert gate.shape[0] == (p2e.VERIFY_GATE_COLS if program == 0 else p2e.GLV_MUL_GATE_COLS)
    for i in range(cols.shape[1]):
        if not valid[i]:
            continue
        ins = CC.unpack_inputs([inputs[:, k, :] for k in range(inputs.shape[1])], i)
        c = (CC.check_verify if program == 0 else CC.check_glv_mul)(cols[:, i], *ins)
        assert np.array_equal(gate[:, i], np.array(c.gate, dtype=np.uint64))
    # the equality internals satisfy the gadget's own relations in Goldilocks: diff * inv == not_equal, diff * not_equal == diff
    # (fixed-base windows call is_equal before the 18 random accesses, MSM digits after: gadgets/curve_fixed_base.rs:57-60,
    # gadgets/curve_msm.rs:68-70)
    blocks = gate[:, 0].reshape(-1, 77)
    n_fb = 66 if program == 0 else 0
    for w, blk in enumerate(blocks):
        off = 0 if w < n_fb else 72
        ne, inv, diff, chk, dn = [int(v) for v in blk[off:off + 5]]
        assert ne in (0, 1) and chk == ne and dn == diff and diff < 16 and (diff * inv - ne) % R.P_GL == 0
        bits = blk[5:] if w < n_fb else blk[:72]
        assert all(sum(int(b) << k for k, b in enumerate(bits[4 * j:4 * j + 4])) == diff for j in range(18))
