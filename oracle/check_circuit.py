"""Constraint replay of verify_secp256k1_message_circuit on a finished witness (TEST INFRASTRUCTURE ONLY).

``oracle/p2e_ref.py`` restates the reference's GENERATORS (what a witness value is); this file restates the
reference's CONSTRAINTS (what a plonky2 prove + verify of the reference's own tests would enforce on that
witness) and replays them, with the operand wiring of the gadgets, on a column vector that came from anywhere:
the committed goldens, the C oracle, or the HIP kernels.  No generator of p2e_ref is called: every one of
the 3 555 hot-path generators' outputs is READ from the witness at its registration-order column and pushed
through the equations of the gadget that registered it.  A witness with any single corrupted column fails
(``tests/test_check_circuit.py`` mutates at least one column of every generator kind).

Since the reference holds no golden vectors for these generators and cannot be built here,
this replay is what ties the numbers to the reference: parity stays "unpinned" at the level of literal
reference outputs, but every value the product emits is shown to satisfy the reference's own circuit.

Restated from (paths relative to /root/reference/src):
  gadgets/nonnative.rs:245-276   add_nonnative          :310-353  add_many_nonnative
  gadgets/nonnative.rs:356-388   sub_nonnative          :390-464  mul_nonnative (two gates + 7 range checks)
  gadgets/nonnative.rs:502-536   inv_nonnative          :584-596  nonnative_conditional_neg
  gadgets/nonnative.rs:180-196   biguint_to_nonnative (range_check = cmp_biguint(x, modulus) == 1)
  gates/mul_nonnative.rs:101-130 MulNonnativeGate::eval_unfiltered      :411-427 CheckSumGate::eval_unfiltered
  gadgets/biguint.rs:181-417     connect / pad / cmp / add / sub / mul / mul_by_bool (literal restatement)
  gadgets/curve.rs:123-243       curve_assert_valid / double / add / conditional_add / conditional_neg
  gadgets/curve_windowed_mul.rs:74-118  random_access_curve_points
  gadgets/split_nonnative.rs:25-72      4-bit / 2-bit digits
  gadgets/curve_fixed_base.rs:18-66, gadgets/curve_msm.rs:21-79, gadgets/glv.rs:53-104, gadgets/ecdsa.rs:30-53

[upstream-from-memory] plonky2_ux (Weobe/plonky2-ux, not in the container) is modelled by what its gates
constrain, which is all the call sites in gadgets/biguint.rs rely on:
  add_many_ux(xs)             -> (sum mod 2^29, sum >> 29)
  add_uxs_with_carry(xs, c)   -> the same with the carry-in added
  mul_ux(a, b)                -> (a*b mod 2^29, a*b >> 29)
  sub_ux(x, y, borrow)        -> (x - y - borrow + 2^29 * borrow_out, borrow_out), borrow_out in {0, 1}
  range_check_ux_circuit(v, bits): every v < 2^bits;  list_le_ux_circuit(a, b) -> [a <= b] (little-endian limbs)
Operands of these gates are asserted to be proper limbs (< 2^29; carries < 2^29): true of every honest witness,
and stricter than the gates themselves, which only makes the replay reject more.

The values those calls RETURN (limb, carry / borrow), together with the mul_biguint_by_bool products and the
cmp_biguint results of the same constraint blocks, are recorded in builder-call order in ``Circuit.ux``: this is
the oracle for the SURVEY.md 8(f) rank 2 matrix (include/p2e.h p2e_ux_witness_batch), "parity unpinned" in the
sense above.  The built-in-generator values (8(f) rank 1: bits, digits, is_equal / not, random-access
selections, bool products) are derived here as well and compared with an ``aux`` vector when one is given.
"""
from __future__ import annotations

import p2e_ref as R
from p2e_ref import BITS, NL, P_GL, FIELD_BASE, FIELD_SCALAR, MODULI, const_limbs

B29 = 1 << BITS
CARRY_OVER_BASE = 34          # gadgets/nonnative.rs:453


class ConstraintViolation(AssertionError):
    def __init__(self, where, what):
        super().__init__(f"{where}: {what}")
        self.where, self.what = where, what


# Where a target's limb comes from: ("col", c) hot-path witness column, ("aux", c) built-in-generator column,
# ("ux", c) constraint-block column, ("in", name, limb) caller input, ("const", value).  Only recorded, never
# used by the checks: it is the wiring table the f2 / f3 kernels are tested against.
class T:
    """A BigUintTarget / NonNativeTarget: limb values + the place each limb lives."""
    __slots__ = ("v", "s")

    def __init__(self, v, s=None):
        self.v = [int(x) for x in v]
        self.s = list(s) if s is not None else [("const", x) for x in self.v]
        assert len(self.v) == len(self.s)

    def __len__(self):
        return len(self.v)

    def value(self):
        return R.value_of(self.v)


def const_t(value):
    """constant_biguint gadgets/biguint.rs:165-175: variable limb count (0 -> no limbs)."""
    return T(const_limbs(value))


def input_t(name, value):
    """a caller-set 9-limb virtual target (pw.set_biguint_target, gadgets/biguint.rs:454-463)."""
    l = R.limbs_of(value, NL)
    return T(l, [("in", name, k) for k in range(NL)])


class Circuit:
    def __init__(self, cols, aux=None, ux=None, record=True, curve=None):
        self.curve = curve or R.SECP256K1
        self.cols = cols
        self.cur = 0                  # registration-order cursor into cols
        self.aux_in = aux
        self.ux_in = ux
        self.aux = []                 # derived built-in-generator values (SURVEY 8(f) rank 1 order)
        self.ux = []                  # constraint-block values (8(f) rank 2), builder-call order
        self.gate = []                # gate-internal values of the built-in gates (rest of 8(f) rank 1), call order
        self.record = record
        self.path = []
        self.gens = []                # (kind, field, first_col, ncols, label, operand sources)
        self.ux_ops = []              # (first_ux_col, ncols, label) per generator's constraint block
        self.counts = {}

    # ---- bookkeeping -------------------------------------------------------------------------------------
    class _Scope:
        def __init__(self, c, name):
            self.c, self.name = c, name

        def __enter__(self):
            self.c.path.append(self.name)

        def __exit__(self, *a):
            self.c.path.pop()

    def scope(self, name):
        return Circuit._Scope(self, name)

    def where(self, what=""):
        return "/".join(self.path) + (":" + what if what else "")

    def fail(self, what, detail):
        raise ConstraintViolation(self.where(what), detail)

    def require(self, cond, what, detail=""):
        if not cond:
            self.fail(what, detail or "constraint violated")

    def read(self, kind, field, n, operands=()):
        """the next generator's n output columns"""
        if self.cur + n > len(self.cols):
            self.fail(kind, "witness too short")
        c0 = self.cur
        vals = [int(v) for v in self.cols[c0:c0 + n]]
        for v in vals:
            if not 0 <= v < P_GL:
                self.fail(kind, f"column value {v} is not a canonical Goldilocks element")
        self.cur += n
        self.gens.append((kind, field, c0, n, "/".join(self.path), tuple(tuple(o.s) for o in operands)))
        self.counts[kind] = self.counts.get(kind, 0) + 1
        return c0, vals

    def _aux(self, vals):
        """values of targets plonky2's own generators fill; returns their aux column sources"""
        c0 = len(self.aux)
        vals = [int(v) for v in vals]
        if self.aux_in is not None:
            got = [int(v) for v in self.aux_in[c0:c0 + len(vals)]]
            if got != vals:
                k = next(i for i, (a, b) in enumerate(zip(got + [None] * len(vals), vals)) if a != b)
                self.fail("aux", f"built-in-generator column {c0 + k}: witness has {got[k] if k < len(got) else None}, "
                                 f"the gadget's constraints force {vals[k]}")
        self.aux.extend(vals)
        return [("aux", c0 + k) for k in range(len(vals))]

    def _ux(self, vals):
        c0 = len(self.ux)
        vals = [int(v) for v in vals]
        if self.ux_in is not None:
            got = [int(v) for v in self.ux_in[c0:c0 + len(vals)]]
            if got != vals:
                k = next(i for i, (a, b) in enumerate(zip(got + [None] * len(vals), vals)) if a != b)
                self.fail("ux", f"constraint-block column {c0 + k}: witness has {got[k] if k < len(got) else None}, "
                                f"the U29 gates force {vals[k]}")
        self.ux.extend(vals)
        return [("ux", c0 + k) for k in range(len(vals))]

    class _UxBlock:
        def __init__(self, c):
            self.c = c

        def __enter__(self):
            self.c0 = len(self.c.ux)

        def __exit__(self, et, ev, tb):
            if et is None:
                self.c.ux_ops.append((self.c0, len(self.c.ux) - self.c0, "/".join(self.c.path)))

    # ---- plonky2_ux model [upstream-from-memory] ------------------------------------------------------------
    def _limb_in(self, v, what):
        self.require(0 <= v < B29, what, f"U29 gate operand {v} is not < 2^29")

    def add_many_ux(self, xs):
        s = 0
        for x in xs:
            self._limb_in(x.v[0], "add_many_ux")
            s += x.v[0]
        lo, hi = s % B29, s >> BITS
        src = self._ux([lo, hi])
        return T([lo], src[:1]), T([hi], src[1:])

    def add_uxs_with_carry(self, xs, carry):
        return self.add_many_ux(list(xs) + [carry])

    def mul_ux(self, a, b):
        self._limb_in(a.v[0], "mul_ux")
        self._limb_in(b.v[0], "mul_ux")
        p = a.v[0] * b.v[0]
        src = self._ux([p % B29, p >> BITS])
        return T([p % B29], src[:1]), T([p >> BITS], src[1:])

    def sub_ux(self, x, y, borrow):
        for t in (x, y, borrow):
            self._limb_in(t.v[0], "sub_ux")
        d = x.v[0] - y.v[0] - borrow.v[0]
        bo = 1 if d < 0 else 0
        src = self._ux([d + bo * B29, bo])
        return T([d + bo * B29], src[:1]), T([bo], src[1:])

    def range_check_ux(self, t, bits, what):
        for v in t.v:
            self.require(0 <= v < (1 << bits), what, f"range check: {v} is not < 2^{bits}")

    def zero_ux(self):
        return T([0])

    # ---- gadgets/biguint.rs (literal) ---------------------------------------------------------------------
    @staticmethod
    def limb(t, i):
        return T(t.v[i:i + 1], t.s[i:i + 1])

    def connect_biguint(self, lhs, rhs, what):                  # :181-196
        m = min(len(lhs), len(rhs))
        for i in range(m):
            self.require(lhs.v[i] == rhs.v[i], what, f"connect_biguint limb {i}: {lhs.v[i]} != {rhs.v[i]}")
        for i in range(m, len(lhs)):
            self.require(lhs.v[i] == 0, what, f"connect_biguint: extra lhs limb {i} = {lhs.v[i]} is not zero")
        for i in range(m, len(rhs)):
            self.require(rhs.v[i] == 0, what, f"connect_biguint: extra rhs limb {i} = {rhs.v[i]} is not zero")

    def pad_biguints(self, a, b):                               # :198-219
        n = max(len(a), len(b))
        pa = T(a.v + [0] * (n - len(a)), a.s + [("const", 0)] * (n - len(a)))
        pb = T(b.v + [0] * (n - len(b)), b.s + [("const", 0)] * (n - len(b)))
        return pa, pb

    def cmp_biguint(self, a, b, what):                          # :221-230 list_le_ux_circuit(a, b) = [a <= b]
        a, b = self.pad_biguints(a, b)
        self.range_check_ux(a, BITS, what + ":cmp")
        self.range_check_ux(b, BITS, what + ":cmp")
        le = int(a.value() <= b.value())
        self._ux([le])
        return le

    def add_biguint(self, a, b):                                # :240-270
        n = max(len(a), len(b))
        out_v, out_s = [], []
        carry = self.zero_ux()
        for i in range(n):
            al = self.limb(a, i) if i < len(a) else self.zero_ux()
            bl = self.limb(b, i) if i < len(b) else self.zero_ux()
            lo, carry = self.add_many_ux([carry, al, bl])
            out_v += lo.v
            out_s += lo.s
        return T(out_v + carry.v, out_s + carry.s)

    def sub_biguint(self, a, b):                                # :272-293 (the final borrow is NOT constrained)
        a, b = self.pad_biguints(a, b)
        out_v, out_s = [], []
        borrow = self.zero_ux()
        for i in range(len(a)):
            r, borrow = self.sub_ux(self.limb(a, i), self.limb(b, i), borrow)
            out_v += r.v
            out_s += r.s
        return T(out_v, out_s)

    def mul_biguint(self, a, b):                                # :295-323
        total = len(a) + len(b)
        to_add = [[] for _ in range(total)]
        for i in range(len(a)):
            for j in range(len(b)):
                p, c = self.mul_ux(self.limb(a, i), self.limb(b, j))
                to_add[i + j].append(p)
                to_add[i + j + 1].append(c)
        out_v, out_s = [], []
        carry = self.zero_ux()
        for summands in to_add:
            lo, carry = self.add_uxs_with_carry(summands, carry)
            out_v += lo.v
            out_s += lo.s
        return T(out_v + carry.v, out_s + carry.s)

    def mul_biguint_by_bool_ux(self, a, b):                     # :360-374 inside a constraint block (modulus * overflow)
        v = [l * b % P_GL for l in a.v]
        return T(v, self._ux(v))

    def mul_biguint_by_bool(self, a, b):                        # :360-374 at gadget level (8(f) rank 1 values)
        v = [l * b % P_GL for l in a.v]
        return T(v, self._aux(v))

    # ---- plonky2 built-ins on the path (values only; their gates live upstream) ----------------------------
    def not_(self, b):
        v = (1 - b) % P_GL
        self._aux([v])
        return v

    def is_equal_zero(self, x):
        """builder.is_equal(x, zero) [upstream-from-memory, plonky2 gadgets/arithmetic.rs]: EqualityGenerator sets
        equal = (x == y) and inv = (x - y)^-1 (0 if equal); the gadget then builds not_equal = not(equal),
        diff = x - y, not_equal_check = diff * inv, diff_normalized = diff * not_equal_check and connects
        not_equal == not_equal_check, diff == diff_normalized.  The returned bool is an aux value; the five internal
        targets go to the gate-internal vector."""
        v = int(x == 0)
        self._aux([v])
        inv = pow(x, -1, P_GL) if x % P_GL else 0
        diff = x % P_GL
        check = diff * inv % P_GL
        self.require(check == 1 - v, "is_equal", "not_equal != diff * inv")
        self.require(diff * check % P_GL == diff, "is_equal", "diff != diff * not_equal_check")
        self.gate.extend([1 - v, inv, diff, check, diff * check % P_GL])
        return v

    # ---- gadgets/nonnative.rs --------------------------------------------------------------------------------
    def modulus(self, field):
        return const_t(self.curve.moduli[field])

    def range_check_result(self, x, field, what):               # :180-190: cmp_biguint(x, modulus) connected to one
        self.require(self.cmp_biguint(x, self.modulus(field), what) == 1, what, "range check: value > modulus")

    def add_nonnative(self, a, b, field, range_check=False):    # :245-276
        what = "add_nonnative"
        c0, o = self.read("add", field, NL + 1, (a, b))
        s = T(o[:NL], [("col", c0 + k) for k in range(NL)])
        overflow = o[NL]                                        # add_virtual_bool_target_unsafe: no assert_bool here
        with Circuit._UxBlock(self):
            sum_expected = self.add_biguint(a, b)
            mod_times_overflow = self.mul_biguint_by_bool_ux(self.modulus(field), overflow)
            sum_actual = self.add_biguint(s, mod_times_overflow)
            self.connect_biguint(sum_expected, sum_actual, what)
            if range_check:
                self.range_check_result(s, field, what)
        return s

    def add_many_nonnative(self, xs, field, range_check=False):  # :310-353
        what = "add_many_nonnative"
        if len(xs) == 1:
            return xs[0]
        c0, o = self.read("add_many", field, NL + 1, tuple(xs))
        s = T(o[:NL], [("col", c0 + k) for k in range(NL)])
        overflow = T(o[NL:], [("col", c0 + NL)])
        with Circuit._UxBlock(self):
            self.range_check_ux(s, BITS, what)
            self.range_check_ux(overflow, BITS, what)
            acc = const_t(0)                                    # zero_biguint: no limbs
            for x in xs:
                acc = self.add_biguint(acc, x)
            mod_times_overflow = self.mul_biguint(self.modulus(field), overflow)
            sum_actual = self.add_biguint(s, mod_times_overflow)
            self.connect_biguint(acc, sum_actual, what)
            if range_check:
                self.range_check_result(s, field, what)
        return s

    def sub_nonnative(self, a, b, field, range_check=False):    # :356-388
        what = "sub_nonnative"
        c0, o = self.read("sub", field, NL + 1, (a, b))
        d = T(o[:NL], [("col", c0 + k) for k in range(NL)])
        overflow = o[NL]
        with Circuit._UxBlock(self):
            self.range_check_ux(d, BITS, what)
            self.require(overflow in (0, 1), what, f"assert_bool(overflow): {overflow}")
            diff_plus_b = self.add_biguint(d, b)
            mod_times_overflow = self.mul_biguint_by_bool_ux(self.modulus(field), overflow)
            reduced = self.sub_biguint(diff_plus_b, mod_times_overflow)
            self.connect_biguint(a, reduced, what)
            if range_check:
                self.range_check_result(d, field, what)
        return d

    def mul_nonnative(self, x, y, field, range_check=False):    # :390-464
        what = "mul_nonnative"
        m = self.curve.moduli[field]
        # gate wire order gates/mul_nonnative.rs:41-59,384-390: r[9], q[9], check_sum[17], then b[16] of the CheckSumGate row
        c0, o = self.read("mul", field, 2 * NL + (2 * NL - 1) + (2 * NL - 2), (x, y))
        r, q, cs, b = o[0:9], o[9:18], o[18:35], o[35:51]
        # the gate's x / y wires: operand limbs, zero beyond the operand's limb count (:406-424)
        xw = x.v + [0] * (NL - len(x))
        yw = y.v + [0] * (NL - len(y))
        self.require(len(x) <= NL and len(y) <= NL, what, "operand has more than 9 limbs")
        # MulNonnativeGate::eval_unfiltered gates/mul_nonnative.rs:101-130 (17 constraints, in Goldilocks)
        m29 = const_limbs(m)
        m29 = m29 + [0] * (NL - len(m29))
        for i in range(2 * NL - 1):
            acc = 0
            for j in range(max(0, i - NL + 1), min(i + 1, NL)):
                acc += m29[j] * q[i - j] - xw[j] * yw[i - j]
            if i < NL:
                acc += r[i]
            self.require((acc - cs[i]) % P_GL == 0, what, f"MulNonnativeGate constraint {i}")
        # CheckSumGate::eval_unfiltered gates/mul_nonnative.rs:411-427 (a = check_sum via copy constraints :442-445)
        last = 0
        for i in range(2 * NL - 1):
            if i < 2 * NL - 2:
                ob = (b[i] - (1 << 33)) % P_GL
                self.require((cs[i] + last - B29 * ob) % P_GL == 0, what, f"CheckSumGate constraint {i}")
                last = ob
            else:
                self.require((cs[i] + last) % P_GL == 0, what, "CheckSumGate last constraint a16 + b15 == 0")
        # range checks :453-460
        rt = T(r, [("col", c0 + k) for k in range(NL)])
        self.range_check_ux(x, BITS, what + ":x")
        self.range_check_ux(y, BITS, what + ":y")
        self.range_check_ux(rt, BITS, what + ":r")
        self.range_check_ux(T(q), BITS, what + ":q")
        self.range_check_ux(T(b[0:8]), CARRY_OVER_BASE, what + ":b[0..8]")
        self.range_check_ux(T(b[8:]), CARRY_OVER_BASE, what + ":b[8..16]")
        with Circuit._UxBlock(self):
            if range_check:                                     # biguint_to_nonnative(&r_bigint, range_check) :462-463
                self.range_check_result(rt, field, what)
        return rt

    def inv_nonnative(self, x, field, range_check=False):       # :502-536
        what = "inv_nonnative"
        k = len(x)
        c0, o = self.read("inv", field, 2 * k, (x,))
        inv = T(o[:k], [("col", c0 + i) for i in range(k)])
        div = T(o[k:], [("col", c0 + k + i) for i in range(k)])
        with Circuit._UxBlock(self):
            product = self.mul_biguint(x, inv)
            mod_times_div = self.mul_biguint(self.modulus(field), div)
            expected = self.add_biguint(mod_times_div, const_t(1))
            self.connect_biguint(product, expected, what)
            if range_check:
                self.range_check_result(inv, field, what)
        return inv

    def neg_nonnative(self, x, field, range_check=False):       # :491-500
        return self.sub_nonnative(const_t(0), x, field, range_check)

    def nonnative_conditional_neg(self, x, b, field, range_check=False):   # :584-596
        not_b = self.not_(b)
        neg = self.neg_nonnative(x, field, False)
        t = self.mul_biguint_by_bool(neg, b)
        f = self.mul_biguint_by_bool(x, not_b)
        return self.add_nonnative(t, f, field, range_check)

    def connect_nonnative(self, a, b, what):                    # :207-213
        self.connect_biguint(a, b, what)

    # ---- gadgets/curve.rs ----------------------------------------------------------------------------------------
    def curve_assert_valid(self, p):                            # :123-135
        x, y = p
        F = FIELD_BASE
        a, b = const_t(self.curve.a), const_t(self.curve.b)
        y2 = self.mul_nonnative(y, y, F, True)
        x2 = self.mul_nonnative(x, x, F, False)
        x3 = self.mul_nonnative(x2, x, F, False)
        ax = self.mul_nonnative(a, x, F, False)
        axb = self.add_nonnative(ax, b, F, False)
        rhs = self.add_nonnative(x3, axb, F, True)
        self.connect_nonnative(y2, rhs, "curve_assert_valid: y^2 == x^3 + a x + b")

    def curve_conditional_neg(self, p, b):                      # :149-158
        return p[0], self.nonnative_conditional_neg(p[1], b, FIELD_BASE, True)

    def curve_double(self, p, range_check=False):               # :160-185
        x, y = p
        F = FIELD_BASE
        dy = self.add_nonnative(y, y, F, False)
        idy = self.inv_nonnative(dy, F, False)
        xx = self.mul_nonnative(x, x, F, False)
        t = self.add_many_nonnative([xx, xx, xx, const_t(self.curve.a)], F, False)
        lam = self.mul_nonnative(t, idy, F, False)
        lam2 = self.mul_nonnative(lam, lam, F, False)
        xd = self.add_nonnative(x, x, F, False)
        x3 = self.sub_nonnative(lam2, xd, F, range_check)
        xdf = self.sub_nonnative(x, x3, F, False)
        lx = self.mul_nonnative(lam, xdf, F, False)
        y3 = self.sub_nonnative(lx, y, F, range_check)
        return x3, y3

    def curve_repeated_double(self, p, n, range_check=False):   # :187-200
        for _ in range(n - 1):
            p = self.curve_double(p, False)
        return self.curve_double(p, range_check)

    def curve_add(self, p1, p2, range_check=False):             # :202-223
        (x1, y1), (x2, y2) = p1, p2
        F = FIELD_BASE
        u = self.sub_nonnative(y2, y1, F, False)
        v = self.sub_nonnative(x2, x1, F, False)
        vinv = self.inv_nonnative(v, F, False)
        s = self.mul_nonnative(u, vinv, F, False)
        s2 = self.mul_nonnative(s, s, F, False)
        xs = self.add_nonnative(x2, x1, F, False)
        x3 = self.sub_nonnative(s2, xs, F, range_check)
        xd = self.sub_nonnative(x1, x3, F, False)
        pr = self.mul_nonnative(s, xd, F, False)
        y3 = self.sub_nonnative(pr, y1, F, range_check)
        return x3, y3

    def curve_conditional_add(self, p1, p2, b, range_check=False):   # :225-243
        not_b = self.not_(b)
        s = self.curve_add(p1, p2, False)
        xt = self.mul_biguint_by_bool(s[0], b)
        yt = self.mul_biguint_by_bool(s[1], b)
        xf = self.mul_biguint_by_bool(p1[0], not_b)
        yf = self.mul_biguint_by_bool(p1[1], not_b)
        x = self.add_nonnative(xt, xf, FIELD_BASE, range_check)
        y = self.add_nonnative(yt, yf, FIELD_BASE, range_check)
        return x, y

    # ---- gadgets/split_nonnative.rs:25-72 ------------------------------------------------------------------------
    def _bits(self, t):
        """split_le_base::<2>(limb, 29) per limb: BaseSumGate forces bits in {0, 1} that recombine to the limb,
        which exists iff limb < 2^29"""
        bits = []
        for l in t.v:
            self.require(0 <= l < B29, "split_le_base", f"limb {l} has no 29-bit decomposition")
            bits.extend((l >> i) & 1 for i in range(BITS))
        self._aux(bits)
        return bits

    def split_4(self, t):
        bits = self._bits(t)
        while len(bits) % 4:
            bits.append(0)
        out, comb = [], []
        for i in range(0, len(bits), 4):
            lower = bits[i] + 2 * bits[i + 1]
            upper = bits[i + 2] + 2 * bits[i + 3]
            limb = lower + 4 * upper
            comb += [lower, upper, limb]
            out.append(limb)
        self._aux(comb)
        return out

    def split_2(self, t):
        bits = self._bits(t)
        while len(bits) % 2:
            bits.append(0)
        out = [bits[i] + 2 * bits[i + 1] for i in range(0, len(bits), 2)]
        self._aux(out)
        return out

    def random_access_curve_points(self, idx, table):           # gadgets/curve_windowed_mul.rs:74-118
        self.require(0 <= idx < len(table), "random_access", f"index {idx} out of range")
        px, py = table[idx]
        xv = px.v + [0] * (NL - len(px))                        # .get(i).unwrap_or(&zero)
        yv = py.v + [0] * (NL - len(py))
        src = self._aux(xv + yv)
        # [upstream-from-memory] one RandomAccessGate op per limb (9 x + 9 y): its generator also fills the bit wires of
        # the access index, least significant first (log2(len(table)) of them)
        nbits = (len(table) - 1).bit_length()
        for _ in range(2 * NL):
            self.gate.extend((idx >> b) & 1 for b in range(nbits))
        return T(xv, src[:NL]), T(yv, src[NL:])

    # ---- gadgets/curve_fixed_base.rs:18-66 ------------------------------------------------------------------------
    def fixed_base_curve_mul(self, base, scalar):
        nwin = len(scalar) * 8
        limbs = self.split_4(scalar)
        C = self.curve
        rando = C.hash_point(32)
        result = const_point(rando)
        point = base
        for i, limb in enumerate(limbs):
            if i >= nwin:
                break
            with self.scope(f"win{i}"):
                muls = C.fixed_base_window(point)
                tbl = [const_point(muls[0])] + [const_point(q) for q in muls]
                should_add = self.not_(self.is_equal_zero(limb))
                r = self.random_access_curve_points(limb, tbl)
                result = self.curve_conditional_add(result, r, should_add, False)
            for _ in range(4):
                point = C.double(point)
        with self.scope("unblind"):
            return self.curve_add(result, const_point(C.neg(rando)), True)

    # ---- gadgets/curve_msm.rs:21-79 ---------------------------------------------------------------------------------
    def curve_msm(self, p, q, n, m):
        limbs_n = self.split_2(n)
        limbs_m = self.split_2(m)
        self.require(len(limbs_n) == len(limbs_m), "curve_msm", "digit counts differ")
        num = len(limbs_n)
        rando = R.rando_point()
        rando_t = const_point(rando)
        neg_rando = const_point(R.ec_neg(rando))
        pre = [p] * 16
        cur_p, cur_q = rando_t, rando_t
        with self.scope("table"):
            for i in range(4):
                pre[i] = cur_p
                pre[4 * i] = cur_q
                cur_p = self.curve_add(cur_p, p, False)
                cur_q = self.curve_add(cur_q, q, False)
            for i in range(1, 4):
                pre[i] = self.curve_add(pre[i], neg_rando, False)
                pre[4 * i] = self.curve_add(pre[4 * i], neg_rando, False)
            for i in range(1, 4):
                for j in range(1, 4):
                    pre[i + 4 * j] = self.curve_add(pre[i], pre[4 * j], False)
        result = rando_t
        for d in reversed(range(num)):
            with self.scope(f"digit{d}"):
                result = self.curve_repeated_double(result, 2, False)
                idx = 4 * limbs_m[d] + limbs_n[d]               # mul_add(four, limb_m, limb_n)
                self._aux([idx])
                r = self.random_access_curve_points(idx, pre)
                should_add = self.not_(self.is_equal_zero(idx))
                result = self.curve_conditional_add(result, r, should_add, False)
        spm = rando
        for _ in range(2 * num):
            spm = R.ec_double(spm)
        with self.scope("unblind"):
            return self.curve_add(result, const_point(R.ec_neg(spm)), True)

    # ---- gadgets/glv.rs:53-104 ----------------------------------------------------------------------------------------
    def decompose_secp256k1_scalar(self, k):
        S = FIELD_SCALAR
        c0, o = self.read("glv", S, 12, (k,))
        k1 = T(o[0:5], [("col", c0 + i) for i in range(5)])
        k2 = T(o[5:10], [("col", c0 + 5 + i) for i in range(5)])
        n1, n2 = o[10], o[11]                                   # add_virtual_bool_target_unsafe
        k1_raw = self.nonnative_conditional_neg(k1, n1, S, False)
        k2_raw = self.nonnative_conditional_neg(k2, n2, S, False)
        sb = self.mul_nonnative(const_t(R.GLV_S), k2_raw, S, False)
        sb = self.add_nonnative(sb, k1_raw, S, True)
        self.connect_nonnative(sb, k, "decompose_secp256k1_scalar: k1_raw + GLV_S * k2_raw == k")
        return k1, k2, n1, n2

    def glv_mul(self, p, k):
        with self.scope("decompose"):
            k1, k2, n1, n2 = self.decompose_secp256k1_scalar(k)
        beta_px = self.mul_nonnative(const_t(R.GLV_BETA), p[0], FIELD_BASE, True)
        sp = (beta_px, p[1])
        p_neg = self.curve_conditional_neg(p, n1)
        sp_neg = self.curve_conditional_neg(sp, n2)
        with self.scope("msm"):
            return self.curve_msm(p_neg, sp_neg, k1, k2)

    # ---- gadgets/ecdsa.rs:30-53 -----------------------------------------------------------------------------------------
    def verify_secp256k1_message(self, msg, r, s, pk):
        with self.scope("assert_valid"):
            self.curve_assert_valid(pk)
        c = self.inv_nonnative(s, FIELD_SCALAR, False)
        u1 = self.mul_nonnative(msg, c, FIELD_SCALAR, True)
        u2 = self.mul_nonnative(r, c, FIELD_SCALAR, True)
        with self.scope("fixed_base"):
            point1 = self.fixed_base_curve_mul(R.G, u1)
        with self.scope("glv_mul"):
            point2 = self.glv_mul(pk, u2)
        with self.scope("final_add"):
            point = self.curve_add(point1, point2, True)
        self.connect_nonnative(r, point[0], "verify: connect_nonnative(r, point.x)")

    # ---- gadgets/curve.rs:137-147, :245-285, gadgets/curve_windowed_mul.rs:52-72,131-173,
    # gadgets/ecdsa.rs:55-78.  The rand() points of precompute_window (:57) and curve_scalar_mul (:253) are inputs.
    def curve_neg(self, p, range_check=False):
        return p[0], self.neg_nonnative(p[1], FIELD_BASE, range_check)

    def precompute_window(self, p, g):
        neg = const_point(self.curve.neg(g))
        multiples = [const_point(g)]
        for i in range(1, 16):
            multiples.append(self.curve_add(p, multiples[i - 1], True))
        for i in range(1, 16):
            multiples[i] = self.curve_add(neg, multiples[i], True)
        return multiples

    def curve_scalar_mul_windowed(self, p, n, g, range_check=True):
        C = self.curve
        windows = self.split_4(n)
        starting_point = C.hash_point(25)
        spm = starting_point
        for _ in range(len(windows) * 4):
            spm = C.double(spm)
        result = const_point(starting_point)
        with self.scope("precompute"):
            pre = self.precompute_window(p, g)
        for i in reversed(range(len(windows))):
            with self.scope(f"window{i}"):
                result = self.curve_repeated_double(result, 4, False)
                to_add = self.random_access_curve_points(windows[i], pre)
                should_add = self.not_(self.is_equal_zero(windows[i]))
                result = self.curve_conditional_add(result, to_add, should_add, False)
        with self.scope("unblind"):
            to_add = self.curve_neg(const_point(spm), False)
            return self.curve_add(result, to_add, range_check)

    def curve_scalar_mul(self, p, n, rando, range_check=True):
        bits = self._bits(n)                                    # split_nonnative_to_bits gadgets/nonnative.rs:566-582
        randot = const_point(rando)
        # add_virtual_affine_point_target + connect_affine_point: 9-limb virtual targets that carry the constant's limbs
        # (connect_biguint asserts the limbs beyond the shorter operand to be zero, gadgets/biguint.rs:181-196)
        vx, vy = randot[0].v + [0] * (NL - len(randot[0])), randot[1].v + [0] * (NL - len(randot[1]))
        result = (T(vx, [("virt", "result.x", k) for k in range(NL)]), T(vy, [("virt", "result.y", k) for k in range(NL)]))
        self.connect_nonnative(randot[0], result[0], "curve_scalar_mul: result == rando")
        self.connect_nonnative(randot[1], result[1], "curve_scalar_mul: result == rando")
        two_i_times_p = p
        for i, bit in enumerate(bits):
            with self.scope(f"bit{i}"):
                not_bit = self.not_(bit)
                rp = self.curve_add(result, two_i_times_p, False)
                xt = self.mul_biguint_by_bool(rp[0], bit)
                xf = self.mul_biguint_by_bool(result[0], not_bit)
                yt = self.mul_biguint_by_bool(rp[1], bit)
                yf = self.mul_biguint_by_bool(result[1], not_bit)
                new_x = self.add_nonnative(xt, xf, FIELD_BASE, False)
                new_y = self.add_nonnative(yt, yf, FIELD_BASE, False)
                result = (new_x, new_y)
                two_i_times_p = self.curve_double(two_i_times_p, False)
        with self.scope("unblind"):
            neg_r = self.curve_neg(randot, False)
            return self.curve_add(result, neg_r, range_check)

    def verify_p256_message(self, msg, r, s, pk, g):
        with self.scope("assert_valid"):
            self.curve_assert_valid(pk)
        c = self.inv_nonnative(s, FIELD_SCALAR, False)
        u1 = self.mul_nonnative(msg, c, FIELD_SCALAR, True)
        u2 = self.mul_nonnative(r, c, FIELD_SCALAR, True)
        with self.scope("fixed_base"):
            point1 = self.fixed_base_curve_mul(self.curve.g, u1)
        with self.scope("windowed_mul"):
            point2 = self.curve_scalar_mul_windowed(pk, u2, g, True)
        with self.scope("final_add"):
            point = self.curve_add(point1, point2, True)
        self.connect_nonnative(r, point[0], "verify: connect_nonnative(r, point.x)")

    def finish(self):
        self.require(self.cur == len(self.cols), "end", f"{len(self.cols) - self.cur} witness columns were never read")
        if self.aux_in is not None:
            self.require(len(self.aux_in) == len(self.aux), "end", "aux vector length")
        if self.ux_in is not None:
            self.require(len(self.ux_in) == len(self.ux), "end", "ux vector length")


def const_point(pt):
    """constant_affine_point gadgets/curve.rs:99-105"""
    return const_t(pt[0]), const_t(pt[1])


# ---- entry points ---------------------------------------------------------------------------------------------------------
def check_verify(cols, msg, r, s, pkx, pky, aux=None, ux=None):
    """Replays every constraint of verify_secp256k1_message_circuit on the 82 615 hot-path columns `cols` of ONE
    signature with inputs (msg, r, s, pk) as integers.  Raises ConstraintViolation; returns the Circuit (its
    .aux / .ux are the derived 8(f) rank 1 / rank 2 vectors, .gens the generator table with operand wiring)."""
    c = Circuit(cols, aux=aux, ux=ux)
    c.verify_secp256k1_message(input_t("msg", msg), input_t("r", r), input_t("s", s),
                               (input_t("pkx", pkx), input_t("pky", pky)))
    c.finish()
    return c


def check_glv_mul(cols, px, py, k, aux=None, ux=None):
    c = Circuit(cols, aux=aux, ux=ux)
    c.glv_mul((input_t("pkx", px), input_t("pky", py)), input_t("k", k))
    c.finish()
    return c


def check_windowed_mul(curve, cols, px, py, k, g, aux=None):
    """curve_scalar_mul_windowed(p, k) (gadgets/curve_windowed_mul.rs:131-173) with precompute_window's point g"""
    c = Circuit(cols, aux=aux, curve=curve)
    pt = c.curve_scalar_mul_windowed((input_t("px", px), input_t("py", py)), input_t("k", k), g, True)
    c.finish()
    return c, (pt[0].value(), pt[1].value())


def check_scalar_mul(curve, cols, px, py, k, rando, aux=None):
    """curve_scalar_mul(p, k) (gadgets/curve.rs:245-285) with blinding point rando"""
    c = Circuit(cols, aux=aux, curve=curve)
    pt = c.curve_scalar_mul((input_t("px", px), input_t("py", py)), input_t("k", k), rando, True)
    c.finish()
    return c, (pt[0].value(), pt[1].value())


def check_verify_p256(cols, msg, r, s, pkx, pky, g, aux=None):
    c = Circuit(cols, aux=aux, curve=R.P256)
    c.verify_p256_message(input_t("msg", msg), input_t("r", r), input_t("s", s), (input_t("pkx", pkx), input_t("pky", pky)), g)
    c.finish()
    return c


def unpack_inputs(arrs, i):
    """(n, 32) uint8 little-endian arrays -> ints of element i"""
    return [int.from_bytes(bytes(bytearray(a[i])), "little") for a in arrs]


if __name__ == "__main__":   # python oracle/check_circuit.py: replay the committed goldens
    import os
    import sys
    import time

    import numpy as np

    gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    z = np.load(os.path.join(gold, "verify_golden.npz"))
    a = np.load(os.path.join(gold, "aux_golden.npz"))
    for i in range(z["cols"].shape[1]):
        t = time.time()
        ins = [int.from_bytes(bytes(bytearray(z["inputs"][i, k])), "little") for k in range(5)]
        try:
            c = check_verify(z["cols"][:, i], *ins, aux=a["verify"][:, i])
            print(f"signature {i}: {len(c.gens)} generators, {len(c.aux)} built-in values, {len(c.ux)} U29 values: "
                  f"all constraints hold ({time.time() - t:.2f} s)")
        except ConstraintViolation as e:
            print(f"signature {i}: VIOLATION {e} (golden valid flag {int(z['valid'][i])})")
            if z["valid"][i]:
                sys.exit(1)
