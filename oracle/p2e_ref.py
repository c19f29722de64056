"""Python big-int restatement of the plonky2-ecdsa witness generators (TEST INFRASTRUCTURE ONLY).

This file is the *independent* restatement used (a) to generate the golden fixtures under
``tests/golden/`` (see ``oracle/gen_golden.py``) and (b) to cross-check the fixed-width C oracle
(``oracle/p2e_oracle.c``).  It is never imported by the product path (``plonky2-ecdsa_amd/``); only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may touch ``oracle/``.

PARITY STATUS: "parity unpinned" at the level of literal reference outputs -- the reference
(Weobe/plonky2-ecdsa, Rust) holds no golden vectors for these generators and
cannot be built here (no rustc; plonky2 / plonky2_ux / num are un-vendored git deps).  What pins
this restatement instead: (i) every output is re-checked against the reference's *constraint
equations* (``check_*`` functions below, citing the gate / gadget lines they restate), (ii) the
outputs are mathematically unique given those equations except the GLV rounding rule, which is
restated from num::rational::Ratio::round, and (iii) two independent implementations (this big-int
one and the fixed-width C one) must agree bit for bit.

Every function cites the reference file:line it follows (paths relative to /root/reference/src).
Third-party arithmetic that is absent from /root/reference (plonky2 Secp256K1Base/Scalar,
Goldilocks, num::BigUint, Keccak-256) is restated from its published definition.
"""
from __future__ import annotations

# ----------------------------------------------------------------------------------------------
# constants (curve/secp256k1.rs:15-38, curve/glv.rs:11-32, gadgets/nonnative.rs:32)
# ----------------------------------------------------------------------------------------------
P = 2**256 - 2**32 - 977                     # Secp256K1Base::order()
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141  # Secp256K1Scalar::order()
P_GL = 2**64 - 2**32 + 1                     # Goldilocks
BITS = 29                                    # gadgets/nonnative.rs:32
NL = 9                                       # ceil(256/29): gadgets/nonnative.rs:176-178, gates/mul_nonnative.rs:37-39
MASK29 = (1 << BITS) - 1

FIELD_BASE = 0
FIELD_SCALAR = 1
FIELD_P256_BASE = 2
FIELD_P256_SCALAR = 3
MODULI = {FIELD_BASE: P, FIELD_SCALAR: N}     # secp256k1; the P-256 pair is added below (MODULI[2], MODULI[3])


def _u64le(ws):
    return sum(int(w) << (64 * i) for i, w in enumerate(ws))


GX = _u64le([0x59F2815B16F81798, 0x029BFCDB2DCE28D9, 0x55A06295CE870B07, 0x79BE667EF9DCBBAC])
GY = _u64le([0x9C47D08FFB10D4B8, 0xFD17B448A6855419, 0x5DA4FBFC0E1108A8, 0x483ADA7726A3C465])
CURVE_A = 0
CURVE_B = 7
GLV_BETA = _u64le([13923278643952681454, 11308619431505398165, 7954561588662645993, 8856726876819556112])
GLV_S = _u64le([16069571880186789234, 1310022930574435960, 11900229862571533402, 6008836872998760672])
GLV_A1 = _u64le([16747920425669159701, 3496713202691238861, 0, 0])
GLV_MINUS_B1 = _u64le([8022177200260244675, 16448129721693014056, 0, 0])
GLV_A2 = _u64le([6323353552219852760, 1498098850674701302, 1, 0])
GLV_B2 = _u64le([16747920425669159701, 3496713202691238861, 0, 0])

# error bits (mirrors include/p2e.h)
ERR_LIMB_RANGE = 1        # gates/mul_nonnative.rs:262,271,275-276 ; gadgets/biguint.rs:456,473
ERR_VALUE_GE_2_256 = 2    # Secp256K1*::from_noncanonical_biguint (template field/p256_base.rs:121-130)
ERR_INVERSE_OF_ZERO = 4   # gadgets/nonnative.rs:863
ERR_CARRY_RANGE = 8       # gates/mul_nonnative.rs:527
ERR_DIVISION_BY_ZERO = 32  # BigUint::div_rem by zero panics (BigUintDivRemGenerator)
ERR_QUOTIENT_RANGE = 16   # q does not fit the gate's nine q wires (documented deviation: flagged, not emitted)


class RefPanic(Exception):
    """The reference would panic / return Err here.  ``code`` is the error bit."""

    def __init__(self, code, msg=""):
        super().__init__(msg or f"reference panic, code {code}")
        self.code = code


# ----------------------------------------------------------------------------------------------
# Keccak-256 (legacy 0x01 padding) -- plonky2 KeccakHash<32>::hash_no_pad uses keccak_hash::keccak
# ----------------------------------------------------------------------------------------------
_KECCAK_RC = [
    0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000,
    0x000000000000808B, 0x0000000080000001, 0x8000000080008081, 0x8000000000008009,
    0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
    0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003,
    0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
    0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008,
]
_KECCAK_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]
_M64 = (1 << 64) - 1


def _rol(x, n):
    n %= 64
    return ((x << n) | (x >> (64 - n))) & _M64 if n else x


def _keccak_f(a):
    for rc in _KECCAK_RC:
        c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
        d = [c[(x - 1) % 5] ^ _rol(c[(x + 1) % 5], 1) for x in range(5)]
        a = [[a[x][y] ^ d[x] for y in range(5)] for x in range(5)]
        b = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                b[y][(2 * x + 3 * y) % 5] = _rol(a[x][y], _KECCAK_ROT[x][y])
        a = [[b[x][y] ^ ((~b[(x + 1) % 5][y]) & b[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        a[0][0] ^= rc
    return a


def keccak256(data: bytes) -> bytes:
    rate = 136
    msg = bytearray(data)
    msg.append(0x01)
    while len(msg) % rate:
        msg.append(0)
    msg[-1] |= 0x80
    a = [[0] * 5 for _ in range(5)]
    for off in range(0, len(msg), rate):
        blk = msg[off:off + rate]
        for i in range(rate // 8):
            a[i % 5][i // 5] ^= int.from_bytes(blk[8 * i:8 * i + 8], "little")
        a = _keccak_f(a)
    out = b"".join(a[i % 5][i // 5].to_bytes(8, "little") for i in range(4))
    return out


# ----------------------------------------------------------------------------------------------
# native curve helpers (curve/curve_types.rs:83-102 double, :262-269 neg; curve_adds.rs affine add)
# only used for *constants* (rando, fixed-base table) and for synthesising inputs.
# ----------------------------------------------------------------------------------------------
def ec_add(p1, p2):
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        return ec_double(p1)
    lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    return x3, (lam * (x1 - x3) - y1) % P


def ec_double(p):
    if p is None:
        return None
    x1, y1 = p
    lam = (3 * x1 * x1 + CURVE_A) * pow(2 * y1, -1, P) % P
    x3 = (lam * lam - 2 * x1) % P
    return x3, (lam * (x1 - x3) - y1) % P


def ec_neg(p):
    return None if p is None else (p[0], (-p[1]) % P)


def ec_mul(k, p):
    r = None
    while k:
        if k & 1:
            r = ec_add(r, p)
        p = ec_double(p)
        k >>= 1
    return r


G = (GX, GY)


class Curve:
    """A short-Weierstrass curve of the crate (curve/curve_types.rs:15-42): base modulus p, group order n, a, b, G.
    The native helpers below are only used for CONSTANTS (blinding points, fixed-base tables) and to synthesise
    inputs; FIELD_BASE / FIELD_SCALAR index `moduli`."""

    def __init__(self, name, p, n, a, b, g):
        self.name, self.p, self.n, self.a, self.b, self.g = name, p, n, a % p, b % p, g
        self.moduli = {FIELD_BASE: p, FIELD_SCALAR: n}
        self._fb = {}

    def add(self, p1, p2):                                      # curve/curve_adds.rs (affine, incl. the neutral element)
        if p1 is None:
            return p2
        if p2 is None:
            return p1
        (x1, y1), (x2, y2) = p1, p2
        if x1 == x2:
            if (y1 + y2) % self.p == 0:
                return None
            return self.double(p1)
        lam = (y2 - y1) * pow(x2 - x1, -1, self.p) % self.p
        x3 = (lam * lam - x1 - x2) % self.p
        return x3, (lam * (x1 - x3) - y1) % self.p

    def double(self, pt):                                       # curve/curve_types.rs:83-102
        if pt is None:
            return None
        x1, y1 = pt
        lam = (3 * x1 * x1 + self.a) * pow(2 * y1, -1, self.p) % self.p
        x3 = (lam * lam - 2 * x1) % self.p
        return x3, (lam * (x1 - x3) - y1) % self.p

    def neg(self, pt):
        return None if pt is None else (pt[0], (-pt[1]) % self.p)

    def mul(self, k, pt):
        r = None
        while k:
            if k & 1:
                r = self.add(r, pt)
            pt = self.double(pt)
            k >>= 1
        return r

    def on_curve(self, pt):
        x, y = pt
        return (y * y - (x * x * x + self.a * x + self.b)) % self.p == 0

    def hash_point(self, nbytes):
        """KeccakHash::<N>::hash_no_pad(&[F::ZERO]) read as a little-endian integer, times G
        (gadgets/curve_fixed_base.rs:34-38 and gadgets/curve_msm.rs:33-37 with N = 32,
        gadgets/curve_windowed_mul.rs:140-144 with N = 25).  [upstream-from-memory: plonky2 hash/keccak.rs --
        KeccakHash<N> keeps the first N bytes of keccak256 of the field elements' 8-byte LE encodings;
        from_noncanonical_biguint does not reduce]"""
        scalar = int.from_bytes(keccak256(bytes(8))[:nbytes], "little")
        return self.mul(scalar, self.g)

    def fixed_base_window(self, point):
        """t * point for t = 1..15 (gadgets/curve_fixed_base.rs:45-54)."""
        if point not in self._fb:
            out, acc = [], None
            for _ in range(15):
                acc = self.add(point, acc)
                out.append(acc)
            self._fb[point] = out
        return self._fb[point]


SECP256K1 = Curve("secp256k1", P, N, CURVE_A, CURVE_B, G)
# curve/p256.rs:15-57, field/p256_base.rs:14-17, field/p256_scalar.rs:5
P256_P = 0xffffffff00000001000000000000000000000000ffffffffffffffffffffffff
P256_N = 0xffffffff00000000ffffffffffffffffbce6faada7179e84f3b9cac2fc632551
P256 = Curve("p256", P256_P, P256_N, P256_P - 3,
             0x5AC635D8AA3A93E7B3EBBD55769886BC651D06B0CC53B0F63BCE3C3E27D2604B,
             (0x6B17D1F2E12C4247F8BCE6E563A440F277037D812DEB33A0F4A13945D898C296,
              0x4FE342E2FE1A7F9B8EE7EB4A7C0F9E162BCE33576B315ECECBB6406837BF51F5))
CURVES = {"secp256k1": SECP256K1, "p256": P256}
MODULI[FIELD_P256_BASE], MODULI[FIELD_P256_SCALAR] = P256_P, P256_N


def rando_point():
    """gadgets/curve_fixed_base.rs:34-38 == gadgets/curve_msm.rs:33-37.

    hash_0 = KeccakHash::<32>::hash_no_pad(&[F::ZERO]) = keccak256 of the 8-byte LE encoding of 0
    [upstream-from-memory: plonky2 hash/keccak.rs]; scalar = BigUint::from_bytes_le(bytes), NOT
    reduced (from_noncanonical_biguint)."""
    h = keccak256(bytes(8))
    scalar = int.from_bytes(h, "little")
    return ec_mul(scalar, G)


# ----------------------------------------------------------------------------------------------
# gadgets/biguint.rs:27-51 convert_base (literal restatement)
# ----------------------------------------------------------------------------------------------
def convert_base(x, frm, to):
    res = []
    rem = 0
    offset = 0
    mask = (1 << to) - 1
    for d in x:
        rem += d << offset
        offset += frm
        while offset >= to:
            cur = rem & mask
            if cur >= 1 << 32:
                raise RefPanic(ERR_LIMB_RANGE, "Value doesn't fit in u32")
            res.append(cur)
            rem >>= to
            offset -= to
    if rem != 0:
        if rem >= 1 << 32:
            raise RefPanic(ERR_LIMB_RANGE, "Value doesn't fit in u32")
        res.append(rem)
    return res


def to_u32_digits(v):
    """num::BigUint::to_u32_digits (zero -> empty vec)."""
    out = []
    while v:
        out.append(v & 0xFFFFFFFF)
        v >>= 32
    return out


def limbs_of(v, num_limbs):
    """set_biguint_target: gadgets/biguint.rs:454-463,471-480."""
    limbs = convert_base(to_u32_digits(v), 32, BITS)
    if len(limbs) > num_limbs:
        raise RefPanic(ERR_LIMB_RANGE, "assert!(target.num_limbs() >= limbs.len())")
    return limbs + [0] * (num_limbs - len(limbs))


def value_of(limbs):
    """get_biguint_target: gadgets/biguint.rs:444-452 (limbs are arbitrary Goldilocks elements)."""
    acc = 0
    for l in reversed(limbs):
        acc = (acc << BITS) + l
    return acc


def const_limbs(v):
    """constant_biguint: gadgets/biguint.rs:165-175 -- variable limb count, 0 -> no limbs (Q5)."""
    return convert_base(to_u32_digits(v), 32, BITS)


def canon(v, m):
    """FF::from_noncanonical_biguint (panics >= 2^256) then to_canonical_biguint (ONE cond. subtract).
    [plonky2 secp256k1_base.rs; in-tree template field/p256_base.rs:121-130,162-168]."""
    if v >> 256:
        raise RefPanic(ERR_VALUE_GE_2_256, "error converting to u64 array")
    return v - m if v >= m else v


# ----------------------------------------------------------------------------------------------
# the seven run_once bodies
# ----------------------------------------------------------------------------------------------
def gen_add(a_limbs, b_limbs, m):
    """NonNativeAdditionGenerator::run_once gadgets/nonnative.rs:626-645.  Q1: strict '>'."""
    a = canon(value_of(a_limbs), m)
    b = canon(value_of(b_limbs), m)
    s = a + b
    if s > m:
        return limbs_of(s - m, NL), 1
    return limbs_of(s, NL), 0


def gen_sub(a_limbs, b_limbs, m):
    """NonNativeSubtractionGenerator::run_once gadgets/nonnative.rs:792-810."""
    a = canon(value_of(a_limbs), m)
    b = canon(value_of(b_limbs), m)
    if a >= b:
        return limbs_of(a - b, NL), 0
    return limbs_of(m + a - b, NL), 1


def gen_add_many(summands, m):
    """NonNativeMultipleAddsGenerator::run_once gadgets/nonnative.rs:696-728 (true div_rem; Q11)."""
    s = sum(canon(value_of(x), m) for x in summands)
    ov, red = divmod(s, m)
    ov32 = (ov & _M64) & 0xFFFFFFFF  # to_u64_digits()[0] as u32
    return limbs_of(red, NL), ov32


def gen_inv(x_limbs, m):
    """NonNativeInverseGenerator::run_once gadgets/nonnative.rs:857-872.
    Output limb count = x.value.num_limbs() (gadgets/nonnative.rs:507-509)."""
    k = len(x_limbs)
    x = canon(value_of(x_limbs), m)
    if x == 0:
        raise RefPanic(ERR_INVERSE_OF_ZERO, "Tried to invert zero")
    inv = pow(x, -1, m)
    div = (x * inv) // m
    return limbs_of(inv, k), limbs_of(div, k)


def gen_div_rem(a_limbs, b_limbs):
    """BigUintDivRemGenerator::run_once gadgets/biguint.rs:508-518; output limb counts as div_rem_biguint allocates
    them (:391-397): div has a_len - b_len + 1 limbs (0 if b_len > a_len + 1), rem has b_len.
    Deviation of the batch API (all backends): limbs must be proper 29-bit limbs, else ERR_LIMB_RANGE (the reference's
    get_biguint_target would sum arbitrary field elements)."""
    if any(l >> BITS for l in list(a_limbs) + list(b_limbs)):
        raise RefPanic(ERR_LIMB_RANGE, "limb >= 2^29")
    a, b = value_of(a_limbs), value_of(b_limbs)
    if b == 0:
        raise RefPanic(ERR_DIVISION_BY_ZERO, "attempt to divide by zero")
    div, rem = divmod(a, b)
    na, nb = len(a_limbs), len(b_limbs)
    nd = 0 if nb > na + 1 else na - nb + 1
    return limbs_of(div, nd), limbs_of(rem, nb)       # set_biguint_target asserts the value fits (:456,473)


def gl(v):
    return v % P_GL


def gen_mul(x_limbs, y_limbs, m):
    """MulNonnativeGenerator::run_once gates/mul_nonnative.rs:249-324.
    x/y are the 9 gate wires (the gadget zero-pads shorter operands, gadgets/nonnative.rs:406-424).
    Returns r[9], q[9], check_sum[17] (gate wire order :41-59)."""
    assert len(x_limbs) == NL and len(y_limbs) == NL
    m29 = convert_base(to_u32_digits(m), 32, BITS)
    for v in list(x_limbs) + list(y_limbs):
        if v >= 1 << 32 or v >= 1 << BITS:
            raise RefPanic(ERR_LIMB_RANGE, "limb not < 2^29")
    x = value_of(x_limbs)   # == BigUint::from_slice(convert_base(x29, 29, 32))
    y = value_of(y_limbs)
    q, r = divmod(x * y, m)
    q29 = convert_base(to_u32_digits(q), 32, BITS)
    r29 = convert_base(to_u32_digits(r), 32, BITS)
    if len(q29) > NL:
        # The reference would set only q[0..9] and produce an unsatisfiable witness (CheckSumGenerator
        # then usually panics at :527).  Outside the gate's domain; we flag it instead.
        raise RefPanic(ERR_QUOTIENT_RANGE, "quotient does not fit 9 limbs")
    q29 += [0] * max(0, NL - len(q29))
    r29 += [0] * max(0, NL - len(r29))
    m29 += [0] * max(0, NL - len(m29))
    cs = []
    for i in range(2 * NL - 1):
        lo = max(0, i - NL + 1)
        hi = min(i + 1, NL)
        acc = 0
        for j in range(lo, hi):
            acc = gl(acc + gl(q29[i - j] * m29[j]) - gl(x_limbs[j] * y_limbs[i - j]))
        if i < NL:
            acc = gl(acc + r29[i])
        cs.append(acc)
    return r29[:NL], q29[:NL], cs


_INV_2_29 = pow(1 << BITS, -1, P_GL)


def gen_checksum(a):
    """CheckSumGenerator::run_once gates/mul_nonnative.rs:513-531 (Goldilocks field division)."""
    assert len(a) == 2 * NL - 1
    last = 0
    out = []
    for i in range(2 * NL - 2):
        b = gl((a[i] + last) * _INV_2_29)
        v = gl(b + (1 << 33))
        out.append(v)
        last = b
        if v >= 1 << 34:
            raise RefPanic(ERR_CARRY_RANGE, "carry out of range")
    return out


def glv_decompose(k):
    """curve/glv.rs:39-77 decompose_secp256k1_scalar; rounding = num Ratio::round (Q12)."""
    def rnd(num, den):
        # Ratio::new reduces; for odd reduced denominator: up iff fract.numer >= den/2 + 1
        from math import gcd
        g = gcd(num, den)
        num //= g
        den //= g
        q, r = divmod(num, den)
        if den % 2 == 0:
            up = r >= den // 2
        else:
            up = r >= den // 2 + 1
        return q + 1 if up else q

    c1 = canon(rnd(GLV_B2 * k, N), N)
    c2 = canon(rnd(GLV_MINUS_B1 * k, N), N)
    k1_raw = (k - c1 * GLV_A1 - c2 * GLV_A2) % N
    k2_raw = (c1 * GLV_MINUS_B1 - c2 * GLV_B2) % N
    assert (k1_raw + GLV_S * k2_raw) % N == k % N
    half = N // 2
    k1_neg = k1_raw > half
    k1 = N - k1_raw if k1_neg else k1_raw
    k2_neg = k2_raw > half
    k2 = N - k2_raw if k2_neg else k2_raw
    return k1, k2, int(k1_neg), int(k2_neg)


def gen_glv(k_limbs):
    """GLVDecompositionGenerator::run_once gadgets/glv.rs:128-142; k1/k2 have ceil(128/29)=5 limbs
    (gadgets/glv.rs:62-63)."""
    k = canon(value_of(k_limbs), N)
    k1, k2, n1, n2 = glv_decompose(k)
    return limbs_of(k1, 5), limbs_of(k2, 5), n1, n2


# ----------------------------------------------------------------------------------------------
# constraint checkers (what actually pins the generator outputs)
# ----------------------------------------------------------------------------------------------
def check_mul_gate(x, y, r, q, cs, m):
    """MulNonnativeGate::eval_unfiltered gates/mul_nonnative.rs:101-130 -- all 17 constraints zero."""
    m29 = const_limbs(m)
    m29 += [0] * (NL - len(m29))
    for i in range(2 * NL - 1):
        lo = max(0, i - NL + 1)
        hi = min(i + 1, NL)
        acc = 0
        for j in range(lo, hi):
            acc += m29[j] * q[i - j] - x[j] * y[i - j]
        if i < NL:
            acc += r[i]
        if gl(acc - cs[i]) != 0:
            return False
    return True


def check_checksum_gate(a, b):
    """CheckSumGate::eval_unfiltered gates/mul_nonnative.rs:411-427 + range checks nonnative.rs:459-460."""
    last = 0
    for i in range(2 * NL - 1):
        if i < 2 * NL - 2:
            ob = gl(b[i] - (1 << 33))
            if gl(a[i] + last - (1 << BITS) * ob) != 0:
                return False
            last = ob
            if b[i] >= 1 << 34:
                return False
        elif gl(a[i] + last) != 0:
            return False
    return True


# ----------------------------------------------------------------------------------------------
# witness-time walk of the gadgets.  A "nonnative target" is just its list of limb values.
# ----------------------------------------------------------------------------------------------
class Walker:
    """Evaluates the gadget schedule on concrete values and records every hot-path generator output in
    registration order.  ``cols`` is the flat Goldilocks column vector,
    ``ops`` the (kind, field, first_col, ncols, label) table that include/p2e.h's
    p2e_schedule_describe mirrors."""

    def __init__(self, curve=None):
        self.curve = curve or SECP256K1
        self.cols = []
        self.ops = []
        self._path = []
        # values of the targets that plonky2's BUILT-IN generators fill on the same
        # path (bool selects, window bits/digits, random-access selections, is_equal / not results), in the
        # order the gadgets create them.  Separate vector: the hot-path columns above are unchanged.
        self.aux = []
        self.aux_ops = []

    # -- bookkeeping
    def _rec(self, kind, field, vals):
        self.ops.append((kind, field, len(self.cols), len(vals), "/".join(self._path)))
        self.cols.extend(int(v) for v in vals)

    def _aux(self, kind, vals):
        self.aux_ops.append((kind, len(self.aux), len(vals), "/".join(self._path)))
        self.aux.extend(int(v) for v in vals)

    def not_(self, b):                                          # plonky2 builder.not: 1 - b
        v = 1 - b
        self._aux("not", [v])
        return v

    def is_equal_zero(self, x):                                 # builder.is_equal(x, zero) result
        v = int(x == 0)
        self._aux("is_zero", [v])
        return v

    class _Scope:
        def __init__(self, w, name):
            self.w, self.name = w, name

        def __enter__(self):
            self.w._path.append(self.name)

        def __exit__(self, *a):
            self.w._path.pop()

    def scope(self, name):
        return Walker._Scope(self, name)

    # -- gadgets/nonnative.rs
    def add_nonnative(self, a, b, field):                       # :245-276
        s, ov = gen_add(a, b, self.curve.moduli[field])
        self._rec("add", field, s + [ov])
        return s

    def sub_nonnative(self, a, b, field):                       # :356-388
        d, ov = gen_sub(a, b, self.curve.moduli[field])
        self._rec("sub", field, d + [ov])
        return d

    def add_many_nonnative(self, xs, field):                    # :310-353
        if len(xs) == 1:
            return xs[0]
        s, ov = gen_add_many(xs, self.curve.moduli[field])
        self._rec("add_many", field, s + [ov])
        return s

    def mul_nonnative(self, x, y, field):                       # :390-464
        m = self.curve.moduli[field]
        xw = list(x) + [0] * (NL - len(x))
        yw = list(y) + [0] * (NL - len(y))
        r, q, cs = gen_mul(xw, yw, m)
        b = gen_checksum(cs)
        assert check_mul_gate(xw, yw, r, q, cs, m) and check_checksum_gate(cs, b)
        self._rec("mul", field, r + q + cs + b)
        return r

    def inv_nonnative(self, x, field):                          # :502-536
        inv, div = gen_inv(x, self.curve.moduli[field])
        self._rec("inv", field, inv + div)
        return inv

    def neg_nonnative(self, x, field):                          # :491-500 (zero constant has 0 limbs)
        return self.sub_nonnative(const_limbs(0), x, field)

    def mul_by_bool(self, a, b):                                # gadgets/biguint.rs:360-374: one mul per limb
        v = [gl(l * b) for l in a]
        self._aux("mul_by_bool", v)
        return v

    def nonnative_conditional_neg(self, x, b, field):           # :584-596
        not_b = self.not_(b)
        neg = self.neg_nonnative(x, field)
        t = self.mul_by_bool(neg, b)
        f = self.mul_by_bool(x, not_b)
        return self.add_nonnative(t, f, field)

    # -- gadgets/curve.rs
    def curve_assert_valid(self, p):                            # :123-135
        x, y = p
        a = const_limbs(self.curve.a)
        b = const_limbs(self.curve.b)
        y2 = self.mul_nonnative(y, y, FIELD_BASE)
        x2 = self.mul_nonnative(x, x, FIELD_BASE)
        x3 = self.mul_nonnative(x2, x, FIELD_BASE)
        ax = self.mul_nonnative(a, x, FIELD_BASE)
        axb = self.add_nonnative(ax, b, FIELD_BASE)
        rhs = self.add_nonnative(x3, axb, FIELD_BASE)
        return y2 == rhs                                        # connect_nonnative

    def curve_conditional_neg(self, p, b):                      # :149-158
        return p[0], self.nonnative_conditional_neg(p[1], b, FIELD_BASE)

    def curve_double(self, p):                                  # :160-185
        x, y = p
        F = FIELD_BASE
        dy = self.add_nonnative(y, y, F)
        idy = self.inv_nonnative(dy, F)
        xx = self.mul_nonnative(x, x, F)
        t = self.add_many_nonnative([xx, xx, xx, const_limbs(self.curve.a)], F)
        lam = self.mul_nonnative(t, idy, F)
        lam2 = self.mul_nonnative(lam, lam, F)
        xd = self.add_nonnative(x, x, F)
        x3 = self.sub_nonnative(lam2, xd, F)
        xdf = self.sub_nonnative(x, x3, F)
        lx = self.mul_nonnative(lam, xdf, F)
        y3 = self.sub_nonnative(lx, y, F)
        return x3, y3

    def curve_repeated_double(self, p, n):                      # :187-200
        for _ in range(n):
            p = self.curve_double(p)
        return p

    def curve_add(self, p1, p2):                                # :202-223
        (x1, y1), (x2, y2) = p1, p2
        F = FIELD_BASE
        u = self.sub_nonnative(y2, y1, F)
        v = self.sub_nonnative(x2, x1, F)
        vinv = self.inv_nonnative(v, F)
        s = self.mul_nonnative(u, vinv, F)
        s2 = self.mul_nonnative(s, s, F)
        xs = self.add_nonnative(x2, x1, F)
        x3 = self.sub_nonnative(s2, xs, F)
        xd = self.sub_nonnative(x1, x3, F)
        pr = self.mul_nonnative(s, xd, F)
        y3 = self.sub_nonnative(pr, y1, F)
        return x3, y3

    def curve_conditional_add(self, p1, p2, b):                 # :225-243 (Q7: add always computed)
        not_b = self.not_(b)
        s = self.curve_add(p1, p2)
        xt = self.mul_by_bool(s[0], b)
        yt = self.mul_by_bool(s[1], b)
        xf = self.mul_by_bool(p1[0], not_b)
        yf = self.mul_by_bool(p1[1], not_b)
        x = self.add_nonnative(xt, xf, FIELD_BASE)
        y = self.add_nonnative(yt, yf, FIELD_BASE)
        return x, y

    # -- gadgets/split_nonnative.rs:25-72 (bits of each 29-bit limb, LE, limb-major, zero padded)
    @staticmethod
    def _bits(limbs):
        bits = []
        for l in limbs:
            if l >> BITS:
                raise RefPanic(ERR_LIMB_RANGE, "split_le_base: limb >= 2^29")
            bits.extend((l >> i) & 1 for i in range(BITS))
        return bits

    def split_4(self, limbs):                                   # :25-50
        bits = self._bits(limbs)
        self._aux("bits", bits)                                 # split_le_base::<2>(limb, 29) per limb
        while len(bits) % 4:
            bits.append(0)                                      # builder.zero(): a constant, not a witness value
        out, comb = [], []
        for i in range(0, len(bits), 4):
            lower = bits[i] + 2 * bits[i + 1]                   # mul_add(b, two, a)
            upper = bits[i + 2] + 2 * bits[i + 3]               # mul_add(d, two, c)
            limb = lower + 4 * upper                            # mul_add(upper, four, lower)
            comb += [lower, upper, limb]
            out.append(limb)
        self._aux("comb4", comb)
        return out

    def split_2(self, limbs):                                   # :52-72
        bits = self._bits(limbs)
        self._aux("bits", bits)
        while len(bits) % 2:
            bits.append(0)
        out = [bits[i] + 2 * bits[i + 1] for i in range(0, len(bits), 2)]
        self._aux("comb2", out)
        return out

    def random_access_point(self, idx, table):                  # gadgets/curve_windowed_mul.rs:74-118
        x, y = table[idx]
        x, y = list(x) + [0] * (NL - len(x)), list(y) + [0] * (NL - len(y))
        self._aux("random_access", x + y)                       # selected x limbs, then selected y limbs
        return x, y

    # -- gadgets/curve_fixed_base.rs:18-66
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
                muls = C.fixed_base_window(point)               # t*P_i, t=1..15 ; slot 0 := slot 1 (Q8')
                tbl = [const_point(muls[0])] + [const_point(q) for q in muls]
                should_add = self.not_(self.is_equal_zero(limb))
                r = self.random_access_point(limb, tbl)
                result = self.curve_conditional_add(result, r, should_add)
            for _ in range(4):
                point = C.double(point)
        with self.scope("unblind"):
            return self.curve_add(result, const_point(C.neg(rando)))

    # -- gadgets/curve_msm.rs:21-79
    def curve_msm(self, p, q, n, m):
        limbs_n = self.split_2(n)
        limbs_m = self.split_2(m)
        assert len(limbs_n) == len(limbs_m)
        num = len(limbs_n)
        rando = rando_point()
        rando_t = const_point(rando)
        neg_rando = const_point(ec_neg(rando))
        pre = [p] * 16
        cur_p, cur_q = rando_t, rando_t
        with self.scope("table"):
            for i in range(4):
                pre[i] = cur_p
                pre[4 * i] = cur_q
                cur_p = self.curve_add(cur_p, p)
                cur_q = self.curve_add(cur_q, q)
            for i in range(1, 4):
                pre[i] = self.curve_add(pre[i], neg_rando)
                pre[4 * i] = self.curve_add(pre[4 * i], neg_rando)
            for i in range(1, 4):
                for j in range(1, 4):
                    pre[i + 4 * j] = self.curve_add(pre[i], pre[4 * j])
        result = rando_t
        for d in reversed(range(num)):                          # Q9: MSB first
            with self.scope(f"digit{d}"):
                result = self.curve_repeated_double(result, 2)
                idx = 4 * limbs_m[d] + limbs_n[d]               # mul_add(four, limb_m, limb_n)
                self._aux("index", [idx])
                r = self.random_access_point(idx, pre)
                should_add = self.not_(self.is_equal_zero(idx))
                result = self.curve_conditional_add(result, r, should_add)
        spm = rando
        for _ in range(2 * num):
            spm = ec_double(spm)
        with self.scope("unblind"):
            return self.curve_add(result, const_point(ec_neg(spm)))

    # -- gadgets/curve.rs:137-147
    def curve_neg(self, p):
        return p[0], self.neg_nonnative(p[1], FIELD_BASE)

    # -- gadgets/curve_windowed_mul.rs:52-72.  `g` = the point precompute_window draws with rand() at circuit-build
    # time (:57): an input of the restatement
    def precompute_window(self, p, g):
        C = self.curve
        neg = const_point(C.neg(g))
        multiples = [const_point(g)]
        for i in range(1, 16):
            multiples.append(self.curve_add(p, multiples[i - 1]))
        for i in range(1, 16):
            multiples[i] = self.curve_add(neg, multiples[i])
        return multiples

    # -- gadgets/curve_windowed_mul.rs:131-173
    def curve_scalar_mul_windowed(self, p, n, g):
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
                result = self.curve_repeated_double(result, 4)
                to_add = self.random_access_point(windows[i], pre)
                should_add = self.not_(self.is_equal_zero(windows[i]))
                result = self.curve_conditional_add(result, to_add, should_add)
        with self.scope("unblind"):
            to_add = self.curve_neg(const_point(spm))
            return self.curve_add(result, to_add)

    # -- gadgets/curve.rs:245-285.  `rando` = the rand() blinding point of :253, an input as above
    def curve_scalar_mul(self, p, n, rando):
        bits = self._bits(n)                                    # split_nonnative_to_bits gadgets/nonnative.rs:566-582
        self._aux("bits", bits)
        randot = const_point(rando)
        result = (pad9(randot[0]), pad9(randot[1]))             # add_virtual_affine_point_target + connect_affine_point
        two_i_times_p = p
        for i, bit in enumerate(bits):
            with self.scope(f"bit{i}"):
                not_bit = self.not_(bit)
                rp = self.curve_add(result, two_i_times_p)
                xt = self.mul_by_bool(rp[0], bit)
                xf = self.mul_by_bool(result[0], not_bit)
                yt = self.mul_by_bool(rp[1], bit)
                yf = self.mul_by_bool(result[1], not_bit)
                new_x = self.add_nonnative(xt, xf, FIELD_BASE)
                new_y = self.add_nonnative(yt, yf, FIELD_BASE)
                result = (new_x, new_y)
                two_i_times_p = self.curve_double(two_i_times_p)
        with self.scope("unblind"):
            neg_r = self.curve_neg(randot)
            return self.curve_add(result, neg_r)

    # -- gadgets/ecdsa.rs:55-78
    def verify_p256_message(self, msg, r, s, pk, g):
        C = self.curve
        with self.scope("assert_valid"):
            ok_curve = self.curve_assert_valid(pk)
        c = self.inv_nonnative(s, FIELD_SCALAR)
        u1 = self.mul_nonnative(msg, c, FIELD_SCALAR)
        u2 = self.mul_nonnative(r, c, FIELD_SCALAR)
        with self.scope("fixed_base"):
            point1 = self.fixed_base_curve_mul(C.g, u1)
        with self.scope("windowed_mul"):
            point2 = self.curve_scalar_mul_windowed(pk, u2, g)
        with self.scope("final_add"):
            point = self.curve_add(point1, point2)
        ok_sig = pad9(point[0]) == pad9(r)
        return ok_curve and ok_sig

    # -- gadgets/glv.rs:53-104
    def decompose_secp256k1_scalar(self, k):
        k1, k2, n1, n2 = gen_glv(k)
        self._rec("glv", FIELD_SCALAR, k1 + k2 + [n1, n2])
        S = FIELD_SCALAR
        k1_raw = self.nonnative_conditional_neg(k1, n1, S)
        k2_raw = self.nonnative_conditional_neg(k2, n2, S)
        sb = self.mul_nonnative(const_limbs(GLV_S), k2_raw, S)
        sb = self.add_nonnative(sb, k1_raw, S)
        ok = pad9(sb) == pad9(k)                                # connect_nonnative
        return k1, k2, n1, n2, ok

    def glv_mul(self, p, k):
        with self.scope("decompose"):
            k1, k2, n1, n2, ok = self.decompose_secp256k1_scalar(k)
        beta_px = self.mul_nonnative(const_limbs(GLV_BETA), p[0], FIELD_BASE)
        sp = (beta_px, p[1])
        p_neg = self.curve_conditional_neg(p, n1)
        sp_neg = self.curve_conditional_neg(sp, n2)
        with self.scope("msm"):
            return self.curve_msm(p_neg, sp_neg, k1, k2), ok

    # -- gadgets/ecdsa.rs:30-53
    def verify_secp256k1_message(self, msg, r, s, pk):
        """msg, r, s, pk.x, pk.y are 9-limb targets (virtual, set through pw.set_biguint_target)."""
        with self.scope("assert_valid"):
            ok_curve = self.curve_assert_valid(pk)
        c = self.inv_nonnative(s, FIELD_SCALAR)
        u1 = self.mul_nonnative(msg, c, FIELD_SCALAR)
        u2 = self.mul_nonnative(r, c, FIELD_SCALAR)
        with self.scope("fixed_base"):
            point1 = self.fixed_base_curve_mul(G, u1)
        with self.scope("glv_mul"):
            point2, ok_glv = self.glv_mul(pk, u2)
        with self.scope("final_add"):
            point = self.curve_add(point1, point2)
        ok_sig = pad9(point[0]) == pad9(r)
        return ok_curve and ok_glv and ok_sig


def pad9(l):
    return list(l) + [0] * (NL - len(l))


def const_point(pt):
    """constant_affine_point gadgets/curve.rs:99-105 -> constant limb lists (variable length, Q5)."""
    return const_limbs(pt[0]), const_limbs(pt[1])


_FB_CACHE = {}


def fixed_base_window(point):
    """t * point for t = 1..15 (gadgets/curve_fixed_base.rs:45-54)."""
    if point not in _FB_CACHE:
        out, acc = [], None
        for _ in range(15):
            acc = ec_add(point, acc)
            out.append(acc)
        _FB_CACHE[point] = out
    return _FB_CACHE[point]


# ----------------------------------------------------------------------------------------------
# convenience entry points
# ----------------------------------------------------------------------------------------------
NUM_VERIFY_COLS = 82615
NUM_VERIFY_AUX = 8959      # 459 (4-bit split) + 66*57 (fixed-base windows) + 30 + 38 + 436 + 73*58 (glv_mul)
NUM_GLV_MUL_AUX = 4738     # the glv_mul part alone


def verify_witness(msg, r, s, pkx, pky):
    """Full column vector of one verify_secp256k1_message_circuit instance. Returns (cols, ok, ops)."""
    w = Walker()
    ok = w.verify_secp256k1_message(limbs_of(msg, NL), limbs_of(r, NL), limbs_of(s, NL),
                                    (limbs_of(pkx, NL), limbs_of(pky, NL)))
    assert len(w.cols) == NUM_VERIFY_COLS, len(w.cols)
    return w.cols, ok, w.ops


def verify_witness_aux(msg, r, s, pkx, pky):
    """(cols, aux, ok, aux_ops): the hot-path columns plus the built-in-generator values (Walker.aux)."""
    w = Walker()
    ok = w.verify_secp256k1_message(limbs_of(msg, NL), limbs_of(r, NL), limbs_of(s, NL),
                                    (limbs_of(pkx, NL), limbs_of(pky, NL)))
    assert len(w.aux) == NUM_VERIFY_AUX, len(w.aux)
    return w.cols, w.aux, ok, w.aux_ops


def glv_mul_witness(pkx, pky, k):
    w = Walker()
    (_pt, ok) = w.glv_mul((limbs_of(pkx, NL), limbs_of(pky, NL)), limbs_of(k, NL))
    return w.cols, ok, w.ops


def glv_mul_witness_aux(pkx, pky, k):
    w = Walker()
    (_pt, ok) = w.glv_mul((limbs_of(pkx, NL), limbs_of(pky, NL)), limbs_of(k, NL))
    assert len(w.aux) == NUM_GLV_MUL_AUX, len(w.aux)
    return w.cols, w.aux, ok, w.aux_ops


def windowed_mul_witness(curve, px, py, k, g):
    """curve_scalar_mul_windowed(p, k) with precompute_window's random point `g`: (cols, aux, ops, result point)"""
    w = Walker(curve)
    pt = w.curve_scalar_mul_windowed((limbs_of(px, NL), limbs_of(py, NL)), limbs_of(k, NL), g)
    return w.cols, w.aux, w.ops, (value_of(pt[0]), value_of(pt[1]))


def scalar_mul_witness(curve, px, py, k, rando):
    """curve_scalar_mul(p, k) with blinding point `rando`"""
    w = Walker(curve)
    pt = w.curve_scalar_mul((limbs_of(px, NL), limbs_of(py, NL)), limbs_of(k, NL), rando)
    return w.cols, w.aux, w.ops, (value_of(pt[0]), value_of(pt[1]))


def verify_p256_witness(msg, r, s, pkx, pky, g):
    """verify_p256_message_circuit: (cols, aux, ok, ops)"""
    w = Walker(P256)
    ok = w.verify_p256_message(limbs_of(msg, NL), limbs_of(r, NL), limbs_of(s, NL), (limbs_of(pkx, NL), limbs_of(pky, NL)), g)
    return w.cols, w.aux, ok, w.ops


def synth_signature_curve(curve, rng):
    """sign_message curve/ecdsa.rs:25-40 on any curve of the crate"""
    while True:
        sk, msg, k = rng.below(curve.n), rng.below(curve.n), rng.below(curve.n)
        pk = curve.mul(sk, curve.g)
        rr = curve.mul(k, curve.g)
        r = canon(rr[0], curve.n)
        s = pow(k, -1, curve.n) * (msg + r * sk) % curve.n
        if r == 0 or s == 0:
            continue
        return msg, r, s, pk[0], pk[1]


# deterministic synthetic inputs (splitmix64), restating curve/ecdsa.rs:25-40 sign_message
class SplitMix64:
    def __init__(self, seed):
        self.s = seed & _M64

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & _M64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        return z ^ (z >> 31)

    def below(self, m):
        while True:
            v = _u64le([self.next() for _ in range(4)])
            if 0 < v < m:
                return v


def synth_signature_at(seed, index):
    """Signature `index` of the synthetic batch `seed` (same stream as p2e_synth_signatures)."""
    return synth_signature(SplitMix64((seed ^ (0x9E3779B97F4A7C15 * (index + 1))) & _M64))


def synth_signature(rng):
    """(msg, r, s, pkx, pky) of a valid signature; sk, msg, nonce uniform in [1, n)."""
    while True:
        sk = rng.below(N)
        msg = rng.below(N)
        k = rng.below(N)
        pk = ec_mul(sk, G)
        rr = ec_mul(k, G)
        if rr[0] == 0:
            continue
        r = canon(rr[0], N)          # base_to_scalar curve/curve_types.rs:280-282 (no reduction needed beyond canon)
        s = pow(k, -1, N) * (msg + r * sk) % N
        if r == 0 or s == 0:
            continue
        return msg, r, s, pk[0], pk[1]
