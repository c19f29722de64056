/* C restatement of the plonky2-ecdsa hot-path witness generators -- TEST INFRASTRUCTURE ONLY.
 * See p2e_oracle.h for the status header ("parity unpinned" vs literal reference outputs; pinned by
 * the Python big-int restatement + the reference's constraint equations).
 *
 * Deliberately algorithm-faithful to the reference's CPU path and deliberately DIFFERENT from the
 * product's HIP code: base-2^32 schoolbook multiplication + Knuth algorithm D division (what
 * num::BigUint does), one Fermat exponentiation per inverse (plonky2 Secp256K1*::try_inverse =
 * exp_biguint(order-2) [upstream-from-memory; in-tree template field/p256_base.rs:112-119]), affine
 * incomplete curve formulas, one signature after another.  File:line citations are relative to
 * /root/reference/src.
 */
#include "p2e_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <ucontext.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint32_t w32;
typedef uint64_t u64;
typedef unsigned __int128 u128;

#define NL 9
#define BITS 29
#define MASK29 ((1u << BITS) - 1)
#define P_GL 0xFFFFFFFF00000001ull

/* ------------------------------------------------------------------------------------------ */
/* base-2^32 multi-precision helpers                                                            */
/* ------------------------------------------------------------------------------------------ */
static int bn_cmp(const w32 *a, const w32 *b, int n) {
    for (int i = n - 1; i >= 0; i--) {
        if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    }
    return 0;
}
static int bn_is_zero(const w32 *a, int n) {
    for (int i = 0; i < n; i++)
        if (a[i]) return 0;
    return 1;
}
static w32 bn_add(w32 *r, const w32 *a, const w32 *b, int n) {
    u64 c = 0;
    for (int i = 0; i < n; i++) {
        c += (u64)a[i] + b[i];
        r[i] = (w32)c;
        c >>= 32;
    }
    return (w32)c;
}
static w32 bn_sub(w32 *r, const w32 *a, const w32 *b, int n) {
    u64 br = 0;
    for (int i = 0; i < n; i++) {
        u64 d = (u64)a[i] - b[i] - br;
        r[i] = (w32)d;
        br = (d >> 63) & 1;
    }
    return (w32)br;
}
static void bn_mul(const w32 *a, int an, const w32 *b, int bn, w32 *r) {
    memset(r, 0, sizeof(w32) * (size_t)(an + bn));
    for (int i = 0; i < an; i++) {
        u64 c = 0;
        for (int j = 0; j < bn; j++) {
            c += (u64)a[i] * b[j] + r[i + j];
            r[i + j] = (w32)c;
            c >>= 32;
        }
        r[i + bn] = (w32)c;
    }
}
/* Knuth TAOCP vol.2 4.3.1 algorithm D.  u: un words, v: vn words (v[vn-1] != 0, vn >= 2), un >= vn.
 * q: un-vn+1 words, r: vn words. */
static void bn_divrem(const w32 *u, int un, const w32 *v, int vn, w32 *q, w32 *r) {
    w32 un_[40], vn_[12];
    int s = __builtin_clz(v[vn - 1]);
    for (int i = vn - 1; i > 0; i--) vn_[i] = s ? (v[i] << s) | (v[i - 1] >> (32 - s)) : v[i];
    vn_[0] = v[0] << s;
    un_[un] = s ? u[un - 1] >> (32 - s) : 0;
    for (int i = un - 1; i > 0; i--) un_[i] = s ? (u[i] << s) | (u[i - 1] >> (32 - s)) : u[i];
    un_[0] = u[0] << s;
    for (int j = un - vn; j >= 0; j--) {
        u64 num = ((u64)un_[j + vn] << 32) | un_[j + vn - 1];
        u64 qhat = num / vn_[vn - 1];
        u64 rhat = num % vn_[vn - 1];
        while (qhat >= (1ull << 32) || qhat * vn_[vn - 2] > ((rhat << 32) | un_[j + vn - 2])) {
            qhat--;
            rhat += vn_[vn - 1];
            if (rhat >= (1ull << 32)) break;
        }
        /* multiply and subtract */
        int64_t borrow = 0;
        u64 carry = 0;
        for (int i = 0; i < vn; i++) {
            u64 p = qhat * vn_[i] + carry;
            carry = p >> 32;
            int64_t t = (int64_t)un_[i + j] - borrow - (int64_t)(p & 0xFFFFFFFFull);
            un_[i + j] = (w32)t;
            borrow = t < 0 ? 1 : 0;
        }
        int64_t t = (int64_t)un_[j + vn] - borrow - (int64_t)carry;
        un_[j + vn] = (w32)t;
        q[j] = (w32)qhat;
        if (t < 0) { /* add back */
            q[j]--;
            u64 c = 0;
            for (int i = 0; i < vn; i++) {
                c += (u64)un_[i + j] + vn_[i];
                un_[i + j] = (w32)c;
                c >>= 32;
            }
            un_[j + vn] += (w32)c;
        }
    }
    for (int i = 0; i < vn; i++) r[i] = s ? (un_[i] >> s) | ((u64)un_[i + 1] << (32 - s)) : un_[i];
}

/* ------------------------------------------------------------------------------------------ */
/* moduli and constants (curve/secp256k1.rs:15-38, curve/glv.rs:11-32)                          */
/* ------------------------------------------------------------------------------------------ */
static const w32 MOD_P[8] = {0xFFFFFC2F, 0xFFFFFFFE, 0xFFFFFFFF, 0xFFFFFFFF,
                             0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF};
static const w32 MOD_N[8] = {0xD0364141, 0xBFD25E8C, 0xAF48A03B, 0xBAAEDCE6,
                             0xFFFFFFFE, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF};
#define W64(x) (w32)((x) & 0xFFFFFFFFull), (w32)((x) >> 32)
static const w32 GEN_X[8] = {W64(0x59F2815B16F81798ull), W64(0x029BFCDB2DCE28D9ull), W64(0x55A06295CE870B07ull),
                             W64(0x79BE667EF9DCBBACull)};
static const w32 GEN_Y[8] = {W64(0x9C47D08FFB10D4B8ull), W64(0xFD17B448A6855419ull), W64(0x5DA4FBFC0E1108A8ull),
                             W64(0x483ADA7726A3C465ull)};
static const w32 GLV_BETA[8] = {W64(13923278643952681454ull), W64(11308619431505398165ull),
                                W64(7954561588662645993ull), W64(8856726876819556112ull)};
static const w32 GLV_S[8] = {W64(16069571880186789234ull), W64(1310022930574435960ull),
                             W64(11900229862571533402ull), W64(6008836872998760672ull)};
static const w32 GLV_A1[8] = {W64(16747920425669159701ull), W64(3496713202691238861ull), 0, 0, 0, 0};
static const w32 GLV_MINUS_B1[8] = {W64(8022177200260244675ull), W64(16448129721693014056ull), 0, 0, 0, 0};
static const w32 GLV_A2[8] = {W64(6323353552219852760ull), W64(1498098850674701302ull), 1, 0, 0, 0};
static const w32 GLV_B2[8] = {W64(16747920425669159701ull), W64(3496713202691238861ull), 0, 0, 0, 0};
/* keccak256 of eight zero bytes, read as a little-endian integer (gadgets/curve_fixed_base.rs:34-37);
 * tests/ re-derive this digest with oracle/p2e_ref.py's Keccak. */
static const uint8_t HASH0_LE[32] = {0x01, 0x1b, 0x4d, 0x03, 0xdd, 0x8c, 0x01, 0xf1, 0x04, 0x91, 0x43,
                                     0xcf, 0x9c, 0x4c, 0x81, 0x7e, 0x4b, 0x16, 0x7f, 0x1d, 0x1b, 0x83,
                                     0xe5, 0xc6, 0xf0, 0xf1, 0x0d, 0x89, 0xba, 0x1e, 0x7b, 0xce};

/* NIST P-256: field/p256_base.rs:14-17, field/p256_scalar.rs:5-9, curve/p256.rs:15-57 */
static const w32 MOD_P256[8] = {0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0x00000000,
                                0x00000000, 0x00000000, 0x00000001, 0xFFFFFFFF};
static const w32 MOD_N256[8] = {0xFC632551, 0xF3B9CAC2, 0xA7179E84, 0xBCE6FAAD,
                                0xFFFFFFFF, 0xFFFFFFFF, 0x00000000, 0xFFFFFFFF};
static const w32 P256_A[8] = {W64(0xFFFFFFFFFFFFFFFCull), W64(0x00000000FFFFFFFFull), W64(0x0000000000000000ull),
                              W64(0xFFFFFFFF00000001ull)};
static const w32 P256_B[8] = {W64(0x3BCE3C3E27D2604Bull), W64(0x651D06B0CC53B0F6ull), W64(0xB3EBBD55769886BCull),
                              W64(0x5AC635D8AA3A93E7ull)};
static const w32 P256_GX[8] = {W64(0xF4A13945D898C296ull), W64(0x77037D812DEB33A0ull), W64(0xF8BCE6E563A440F2ull),
                               W64(0x6B17D1F2E12C4247ull)};
static const w32 P256_GY[8] = {W64(0xCBB6406837BF51F5ull), W64(0x2BCE33576B315ECEull), W64(0x8EE7EB4A7C0F9E16ull),
                               W64(0x4FE342E2FE1A7F9Bull)};
static const w32 SECP_A[8] = {0, 0, 0, 0, 0, 0, 0, 0};
static const w32 SECP_B[8] = {7, 0, 0, 0, 0, 0, 0, 0};

static const w32 *modulus_of(int field) {
    switch (field) {
    case P2E_O_FIELD_SCALAR: return MOD_N;
    case P2E_O_FIELD_P256_BASE: return MOD_P256;
    case P2E_O_FIELD_P256_SCALAR: return MOD_N256;
    default: return MOD_P;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* prime-field ops the way the reference does them: BigUint mul + mod_floor, Fermat inverse     */
/* ------------------------------------------------------------------------------------------ */
static void fe_mulmod(w32 *r, const w32 *a, const w32 *b, const w32 *m) {
    w32 prod[16], q[9];
    bn_mul(a, 8, b, 8, prod);
    bn_divrem(prod, 16, m, 8, q, r);
}
static void fe_addmod(w32 *r, const w32 *a, const w32 *b, const w32 *m) {
    w32 c = bn_add(r, a, b, 8);
    if (c || bn_cmp(r, m, 8) >= 0) bn_sub(r, r, m, 8);
}
static void fe_submod(w32 *r, const w32 *a, const w32 *b, const w32 *m) {
    if (bn_sub(r, a, b, 8)) bn_add(r, r, m, 8);
}
static void fe_powmod(w32 *r, const w32 *a, const w32 *e, const w32 *m) {
    w32 acc[8] = {1, 0, 0, 0, 0, 0, 0, 0};
    int started = 0;
    for (int i = 255; i >= 0; i--) {
        if (started) fe_mulmod(acc, acc, acc, m);
        if ((e[i >> 5] >> (i & 31)) & 1) {
            fe_mulmod(acc, acc, a, m);
            started = 1;
        }
    }
    memcpy(r, acc, 32);
}
static void fe_invmod(w32 *r, const w32 *a, const w32 *m) {
    w32 e[8], two[8] = {2, 0, 0, 0, 0, 0, 0, 0};
    bn_sub(e, m, two, 8);
    fe_powmod(r, a, e, m);
}
/* to_canonical_biguint: ONE conditional subtraction (template field/p256_base.rs:162-168) */
static void fe_canon(w32 *a, const w32 *m) {
    if (bn_cmp(a, m, 8) >= 0) bn_sub(a, a, m, 8);
}

/* ------------------------------------------------------------------------------------------ */
/* limb <-> word conversion (gadgets/biguint.rs:27-51 convert_base, :444-452 get_biguint_target) */
/* ------------------------------------------------------------------------------------------ */
/* sum limb_i * 2^(29 i) for arbitrary u64 limbs; returns 1 if the value does not fit nw words */
static int limbs_to_words(const u64 *limbs, int nl, w32 *out, int nw) {
    w32 acc[16];
    memset(acc, 0, sizeof acc);
    for (int i = 0; i < nl; i++) {
        int bit = BITS * i;
        int wi = bit >> 5, sh = bit & 31;
        u128 v = (u128)limbs[i] << sh; /* up to 96 bits */
        u64 c = 0;
        for (int k = 0; k < 3 || c; k++) {
            c += (u64)acc[wi + k] + (w32)(v & 0xFFFFFFFFu);
            acc[wi + k] = (w32)c;
            c >>= 32;
            v >>= 32;
            if (wi + k + 1 >= 16) break;
        }
    }
    int over = 0;
    for (int i = nw; i < 16; i++) over |= acc[i] != 0;
    memcpy(out, acc, sizeof(w32) * (size_t)nw);
    return over;
}
/* value -> nl 29-bit limbs; returns 1 if it does not fit */
static int words_to_limbs(const w32 *w, int nw, u64 *limbs, int nl) {
    int over = 0;
    int total_bits = nw * 32;
    for (int i = 0; i * BITS < total_bits || i < nl; i++) {
        int bit = BITS * i;
        u64 v = 0;
        if (bit < total_bits) {
            int wi = bit >> 5, sh = bit & 31;
            u64 lo = w[wi];
            u64 hi = wi + 1 < nw ? w[wi + 1] : 0;
            v = ((lo | (hi << 32)) >> sh) & MASK29;
        }
        if (i < nl)
            limbs[i] = v;
        else if (v)
            over = 1;
    }
    return over;
}
static void bytes_to_words(const uint8_t *b, w32 *w) {
    for (int i = 0; i < 8; i++)
        w[i] = (w32)b[4 * i] | ((w32)b[4 * i + 1] << 8) | ((w32)b[4 * i + 2] << 16) | ((w32)b[4 * i + 3] << 24);
}
static void words_to_bytes(const w32 *w, uint8_t *b) {
    for (int i = 0; i < 8; i++) {
        b[4 * i] = (uint8_t)w[i];
        b[4 * i + 1] = (uint8_t)(w[i] >> 8);
        b[4 * i + 2] = (uint8_t)(w[i] >> 16);
        b[4 * i + 3] = (uint8_t)(w[i] >> 24);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Goldilocks                                                                                   */
/* ------------------------------------------------------------------------------------------ */
static u64 gl_reduce128(u128 x) {
    u64 lo = (u64)x, hi = (u64)(x >> 64);
    u64 hh = hi >> 32, hl = hi & 0xFFFFFFFFull;
    /* x = lo + hl*2^64 + hh*2^96 ;  2^64 = 2^32-1, 2^96 = -1  (mod p) */
    u128 t = (u128)lo + (u128)hl * 0xFFFFFFFFull + (u128)P_GL - hh;
    u64 tl = (u64)t, th = (u64)(t >> 64); /* th <= 1 */
    u128 t2 = (u128)tl + (u128)th * 0xFFFFFFFFull;
    u64 r = (u64)t2;
    if ((u64)(t2 >> 64)) r += 0xFFFFFFFFull; /* cannot overflow again */
    if (r >= P_GL) r -= P_GL;
    return r;
}
static u64 gl_mul(u64 a, u64 b) { return gl_reduce128((u128)a * b); }
static u64 gl_add(u64 a, u64 b) {
    u128 s = (u128)a + b;
    if (s >= P_GL) s -= P_GL;
    return (u64)s;
}
static u64 gl_sub(u64 a, u64 b) { return a >= b ? a - b : a + (P_GL - b); }
static u64 gl_pow(u64 a, u64 e) {
    u64 r = 1;
    while (e) {
        if (e & 1) r = gl_mul(r, a);
        a = gl_mul(a, a);
        e >>= 1;
    }
    return r;
}

/* ------------------------------------------------------------------------------------------ */
/* the generators, one element at a time                                                        */
/* ------------------------------------------------------------------------------------------ */
/* canonical value of a limb vector: from_noncanonical_biguint (err >= 2^256) + to_canonical */
static int load_canon(const u64 *limbs, int nl, const w32 *m, w32 *out, uint8_t *err) {
    if (limbs_to_words(limbs, nl, out, 8)) {
        *err |= P2E_O_ERR_VALUE_GE_2_256;
        memset(out, 0, 32);
        return 1;
    }
    fe_canon(out, m);
    return 0;
}

/* gadgets/nonnative.rs:626-645 */
static void gen_add(const u64 *a, int na, const u64 *b, int nb, const w32 *m, u64 *sum, u64 *ov, uint8_t *err) {
    w32 av[8], bv[8], s[9], m9[9];
    int bad = load_canon(a, na, m, av, err) | load_canon(b, nb, m, bv, err);
    s[8] = bn_add(s, av, bv, 8);
    memcpy(m9, m, 32);
    m9[8] = 0;
    if (bn_cmp(s, m9, 9) > 0) { /* Q1: strict > */
        bn_sub(s, s, m9, 9);
        *ov = 1;
    } else
        *ov = 0;
    words_to_limbs(s, 9, sum, NL);
    if (bad) {
        memset(sum, 0, sizeof(u64) * NL);
        *ov = 0;
    }
}
/* gadgets/nonnative.rs:792-810 */
static void gen_sub(const u64 *a, int na, const u64 *b, int nb, const w32 *m, u64 *diff, u64 *ov, uint8_t *err) {
    w32 av[8], bv[8], d[8];
    int bad = load_canon(a, na, m, av, err) | load_canon(b, nb, m, bv, err);
    if (bn_cmp(av, bv, 8) >= 0) {
        bn_sub(d, av, bv, 8);
        *ov = 0;
    } else {
        w32 t[8];
        bn_sub(t, bv, av, 8);
        bn_sub(d, m, t, 8);
        *ov = 1;
    }
    words_to_limbs(d, 8, diff, NL);
    if (bad) {
        memset(diff, 0, sizeof(u64) * NL);
        *ov = 0;
    }
}
/* gadgets/nonnative.rs:696-728; summands[k] each with up to 9 limbs */
static void gen_add_many(const u64 (*summands)[NL], const int *nls, int k, const w32 *m, u64 *sum, u64 *ov,
                         uint8_t *err) {
    w32 acc[12], q[5], r[8];
    memset(acc, 0, sizeof acc);
    int bad = 0;
    for (int i = 0; i < k; i++) {
        w32 v[12];
        memset(v, 0, sizeof v);
        bad |= load_canon(summands[i], nls[i], m, v, err);
        bn_add(acc, acc, v, 12);
    }
    bn_divrem(acc, 12, m, 8, q, r);
    *ov = q[0]; /* to_u64_digits()[0] as u32 */
    words_to_limbs(r, 8, sum, NL);
    if (bad) {
        memset(sum, 0, sizeof(u64) * NL);
        *ov = 0;
    }
}
/* Optimised-CPU baseline variant (variant b): G signatures are walked in lock step, one
 * coroutine each, and every NonNativeInverseGenerator's FF::inverse() of the group is answered by ONE Fermat ladder
 * plus 3 (G - 1) multiplications (Montgomery's trick across signatures at the same schedule step).  Everything else
 * is the faithful-cost algorithm above.  Non-NULL only inside p2e_oracle_verify_witness_lockstep. */
struct lockstep;
static __thread struct lockstep *LS = NULL;
static void ls_invmod(w32 *r, const w32 *a, const w32 *m);

/* gadgets/nonnative.rs:857-872; outputs have k = nl limbs */
static void gen_inv(const u64 *x, int nl, const w32 *m, u64 *inv, u64 *div, uint8_t *err) {
    w32 xv[8], iv[8], prod[16], q[9], r[8];
    memset(inv, 0, sizeof(u64) * (size_t)nl);
    memset(div, 0, sizeof(u64) * (size_t)nl);
    if (load_canon(x, nl, m, xv, err)) return;
    if (bn_is_zero(xv, 8)) {
        *err |= P2E_O_ERR_INVERSE_OF_ZERO;
        return;
    }
    if (LS)
        ls_invmod(iv, xv, m); /* optimised-CPU variant: batched with the other signatures of the lock-step group */
    else
        fe_invmod(iv, xv, m);
    bn_mul(xv, 8, iv, 8, prod);
    bn_divrem(prod, 16, m, 8, q, r);
    if (words_to_limbs(iv, 8, inv, nl) | words_to_limbs(q, 9, div, nl)) *err |= P2E_O_ERR_LIMB_RANGE;
}
/* gates/mul_nonnative.rs:513-531 */
static u64 INV_2_29;
static void gen_checksum(const u64 *a, u64 *b, uint8_t *err) {
    u64 last = 0;
    for (int i = 0; i < 2 * NL - 2; i++) {
        u64 bi = gl_mul(gl_add(a[i], last), INV_2_29);
        u64 v = gl_add(bi, 1ull << 33);
        b[i] = v;
        last = bi;
        if (v >= (1ull << 34)) *err |= P2E_O_ERR_CARRY_RANGE;
    }
}
/* gates/mul_nonnative.rs:249-324; x, y are the 9 gate wires */
static void gen_mul(const u64 *x, const u64 *y, const w32 *m, u64 *r29, u64 *q29, u64 *cs, uint8_t *err) {
    w32 xv[9], yv[9], prod[18], q[11], r[8];
    u64 m29[NL], q_full[13];
    for (int i = 0; i < NL; i++) {
        if (x[i] >> BITS || y[i] >> BITS) {
            *err |= P2E_O_ERR_LIMB_RANGE;
            memset(r29, 0, sizeof(u64) * NL);
            memset(q29, 0, sizeof(u64) * NL);
            memset(cs, 0, sizeof(u64) * (2 * NL - 1));
            return;
        }
    }
    limbs_to_words(x, NL, xv, 9);
    limbs_to_words(y, NL, yv, 9);
    bn_mul(xv, 9, yv, 9, prod);
    bn_divrem(prod, 18, m, 8, q, r);
    words_to_limbs(m, 8, m29, NL);
    words_to_limbs(r, 8, r29, NL);
    words_to_limbs(q, 11, q_full, 13);
    for (int i = NL; i < 13; i++)
        if (q_full[i]) *err |= P2E_O_ERR_QUOTIENT_RANGE; /* q does not fit the gate's 9 q wires */
    memcpy(q29, q_full, sizeof(u64) * NL);
    for (int i = 0; i < 2 * NL - 1; i++) {
        int lo = i - NL + 1 > 0 ? i - NL + 1 : 0;
        int hi = i + 1 < NL ? i + 1 : NL;
        u64 acc = 0;
        for (int j = lo; j < hi; j++) {
            acc = gl_add(acc, gl_sub(gl_mul(q29[i - j], m29[j]), gl_mul(x[j], y[i - j])));
        }
        if (i < NL) acc = gl_add(acc, r29[i]);
        cs[i] = acc;
    }
}
/* curve/glv.rs:39-77 + gadgets/glv.rs:128-142 */
static void glv_round_div(const w32 *c, const w32 *k, w32 *out /*8*/) {
    /* round(c*k / n), num Ratio::round: n odd -> up iff 2*rem > n */
    w32 prod[16], q[9], r[9], r2[9], n9[9];
    bn_mul(c, 8, k, 8, prod);
    bn_divrem(prod, 16, MOD_N, 8, q, r);
    r[8] = 0;
    memcpy(n9, MOD_N, 32);
    n9[8] = 0;
    bn_add(r2, r, r, 9);
    if (bn_cmp(r2, n9, 9) > 0) {
        w32 one[9] = {1, 0, 0, 0, 0, 0, 0, 0, 0};
        bn_add(q, q, one, 9);
    }
    memcpy(out, q, 32); /* < 2^130 */
}
static void gen_glv(const u64 *k_limbs, int nl, u64 *k1l, u64 *k2l, u64 *k1_neg, u64 *k2_neg, uint8_t *err) {
    w32 k[8], c1[8], c2[8], t1[8], t2[8], k1[8], k2[8], half[8];
    memset(k1l, 0, sizeof(u64) * 5);
    memset(k2l, 0, sizeof(u64) * 5);
    *k1_neg = *k2_neg = 0;
    if (load_canon(k_limbs, nl, MOD_N, k, err)) return;
    glv_round_div(GLV_B2, k, c1);
    glv_round_div(GLV_MINUS_B1, k, c2);
    fe_canon(c1, MOD_N);
    fe_canon(c2, MOD_N);
    fe_mulmod(t1, c1, GLV_A1, MOD_N);
    fe_mulmod(t2, c2, GLV_A2, MOD_N);
    fe_submod(k1, k, t1, MOD_N);
    fe_submod(k1, k1, t2, MOD_N);
    fe_mulmod(t1, c1, GLV_MINUS_B1, MOD_N);
    fe_mulmod(t2, c2, GLV_B2, MOD_N);
    fe_submod(k2, t1, t2, MOD_N);
    for (int i = 0; i < 8; i++) half[i] = (MOD_N[i] >> 1) | (i < 7 ? MOD_N[i + 1] << 31 : 0);
    if (bn_cmp(k1, half, 8) > 0) {
        *k1_neg = 1;
        bn_sub(k1, MOD_N, k1, 8);
    }
    if (bn_cmp(k2, half, 8) > 0) {
        *k2_neg = 1;
        bn_sub(k2, MOD_N, k2, 8);
    }
    if (words_to_limbs(k1, 8, k1l, 5) | words_to_limbs(k2, 8, k2l, 5)) *err |= P2E_O_ERR_LIMB_RANGE;
}

/* ------------------------------------------------------------------------------------------ */
/* native curve arithmetic for constants (curve/curve_types.rs:83-102, curve_adds.rs)           */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    w32 x[8], y[8];
} apt;
#define FB_WINDOWS 66
/* one curve of the crate (curve/curve_types.rs:15-42): moduli, a, b, G, and the constants its gadgets derive */
typedef struct curve_t {
    int fbase, fscalar;     /* P2E_O_FIELD_* of C::BaseField / C::ScalarField */
    const w32 *p, *n, *a, *b;
    apt g;
    u64 A_L[NL], B_L[NL];   /* constant_nonnative(C::A / C::B): limbs; the COUNT (convert_base, quirk Q5) is a_nl / b_nl */
    int a_nl, b_nl;
    apt rando, neg_rando;                   /* KeccakHash::<32>(0) * G, gadgets/curve_fixed_base.rs:34-38 */
    u64 RANDO_L[2][NL], NEG_RANDO_L[2][NL];
    u64 FB_TABLE[FB_WINDOWS][16][2][NL];    /* [window][digit][x|y][limb]; slot 0 := slot 1 */
    apt start25, start25_264;               /* KeccakHash::<25>(0) * G and 2^264 times it, gadgets/curve_windowed_mul.rs:140-153 */
} curve_t;
static curve_t SECP, P256C;

static void ecc_double(const curve_t *c, apt *r, const apt *p) {
    w32 l[8], t[8], u[8], x3[8];
    fe_mulmod(t, p->x, p->x, c->p);
    fe_addmod(u, t, t, c->p);
    fe_addmod(t, u, t, c->p); /* 3x^2 */
    fe_addmod(t, t, c->a, c->p);
    fe_addmod(u, p->y, p->y, c->p);
    fe_invmod(u, u, c->p);
    fe_mulmod(l, t, u, c->p);
    fe_mulmod(x3, l, l, c->p);
    fe_submod(x3, x3, p->x, c->p);
    fe_submod(x3, x3, p->x, c->p);
    fe_submod(t, p->x, x3, c->p);
    fe_mulmod(t, l, t, c->p);
    fe_submod(r->y, t, p->y, c->p);
    memcpy(r->x, x3, 32);
}
static void ecc_add(const curve_t *c, apt *r, const apt *p, const apt *q) {
    w32 l[8], t[8], u[8], x3[8];
    fe_submod(t, q->y, p->y, c->p);
    fe_submod(u, q->x, p->x, c->p);
    fe_invmod(u, u, c->p);
    fe_mulmod(l, t, u, c->p);
    fe_mulmod(x3, l, l, c->p);
    fe_submod(x3, x3, p->x, c->p);
    fe_submod(x3, x3, q->x, c->p);
    fe_submod(t, p->x, x3, c->p);
    fe_mulmod(t, l, t, c->p);
    fe_submod(r->y, t, p->y, c->p);
    memcpy(r->x, x3, 32);
}
static void ecc_neg(const curve_t *c, apt *r, const apt *p) {
    w32 z[8] = {0};
    memcpy(r->x, p->x, 32);
    fe_submod(r->y, z, p->y, c->p);
}
/* k * p for an integer k < 2^256 (NOT reduced: from_noncanonical_biguint keeps the hash as it is); k != 0 */
static void ecc_mul(const curve_t *c, apt *r, const w32 *k, const apt *p) {
    apt acc, base = *p;
    int have = 0;
    for (int i = 0; i < 256; i++) {
        if ((k[i >> 5] >> (i & 31)) & 1) {
            if (have)
                ecc_add(c, &acc, &acc, &base);
            else {
                acc = base;
                have = 1;
            }
        }
        ecc_double(c, &base, &base);
    }
    *r = acc;
}

static apt RANDO, NEG_RANDO, NEG_RANDO_146;
static u64 RANDO_L[2][NL], NEG_RANDO_L[2][NL], NEG_RANDO_146_L[2][NL], BETA_L[NL], GLV_S_L[NL];
static int INIT_DONE = 0;

static void point_limbs(const apt *p, u64 out[2][NL]) {
    words_to_limbs(p->x, 8, out[0], NL);
    words_to_limbs(p->y, 8, out[1], NL);
}
/* limbs constant_biguint gives a constant (convert_base gadgets/biguint.rs:27-51: no zero limbs on top) */
static int const_nlimbs(const u64 *l) {
    int n = NL;
    while (n > 0 && l[n - 1] == 0) n--;
    return n;
}
static void curve_init(curve_t *c, int fbase, int fscalar, const w32 *a, const w32 *b, const w32 *gx, const w32 *gy) {
    c->fbase = fbase, c->fscalar = fscalar;
    c->p = modulus_of(fbase), c->n = modulus_of(fscalar), c->a = a, c->b = b;
    memcpy(c->g.x, gx, 32);
    memcpy(c->g.y, gy, 32);
    words_to_limbs(a, 8, c->A_L, NL);
    words_to_limbs(b, 8, c->B_L, NL);
    c->a_nl = const_nlimbs(c->A_L), c->b_nl = const_nlimbs(c->B_L);
    w32 h[8];
    bytes_to_words(HASH0_LE, h);
    ecc_mul(c, &c->rando, h, &c->g);
    ecc_neg(c, &c->neg_rando, &c->rando);
    point_limbs(&c->rando, c->RANDO_L);
    point_limbs(&c->neg_rando, c->NEG_RANDO_L);
    /* KeccakHash::<25>: the first 25 bytes of the same digest (gadgets/curve_windowed_mul.rs:140-144) */
    uint8_t h25[32];
    memset(h25, 0, sizeof h25);
    memcpy(h25, HASH0_LE, 25);
    bytes_to_words(h25, h);
    ecc_mul(c, &c->start25, h, &c->g);
    apt d = c->start25;
    for (int i = 0; i < 4 * FB_WINDOWS; i++) ecc_double(c, &d, &d); /* windows.len() * 4 doublings, :146-152 */
    c->start25_264 = d; /* the gadget negates it itself (curve_neg of the constant, :168-169) */
    /* gadgets/curve_fixed_base.rs:24-30,45-56 */
    apt base = c->g;
    for (int w = 0; w < FB_WINDOWS; w++) {
        apt acc = base;
        for (int t = 1; t < 16; t++) {
            point_limbs(&acc, c->FB_TABLE[w][t]);
            if (t == 1) point_limbs(&acc, c->FB_TABLE[w][0]);
            if (t < 15) {
                if (t == 1)
                    ecc_double(c, &acc, &acc);
                else
                    ecc_add(c, &acc, &acc, &base);
            }
        }
        for (int i = 0; i < 4; i++) ecc_double(c, &base, &base);
    }
}
static void oracle_init(void) {
#pragma omp critical(p2e_oracle_init)
    {
        if (!INIT_DONE) {
            INV_2_29 = gl_pow(1ull << BITS, P_GL - 2);
            curve_init(&SECP, P2E_O_FIELD_BASE, P2E_O_FIELD_SCALAR, SECP_A, SECP_B, GEN_X, GEN_Y);
            curve_init(&P256C, P2E_O_FIELD_P256_BASE, P2E_O_FIELD_P256_SCALAR, P256_A, P256_B, P256_GX, P256_GY);
            RANDO = SECP.rando;
            NEG_RANDO = SECP.neg_rando;
            apt d = RANDO;
            for (int i = 0; i < 146; i++) ecc_double(&SECP, &d, &d); /* gadgets/curve_msm.rs:74 (2*73 doublings) */
            ecc_neg(&SECP, &NEG_RANDO_146, &d);
            point_limbs(&RANDO, RANDO_L);
            point_limbs(&NEG_RANDO, NEG_RANDO_L);
            point_limbs(&NEG_RANDO_146, NEG_RANDO_146_L);
            words_to_limbs(GLV_BETA, 8, BETA_L, NL);
            words_to_limbs(GLV_S, 8, GLV_S_L, NL);
            INIT_DONE = 1;
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* walker: evaluates the gadget schedule for one element, emitting every generator output       */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    u64 l[NL];
} nn; /* always padded to 9 limbs; limb *counts* never change a generator output on this path */
typedef struct {
    nn x, y;
} pt;
typedef struct {
    u64 *out;
    size_t ld, idx, col;
    uint8_t err;
    /* values of the targets plonky2's built-in generators fill on the same path (bool
     * selects, window bits / digits, random-access selections, is_equal / not results) in gadget creation order;
     * a separate column matrix, NULL = not wanted */
    u64 *aux;
    size_t ald, acol;
    const struct curve_t *cv; /* NULL = secp256k1 */
} walker;
static const curve_t *cv_of(const walker *w) { return w->cv ? w->cv : &SECP; }

static void aux_emit(walker *w, const u64 *v, int n) {
    if (w->aux)
        for (int i = 0; i < n; i++) w->aux[(w->acol + (size_t)i) * w->ald + w->idx] = v[i];
    w->acol += (size_t)n;
}
static u64 w_not(walker *w, u64 b) { /* builder.not(b) = 1 - b */
    u64 v = 1 - b;
    aux_emit(w, &v, 1);
    return v;
}
static u64 w_is_zero(walker *w, u64 x) { /* builder.is_equal(x, zero) */
    u64 v = x == 0;
    aux_emit(w, &v, 1);
    return v;
}

static void emit(walker *w, const u64 *v, int n) {
    if (w->out)
        for (int i = 0; i < n; i++) w->out[(w->col + (size_t)i) * w->ld + w->idx] = v[i];
    w->col += (size_t)n;
}
static nn w_add(walker *w, const nn *a, const nn *b, int field) {
    nn s;
    u64 ov;
    gen_add(a->l, NL, b->l, NL, modulus_of(field), s.l, &ov, &w->err);
    emit(w, s.l, NL);
    emit(w, &ov, 1);
    return s;
}
static nn w_sub(walker *w, const nn *a, const nn *b, int field) {
    nn d;
    u64 ov;
    gen_sub(a->l, NL, b->l, NL, modulus_of(field), d.l, &ov, &w->err);
    emit(w, d.l, NL);
    emit(w, &ov, 1);
    return d;
}
static nn w_mul(walker *w, const nn *x, const nn *y, int field) {
    nn r;
    u64 q[NL], cs[2 * NL - 1], b[2 * NL - 2];
    gen_mul(x->l, y->l, modulus_of(field), r.l, q, cs, &w->err);
    gen_checksum(cs, b, &w->err);
    emit(w, r.l, NL);
    emit(w, q, NL);
    emit(w, cs, 2 * NL - 1);
    emit(w, b, 2 * NL - 2);
    return r;
}
static nn w_inv(walker *w, const nn *x, int field) {
    nn inv;
    u64 div[NL];
    gen_inv(x->l, NL, modulus_of(field), inv.l, div, &w->err);
    emit(w, inv.l, NL);
    emit(w, div, NL);
    return inv;
}
static nn nn_zero(void) {
    nn z;
    memset(&z, 0, sizeof z);
    return z;
}
static nn nn_from(const u64 *l) {
    nn z;
    memcpy(z.l, l, sizeof z.l);
    return z;
}
/* gadgets/biguint.rs:360-374: one builder.mul per limb of `a`; nl = number of limbs the target really has */
static nn w_mul_bool(walker *w, const nn *a, int nl, u64 b) {
    nn r;
    for (int i = 0; i < NL; i++) r.l[i] = gl_mul(a->l[i], b);
    aux_emit(w, r.l, nl);
    return r;
}
static nn w_cond_neg(walker *w, const nn *x, int nl, u64 b, int field) { /* gadgets/nonnative.rs:584-596 */
    u64 not_b = w_not(w, b);
    nn z = nn_zero();
    nn neg = w_sub(w, &z, x, field);
    nn t = w_mul_bool(w, &neg, NL, b), f = w_mul_bool(w, x, nl, not_b);
    return w_add(w, &t, &f, field);
}
static pt w_curve_add(walker *w, const pt *p1, const pt *p2) { /* gadgets/curve.rs:202-223 */
    const int F = cv_of(w)->fbase;
    nn u = w_sub(w, &p2->y, &p1->y, F);
    nn v = w_sub(w, &p2->x, &p1->x, F);
    nn vi = w_inv(w, &v, F);
    nn s = w_mul(w, &u, &vi, F);
    nn s2 = w_mul(w, &s, &s, F);
    nn xs = w_add(w, &p2->x, &p1->x, F);
    pt r;
    r.x = w_sub(w, &s2, &xs, F);
    nn xd = w_sub(w, &p1->x, &r.x, F);
    nn pr = w_mul(w, &s, &xd, F);
    r.y = w_sub(w, &pr, &p1->y, F);
    return r;
}
static pt w_curve_double(walker *w, const pt *p) { /* gadgets/curve.rs:160-185 */
    const curve_t *cv = cv_of(w);
    const int F = cv->fbase;
    nn dy = w_add(w, &p->y, &p->y, F);
    nn idy = w_inv(w, &dy, F);
    nn xx = w_mul(w, &p->x, &p->x, F);
    u64 summ[4][NL];
    int nls[4] = {NL, NL, NL, cv->a_nl};
    memcpy(summ[0], xx.l, sizeof xx.l);
    memcpy(summ[1], xx.l, sizeof xx.l);
    memcpy(summ[2], xx.l, sizeof xx.l);
    memcpy(summ[3], cv->A_L, sizeof summ[3]); /* constant_nonnative(C::A): 0 limbs on secp256k1 */
    nn t;
    u64 ov;
    gen_add_many((const u64(*)[NL])summ, nls, 4, cv->p, t.l, &ov, &w->err);
    emit(w, t.l, NL);
    emit(w, &ov, 1);
    nn l = w_mul(w, &t, &idy, F);
    nn l2 = w_mul(w, &l, &l, F);
    nn xd2 = w_add(w, &p->x, &p->x, F);
    pt r;
    r.x = w_sub(w, &l2, &xd2, F);
    nn xdf = w_sub(w, &p->x, &r.x, F);
    nn lx = w_mul(w, &l, &xdf, F);
    r.y = w_sub(w, &lx, &p->y, F);
    return r;
}
/* gadgets/curve.rs:225-243; nlx, nly = limb counts of p1 (a constant point may have fewer than 9) */
static pt w_curve_cond_add(walker *w, const pt *p1, int nlx, int nly, const pt *p2, u64 b) {
    u64 not_b = w_not(w, b);
    pt s = w_curve_add(w, p1, p2);
    nn xt = w_mul_bool(w, &s.x, NL, b);
    nn yt = w_mul_bool(w, &s.y, NL, b);
    nn xf = w_mul_bool(w, &p1->x, nlx, not_b);
    nn yf = w_mul_bool(w, &p1->y, nly, not_b);
    pt r;
    r.x = w_add(w, &xt, &xf, cv_of(w)->fbase);
    r.y = w_add(w, &yt, &yf, cv_of(w)->fbase);
    return r;
}
static pt pt_from(u64 l[2][NL]) {
    pt p;
    memcpy(p.x.l, l[0], sizeof p.x.l);
    memcpy(p.y.l, l[1], sizeof p.y.l);
    return p;
}
/* digit t of width wbits (2 or 4) of a limb vector: bits of each 29-bit limb LE, limb-major
 * (gadgets/split_nonnative.rs:25-72) */
static unsigned digit_of(const u64 *limbs, int nl, int wbits, int t) {
    unsigned d = 0;
    for (int k = 0; k < wbits; k++) {
        int bit = t * wbits + k;
        int li = bit / BITS;
        if (li < nl) d |= (unsigned)((limbs[li] >> (bit % BITS)) & 1) << k;
    }
    return d;
}
/* split_le_base::<2>(limb, 29) of every limb (gadgets/split_nonnative.rs:33,60) */
static void aux_bits(walker *w, const u64 *limbs, int nl) {
    for (int i = 0; i < nl; i++)
        for (int k = 0; k < BITS; k++) {
            u64 b = (limbs[i] >> k) & 1;
            aux_emit(w, &b, 1);
        }
}
/* random_access_curve_points gadgets/curve_windowed_mul.rs:74-118: selected x limbs, then selected y limbs */
static void aux_point(walker *w, const pt *p) {
    aux_emit(w, p->x.l, NL);
    aux_emit(w, p->y.l, NL);
}
static pt w_fixed_base(walker *w, const nn *scalar) { /* gadgets/curve_fixed_base.rs:18-66; base = the curve's generator */
    const curve_t *cv = cv_of(w);
    u64(*RANDO_L)[NL] = (u64(*)[NL])cv->RANDO_L, (*NEG_RANDO_L)[NL] = (u64(*)[NL])cv->NEG_RANDO_L;
    pt result = pt_from(RANDO_L);
    int nlx = const_nlimbs(RANDO_L[0]), nly = const_nlimbs(RANDO_L[1]);
    /* split_nonnative_to_4_bit_limbs gadgets/split_nonnative.rs:25-50 */
    aux_bits(w, scalar->l, NL);
    for (int i = 0; i < FB_WINDOWS; i++) {
        u64 d = digit_of(scalar->l, NL, 4, i);
        u64 c[3] = {d & 3, d >> 2, d}; /* lower, upper, combined limb */
        aux_emit(w, c, 3);
    }
    for (int i = 0; i < FB_WINDOWS; i++) {
        unsigned d = digit_of(scalar->l, NL, 4, i);
        u64 should_add = w_not(w, w_is_zero(w, d));
        pt r = pt_from((u64(*)[NL])cv->FB_TABLE[i][d]);
        aux_point(w, &r);
        result = w_curve_cond_add(w, &result, nlx, nly, &r, should_add);
        nlx = nly = NL;
    }
    pt nr = pt_from(NEG_RANDO_L);
    return w_curve_add(w, &result, &nr);
}
static pt w_msm(walker *w, const pt *p, const pt *q, const u64 *n5, const u64 *m5) { /* gadgets/curve_msm.rs:21-79 */
    pt pre[16];
    pt rando = pt_from(RANDO_L), nr = pt_from(NEG_RANDO_L);
    /* split_nonnative_to_2_bit_limbs(n), then (m): gadgets/split_nonnative.rs:52-72 */
    for (int which = 0; which < 2; which++) {
        const u64 *l5 = which ? m5 : n5;
        aux_bits(w, l5, 5);
        for (int t = 0; t < 73; t++) {
            u64 dg = digit_of(l5, 5, 2, t);
            aux_emit(w, &dg, 1);
        }
    }
    for (int i = 0; i < 16; i++) pre[i] = *p;
    pt cur_p = rando, cur_q = rando;
    for (int i = 0; i < 4; i++) {
        pre[i] = cur_p;
        pre[4 * i] = cur_q;
        cur_p = w_curve_add(w, &cur_p, p);
        cur_q = w_curve_add(w, &cur_q, q);
    }
    for (int i = 1; i < 4; i++) {
        pre[i] = w_curve_add(w, &pre[i], &nr);
        pre[4 * i] = w_curve_add(w, &pre[4 * i], &nr);
    }
    for (int i = 1; i < 4; i++)
        for (int j = 1; j < 4; j++) pre[i + 4 * j] = w_curve_add(w, &pre[i], &pre[4 * j]);
    pt result = rando;
    for (int d = 72; d >= 0; d--) { /* 5 limbs -> 145 bits -> 146 -> 73 digits, MSB first */
        result = w_curve_double(w, &result);
        result = w_curve_double(w, &result);
        u64 idx = 4 * digit_of(m5, 5, 2, d) + digit_of(n5, 5, 2, d); /* mul_add(four, limb_m, limb_n) */
        aux_emit(w, &idx, 1);
        aux_point(w, &pre[idx]);
        u64 should_add = w_not(w, w_is_zero(w, idx));
        result = w_curve_cond_add(w, &result, NL, NL, &pre[idx], should_add);
    }
    pt to_add = pt_from(NEG_RANDO_146_L);
    return w_curve_add(w, &result, &to_add);
}
static int nn_eq(const nn *a, const nn *b) { return memcmp(a->l, b->l, sizeof a->l) == 0; }
static pt w_glv_mul(walker *w, const pt *p, const nn *k, int *ok) { /* gadgets/glv.rs:53-104 */
    const int S = P2E_O_FIELD_SCALAR;
    nn k1 = nn_zero(), k2 = nn_zero();
    u64 n1, n2;
    gen_glv(k->l, NL, k1.l, k2.l, &n1, &n2, &w->err);
    emit(w, k1.l, 5);
    emit(w, k2.l, 5);
    emit(w, &n1, 1);
    emit(w, &n2, 1);
    nn k1r = w_cond_neg(w, &k1, 5, n1, S);
    nn k2r = w_cond_neg(w, &k2, 5, n2, S);
    nn gs = nn_from(GLV_S_L);
    nn sb = w_mul(w, &gs, &k2r, S);
    sb = w_add(w, &sb, &k1r, S);
    *ok &= nn_eq(&sb, k);
    nn beta = nn_from(BETA_L);
    pt sp;
    sp.x = w_mul(w, &beta, &p->x, P2E_O_FIELD_BASE);
    sp.y = p->y;
    pt pn, spn;
    pn.x = p->x;
    pn.y = w_cond_neg(w, &p->y, NL, n1, P2E_O_FIELD_BASE);
    spn.x = sp.x;
    spn.y = w_cond_neg(w, &sp.y, NL, n2, P2E_O_FIELD_BASE);
    return w_msm(w, &pn, &spn, k1.l, k2.l);
}
static nn nn_from_bytes(const uint8_t *b) {
    w32 wv[8];
    nn r;
    bytes_to_words(b, wv);
    words_to_limbs(wv, 8, r.l, NL);
    return r;
}
static void walk_verify(walker *w, const uint8_t *msg32, const uint8_t *r32, const uint8_t *s32,
                        const uint8_t *pkx32, const uint8_t *pky32, uint8_t *flag) { /* gadgets/ecdsa.rs:30-53 */
    const int F = P2E_O_FIELD_BASE, S = P2E_O_FIELD_SCALAR;
    nn msg = nn_from_bytes(msg32), r = nn_from_bytes(r32), s = nn_from_bytes(s32);
    pt pk;
    pk.x = nn_from_bytes(pkx32);
    pk.y = nn_from_bytes(pky32);
    int ok = 1;
    /* curve_assert_valid gadgets/curve.rs:123-135 */
    nn a0 = nn_zero(), b7 = nn_from(SECP.B_L);
    nn y2 = w_mul(w, &pk.y, &pk.y, F);
    nn x2 = w_mul(w, &pk.x, &pk.x, F);
    nn x3 = w_mul(w, &x2, &pk.x, F);
    nn ax = w_mul(w, &a0, &pk.x, F);
    nn axb = w_add(w, &ax, &b7, F);
    nn rhs = w_add(w, &x3, &axb, F);
    ok &= nn_eq(&y2, &rhs);
    nn c = w_inv(w, &s, S);
    nn u1 = w_mul(w, &msg, &c, S);
    nn u2 = w_mul(w, &r, &c, S);
    pt p1 = w_fixed_base(w, &u1);
    pt p2 = w_glv_mul(w, &pk, &u2, &ok);
    pt sum = w_curve_add(w, &p1, &p2);
    ok &= nn_eq(&sum.x, &r);
    *flag = (uint8_t)ok;
}

/* ------------------------------------------------------------------------------------------ */
/* curve programs: curve_scalar_mul_windowed, curve_scalar_mul, verify_p256_message_circuit    */
/* ------------------------------------------------------------------------------------------ */
static pt pt_from_apt(const apt *a) {
    u64 l[2][NL];
    point_limbs(a, l);
    return pt_from(l);
}
/* gadgets/curve.rs:137-147: neg_nonnative(y) = sub(zero, y), the zero constant has 0 limbs (nonnative.rs:496-499) */
static pt w_curve_neg(walker *w, const pt *p) {
    nn z = nn_zero();
    pt r;
    r.x = p->x;
    r.y = w_sub(w, &z, &p->y, cv_of(w)->fbase);
    return r;
}
/* split_nonnative_to_4_bit_limbs gadgets/split_nonnative.rs:25-50: the bits, then (lower, upper, limb) per nibble */
static void aux_split4(walker *w, const nn *scalar) {
    aux_bits(w, scalar->l, NL);
    for (int i = 0; i < FB_WINDOWS; i++) {
        u64 d = digit_of(scalar->l, NL, 4, i);
        u64 c[3] = {d & 3, d >> 2, d};
        aux_emit(w, c, 3);
    }
}
/* gadgets/curve_windowed_mul.rs:131-173; g = the rand() point of precompute_window (:57) */
static pt w_windowed_mul(walker *w, const pt *p, const nn *n, const apt *g) {
    const curve_t *cv = cv_of(w);
    aux_split4(w, n);
    /* precompute_window :52-72 */
    apt neg_g;
    ecc_neg(cv, &neg_g, g);
    pt pre[16], neg = pt_from_apt(&neg_g);
    pre[0] = pt_from_apt(g);
    for (int i = 1; i < 16; i++) pre[i] = w_curve_add(w, p, &pre[i - 1]);
    for (int i = 1; i < 16; i++) pre[i] = w_curve_add(w, &neg, &pre[i]);
    pt result = pt_from_apt(&cv->start25);
    for (int i = FB_WINDOWS - 1; i >= 0; i--) { /* :157-166, most significant window first */
        for (int k = 0; k < 4; k++) result = w_curve_double(w, &result);
        unsigned d = digit_of(n->l, NL, 4, i);
        aux_point(w, &pre[d]); /* random_access_curve_points, then is_equal, then not */
        u64 should_add = w_not(w, w_is_zero(w, d));
        result = w_curve_cond_add(w, &result, NL, NL, &pre[d], should_add);
    }
    pt to_subtract = pt_from_apt(&cv->start25_264);
    pt to_add = w_curve_neg(w, &to_subtract);
    return w_curve_add(w, &result, &to_add);
}
/* gadgets/curve.rs:245-285; rando = the rand() blinding point of :253 */
static pt w_scalar_mul(walker *w, const pt *p, const nn *n, const apt *rando) {
    const int F = cv_of(w)->fbase;
    aux_bits(w, n->l, NL); /* split_nonnative_to_bits gadgets/nonnative.rs:566-582 */
    pt randot = pt_from_apt(rando);
    pt result = randot; /* add_virtual_affine_point_target + connect: 9 limbs */
    pt two_i_times_p = *p;
    for (int i = 0; i < NL * BITS; i++) {
        u64 bit = (n->l[i / BITS] >> (i % BITS)) & 1;
        u64 not_bit = w_not(w, bit);
        pt rp = w_curve_add(w, &result, &two_i_times_p);
        nn xt = w_mul_bool(w, &rp.x, NL, bit);
        nn xf = w_mul_bool(w, &result.x, NL, not_bit);
        nn yt = w_mul_bool(w, &rp.y, NL, bit);
        nn yf = w_mul_bool(w, &result.y, NL, not_bit);
        result.x = w_add(w, &xt, &xf, F);
        result.y = w_add(w, &yt, &yf, F);
        two_i_times_p = w_curve_double(w, &two_i_times_p);
    }
    pt neg_r = w_curve_neg(w, &randot);
    return w_curve_add(w, &result, &neg_r);
}
/* gadgets/ecdsa.rs:55-78 */
static void walk_verify_p256(walker *w, const uint8_t *msg32, const uint8_t *r32, const uint8_t *s32,
                             const uint8_t *pkx32, const uint8_t *pky32, const apt *g, uint8_t *flag) {
    const curve_t *cv = cv_of(w);
    const int F = cv->fbase, S = cv->fscalar;
    nn msg = nn_from_bytes(msg32), r = nn_from_bytes(r32), s = nn_from_bytes(s32);
    pt pk;
    pk.x = nn_from_bytes(pkx32);
    pk.y = nn_from_bytes(pky32);
    int ok = 1;
    nn a = nn_from(cv->A_L), b = nn_from(cv->B_L); /* curve_assert_valid gadgets/curve.rs:123-135 */
    nn y2 = w_mul(w, &pk.y, &pk.y, F);
    nn x2 = w_mul(w, &pk.x, &pk.x, F);
    nn x3 = w_mul(w, &x2, &pk.x, F);
    nn ax = w_mul(w, &a, &pk.x, F);
    nn axb = w_add(w, &ax, &b, F);
    nn rhs = w_add(w, &x3, &axb, F);
    ok &= nn_eq(&y2, &rhs);
    nn c = w_inv(w, &s, S);
    nn u1 = w_mul(w, &msg, &c, S);
    nn u2 = w_mul(w, &r, &c, S);
    pt p1 = w_fixed_base(w, &u1);
    pt p2 = w_windowed_mul(w, &pk, &u2, g);
    pt sum = w_curve_add(w, &p1, &p2);
    ok &= nn_eq(&sum.x, &r);
    *flag = (uint8_t)ok;
}
typedef struct {
    int kind;
    const curve_t *cv;
    apt blind;
    const uint8_t *msg, *r, *s, *px, *py;
} cp_job;
static void walk_curve_program(walker *w, const cp_job *J, size_t i, uint8_t *flag) {
    w->cv = J->cv;
    if (J->kind == P2E_O_CP_VERIFY) {
        walk_verify_p256(w, J->msg + 32 * i, J->r + 32 * i, J->s + 32 * i, J->px + 32 * i, J->py + 32 * i, &J->blind, flag);
        return;
    }
    pt p;
    p.x = nn_from_bytes(J->px + 32 * i);
    p.y = nn_from_bytes(J->py + 32 * i);
    nn k = nn_from_bytes(J->msg + 32 * i);
    if (J->kind == P2E_O_CP_WINDOWED_MUL)
        (void)w_windowed_mul(w, &p, &k, &J->blind);
    else
        (void)w_scalar_mul(w, &p, &k, &J->blind);
    *flag = 1;
}

/* ------------------------------------------------------------------------------------------ */
/* lock-step groups: batch inversion across signatures (optimised-CPU baseline only)            */
/* ------------------------------------------------------------------------------------------ */
#define LS_MAX 256
#define LS_STACK (512 * 1024)
typedef struct lockstep {
    ucontext_t main, co[LS_MAX];
    char *stack;
    w32 req[LS_MAX][8], res[LS_MAX][8];
    const w32 *mod[LS_MAX];
    int pending[LS_MAX], finished[LS_MAX];
    int cur, count;
    const uint8_t *msg, *r, *s, *pkx, *pky;
    const cp_job *job; /* non-NULL: a curve program instead of verify_secp256k1_message_circuit */
    uint64_t *cols, *aux;
    size_t ld, ald, first;
    uint8_t *err, *flags;
} lockstep;

static void ls_invmod(w32 *r, const w32 *a, const w32 *m) {
    lockstep *L = LS;
    const int k = L->cur;
    memcpy(L->req[k], a, 32);
    L->mod[k] = m;
    L->pending[k] = 1;
    swapcontext(&L->co[k], &L->main); /* resumed once the group's batch inversion has run */
    memcpy(r, L->res[k], 32);
}
static void ls_entry(void) {
    lockstep *L = LS;
    const int k = L->cur;
    const size_t i = L->first + (size_t)k;
    walker w = {L->cols, L->ld, i, 0, 0, L->aux, L->ald, 0, NULL};
    uint8_t f = 0;
    if (L->job)
        walk_curve_program(&w, L->job, i, &f);
    else
        walk_verify(&w, L->msg + 32 * i, L->r + 32 * i, L->s + 32 * i, L->pkx + 32 * i, L->pky + 32 * i, &f);
    L->err[i] = w.err;
    if (L->flags) L->flags[i] = f;
    L->finished[k] = 1; /* returning switches to uc_link = main */
}
/* one Fermat ladder for all pending requests with modulus m (they are non-zero: gen_inv filters zero) */
static void ls_batch_invert(lockstep *L, const w32 *m) {
    static __thread w32 prefix[LS_MAX][8];
    int idx[LS_MAX], cnt = 0;
    for (int k = 0; k < L->count; k++)
        if (L->pending[k] && L->mod[k] == m) idx[cnt++] = k;
    if (!cnt) return;
    w32 acc[8] = {1, 0, 0, 0, 0, 0, 0, 0}, inv[8], t[8];
    for (int j = 0; j < cnt; j++) {
        memcpy(prefix[j], acc, 32);
        fe_mulmod(acc, acc, L->req[idx[j]], m);
    }
    fe_invmod(inv, acc, m);
    for (int j = cnt - 1; j >= 0; j--) {
        fe_mulmod(L->res[idx[j]], inv, prefix[j], m);
        fe_mulmod(t, inv, L->req[idx[j]], m);
        memcpy(inv, t, 32);
        L->pending[idx[j]] = 0;
    }
}
static void ls_run_group(lockstep *L) {
    for (int k = 0; k < L->count; k++) {
        L->pending[k] = L->finished[k] = 0;
        getcontext(&L->co[k]);
        L->co[k].uc_stack.ss_sp = L->stack + (size_t)k * LS_STACK;
        L->co[k].uc_stack.ss_size = LS_STACK;
        L->co[k].uc_link = &L->main;
        makecontext(&L->co[k], ls_entry, 0);
    }
    for (;;) {
        int live = 0;
        for (int k = 0; k < L->count; k++) {
            if (L->finished[k]) continue;
            L->cur = k;
            swapcontext(&L->main, &L->co[k]); /* runs signature k up to its next inverse (or to the end) */
            live += !L->finished[k];
        }
        if (!live) break;
        ls_batch_invert(L, MOD_P);
        ls_batch_invert(L, MOD_N);
        ls_batch_invert(L, MOD_P256);
        ls_batch_invert(L, MOD_N256);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* batch entry points                                                                           */
/* ------------------------------------------------------------------------------------------ */
static void gather(const u64 *base, size_t ld, size_t i, u64 *out, int n) {
    for (int k = 0; k < n; k++) out[k] = base[(size_t)k * ld + i];
}
static void scatter(u64 *base, size_t ld, size_t i, const u64 *in, int n) {
    for (int k = 0; k < n; k++) base[(size_t)k * ld + i] = in[k];
}
static long count_err(const uint8_t *err, size_t n) {
    long c = 0;
    for (size_t i = 0; i < n; i++) c += err[i] != 0;
    return c;
}

long p2e_oracle_mul_witness(int field, const uint64_t *x, const uint64_t *y, uint64_t *r, uint64_t *q,
                            uint64_t *cs, uint64_t *b, size_t n, size_t ld, uint8_t *err) {
    oracle_init();
    const w32 *m = modulus_of(field);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        u64 xl[NL], yl[NL], rl[NL], ql[NL], csl[2 * NL - 1], bl[2 * NL - 2];
        uint8_t e = 0;
        gather(x, ld, i, xl, NL);
        gather(y, ld, i, yl, NL);
        gen_mul(xl, yl, m, rl, ql, csl, &e);
        memset(bl, 0, sizeof bl);
        if (!e) gen_checksum(csl, bl, &e);
        scatter(r, ld, i, rl, NL);
        scatter(q, ld, i, ql, NL);
        scatter(cs, ld, i, csl, 2 * NL - 1);
        scatter(b, ld, i, bl, 2 * NL - 2);
        err[i] = e;
    }
    return count_err(err, n);
}
long p2e_oracle_checksum_witness(const uint64_t *a, uint64_t *b, size_t n, size_t ld, uint8_t *err) {
    oracle_init();
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        u64 al[2 * NL - 1], bl[2 * NL - 2];
        uint8_t e = 0;
        gather(a, ld, i, al, 2 * NL - 1);
        gen_checksum(al, bl, &e);
        scatter(b, ld, i, bl, 2 * NL - 2);
        err[i] = e;
    }
    return count_err(err, n);
}
long p2e_oracle_add_witness(int field, const uint64_t *a, const uint64_t *b, uint64_t *sum, uint64_t *ov,
                            size_t n, size_t ld, uint8_t *err) {
    oracle_init();
    const w32 *m = modulus_of(field);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        u64 al[NL], bl[NL], sl[NL], o;
        uint8_t e = 0;
        gather(a, ld, i, al, NL);
        gather(b, ld, i, bl, NL);
        gen_add(al, NL, bl, NL, m, sl, &o, &e);
        scatter(sum, ld, i, sl, NL);
        ov[i] = o;
        err[i] = e;
    }
    return count_err(err, n);
}
long p2e_oracle_sub_witness(int field, const uint64_t *a, const uint64_t *b, uint64_t *diff, uint64_t *ov,
                            size_t n, size_t ld, uint8_t *err) {
    oracle_init();
    const w32 *m = modulus_of(field);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        u64 al[NL], bl[NL], dl[NL], o;
        uint8_t e = 0;
        gather(a, ld, i, al, NL);
        gather(b, ld, i, bl, NL);
        gen_sub(al, NL, bl, NL, m, dl, &o, &e);
        scatter(diff, ld, i, dl, NL);
        ov[i] = o;
        err[i] = e;
    }
    return count_err(err, n);
}
long p2e_oracle_add_many_witness(int field, const uint64_t *summands, int k, uint64_t *sum, uint64_t *ov,
                                 size_t n, size_t ld, uint8_t *err) {
    oracle_init();
    const w32 *m = modulus_of(field);
    if (k < 1 || k > 8) return -1;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        u64 sm[8][NL], sl[NL], o;
        int nls[8];
        uint8_t e = 0;
        for (int t = 0; t < k; t++) {
            gather(summands + (size_t)t * NL * ld, ld, i, sm[t], NL);
            nls[t] = NL;
        }
        gen_add_many((const u64(*)[NL])sm, nls, k, m, sl, &o, &e);
        scatter(sum, ld, i, sl, NL);
        ov[i] = o;
        err[i] = e;
    }
    return count_err(err, n);
}
long p2e_oracle_inv_witness(int field, const uint64_t *x, uint64_t *inv, uint64_t *div, size_t n, size_t ld,
                            uint8_t *err) {
    oracle_init();
    const w32 *m = modulus_of(field);
#pragma omp parallel for schedule(dynamic, 16)
    for (size_t i = 0; i < n; i++) {
        u64 xl[NL], il[NL], dl[NL];
        uint8_t e = 0;
        gather(x, ld, i, xl, NL);
        gen_inv(xl, NL, m, il, dl, &e);
        scatter(inv, ld, i, il, NL);
        scatter(div, ld, i, dl, NL);
        err[i] = e;
    }
    return count_err(err, n);
}
/* number of limbs set_biguint_target sees (gadgets/biguint.rs:455-456): convert_base emits floor(32 D / 29) limbs for
 * the D u32 digits of the value -- zero ones included -- plus one more if bits are left over.  A value can therefore
 * "not fit" a target it would fit arithmetically (e.g. 289..290 bits into 10 limbs): the reference panics, so do we. */
static int ref_limb_len(const w32 *w, int nw) {
    int d = nw;
    while (d > 0 && w[d - 1] == 0) d--;
    int l = (32 * d) / BITS;
    int rest = 0;
    for (int bit = BITS * l; bit < 32 * d; bit++) rest |= (w[bit >> 5] >> (bit & 31)) & 1;
    return l + rest;
}
long p2e_oracle_div_rem(const uint64_t *a, int na, const uint64_t *b, int nb, uint64_t *div, uint64_t *rem, size_t n,
                        size_t ld, uint8_t *err) {
    if (na < 1 || na > 18 || nb < 1 || nb > 9) return -1;
    const int nd = nb > na + 1 ? 0 : na - nb + 1;
#pragma omp parallel for schedule(dynamic, 16)
    for (size_t i = 0; i < n; i++) {
        u64 al[18], bl[9], dl[18], rl[9];
        uint8_t e = 0;
        gather(a, ld, i, al, na);
        gather(b, ld, i, bl, nb);
        for (int k = 0; k < na; k++) e |= (al[k] >> BITS) ? P2E_O_ERR_LIMB_RANGE : 0;
        for (int k = 0; k < nb; k++) e |= (bl[k] >> BITS) ? P2E_O_ERR_LIMB_RANGE : 0;
        memset(dl, 0, sizeof dl);
        memset(rl, 0, sizeof rl);
        if (!e) {
            /* 29-bit limbs -> 32-bit words (no carries: limbs are in range) */
            w32 aw[18], bw[10], q[18], r[10];
            memset(aw, 0, sizeof aw);
            memset(bw, 0, sizeof bw);
            for (int k = 0; k < na; k++) {
                int bit = BITS * k, wi = bit >> 5, sh = bit & 31;
                u64 v = al[k] << sh;
                aw[wi] |= (w32)v;
                if (wi + 1 < 18) aw[wi + 1] |= (w32)(v >> 32);
            }
            for (int k = 0; k < nb; k++) {
                int bit = BITS * k, wi = bit >> 5, sh = bit & 31;
                u64 v = bl[k] << sh;
                bw[wi] |= (w32)v;
                bw[wi + 1] |= (w32)(v >> 32);
            }
            int bn = 10;
            while (bn > 0 && bw[bn - 1] == 0) bn--;
            if (bn == 0) {
                e |= P2E_O_ERR_DIVISION_BY_ZERO;
            } else {
                memset(q, 0, sizeof q);
                memset(r, 0, sizeof r);
                if (bn == 1) { /* short division (bn_divrem needs a divisor of two words or more) */
                    u64 c = 0;
                    for (int k = 17; k >= 0; k--) {
                        c = (c << 32) | aw[k];
                        q[k] = (w32)(c / bw[0]);
                        c %= bw[0];
                    }
                    r[0] = (w32)c;
                } else {
                    bn_divrem(aw, 18, bw, bn, q, r);
                }
                if (words_to_limbs(q, 18, dl, nd) || ref_limb_len(q, 18) > nd) e |= P2E_O_ERR_LIMB_RANGE;
                (void)words_to_limbs(r, 10, rl, nb);
            }
        }
        if (e) {
            memset(dl, 0, sizeof dl);
            memset(rl, 0, sizeof rl);
        }
        scatter(div, ld, i, dl, nd);
        scatter(rem, ld, i, rl, nb);
        err[i] = e;
    }
    return count_err(err, n);
}
long p2e_oracle_glv_decompose(const uint64_t *k, uint64_t *k1, uint64_t *k2, uint64_t *k1_neg,
                              uint64_t *k2_neg, size_t n, size_t ld, uint8_t *err) {
    oracle_init();
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        u64 kl[NL], a[5], b[5], n1, n2;
        uint8_t e = 0;
        gather(k, ld, i, kl, NL);
        gen_glv(kl, NL, a, b, &n1, &n2, &e);
        scatter(k1, ld, i, a, 5);
        scatter(k2, ld, i, b, 5);
        k1_neg[i] = n1;
        k2_neg[i] = n2;
        err[i] = e;
    }
    return count_err(err, n);
}
long p2e_oracle_limb_split(const uint8_t *packed, uint64_t *limbs, size_t n, size_t ld) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        nn v = nn_from_bytes(packed + 32 * i);
        scatter(limbs, ld, i, v.l, NL);
    }
    return 0;
}
long p2e_oracle_limb_pack(const uint64_t *limbs, uint8_t *packed, size_t n, size_t ld, uint8_t *err) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        u64 l[NL];
        w32 w[8];
        uint8_t e = 0;
        gather(limbs, ld, i, l, NL);
        for (int k = 0; k < NL; k++)
            if (l[k] >> BITS) e |= P2E_O_ERR_LIMB_RANGE;
        if (limbs_to_words(l, NL, w, 8)) e |= P2E_O_ERR_VALUE_GE_2_256;
        if (e) memset(w, 0, sizeof w);
        words_to_bytes(w, packed + 32 * i);
        err[i] = e;
    }
    return count_err(err, n);
}

long p2e_oracle_verify_witness(const uint8_t *msg, const uint8_t *r, const uint8_t *s, const uint8_t *pkx,
                               const uint8_t *pky, uint64_t *cols, size_t n, size_t ld, uint8_t *err,
                               uint8_t *flags, int nthreads) {
    return p2e_oracle_verify_witness_aux(msg, r, s, pkx, pky, cols, n, ld, NULL, 0, err, flags, nthreads);
}
long p2e_oracle_verify_witness_aux(const uint8_t *msg, const uint8_t *r, const uint8_t *s, const uint8_t *pkx,
                                   const uint8_t *pky, uint64_t *cols, size_t n, size_t ld, uint64_t *aux,
                                   size_t ald, uint8_t *err, uint8_t *flags, int nthreads) {
    oracle_init();
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t i = 0; i < n; i++) {
        walker w = {cols, ld, i, 0, 0, aux, ald, 0, NULL};
        uint8_t f = 0;
        walk_verify(&w, msg + 32 * i, r + 32 * i, s + 32 * i, pkx + 32 * i, pky + 32 * i, &f);
        err[i] = w.err;
        if (flags) flags[i] = f;
    }
    return count_err(err, n);
}
static long run_lockstep(const cp_job *job, const uint8_t *msg, const uint8_t *r, const uint8_t *s, const uint8_t *pkx,
                         const uint8_t *pky, uint64_t *cols, size_t n, size_t ld, uint64_t *aux, size_t ald, uint8_t *err,
                         uint8_t *flags, int nthreads, int group) {
    if (group < 1) group = 64;
    if (group > LS_MAX) group = LS_MAX;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
    const size_t ngroups = (n + (size_t)group - 1) / (size_t)group;
    int failed = 0;
#pragma omp parallel
    {
        lockstep *L = (lockstep *)malloc(sizeof(lockstep));
        char *stack = (char *)malloc((size_t)group * LS_STACK);
        if (!L || !stack) {
#pragma omp atomic write
            failed = 1;
        } else {
            L->stack = stack;
            L->msg = msg, L->r = r, L->s = s, L->pkx = pkx, L->pky = pky;
            L->job = job;
            L->cols = cols, L->ld = ld, L->aux = aux, L->ald = ald, L->err = err, L->flags = flags;
#pragma omp for schedule(dynamic, 1)
            for (size_t g = 0; g < ngroups; g++) {
                L->first = g * (size_t)group;
                L->count = (int)(n - L->first < (size_t)group ? n - L->first : (size_t)group);
                LS = L;
                ls_run_group(L);
                LS = NULL;
            }
        }
        free(stack);
        free(L);
    }
    if (failed) return -1;
    return count_err(err, n);
}
long p2e_oracle_verify_witness_lockstep(const uint8_t *msg, const uint8_t *r, const uint8_t *s, const uint8_t *pkx,
                                        const uint8_t *pky, uint64_t *cols, size_t n, size_t ld, uint8_t *err,
                                        uint8_t *flags, int nthreads, int group) {
    oracle_init();
    return run_lockstep(NULL, msg, r, s, pkx, pky, cols, n, ld, NULL, 0, err, flags, nthreads, group);
}
/* the lock-step walk recording the built-in-generator values as well (aux[8959][n], as p2e_oracle_verify_witness_aux) */
long p2e_oracle_verify_witness_aux_lockstep(const uint8_t *msg, const uint8_t *r, const uint8_t *s, const uint8_t *pkx,
                                            const uint8_t *pky, uint64_t *cols, size_t n, size_t ld, uint64_t *aux, size_t ald,
                                            uint8_t *err, uint8_t *flags, int nthreads, int group) {
    oracle_init();
    return run_lockstep(NULL, msg, r, s, pkx, pky, cols, n, ld, aux, ald, err, flags, nthreads, group);
}
static int cp_job_init(cp_job *J, int kind, int curve, const uint8_t *bx, const uint8_t *by) {
    if (kind < 1 || kind > 3 || curve < 0 || curve > 1 || (kind == P2E_O_CP_VERIFY && curve != 1) || !bx || !by) return -1;
    oracle_init();
    memset(J, 0, sizeof *J);
    J->kind = kind;
    J->cv = curve ? &P256C : &SECP;
    bytes_to_words(bx, J->blind.x);
    bytes_to_words(by, J->blind.y);
    return 0;
}
long p2e_oracle_curve_program(int kind, int curve, const uint8_t *blind_x32, const uint8_t *blind_y32,
                              const uint8_t *msg, const uint8_t *r, const uint8_t *s, const uint8_t *px,
                              const uint8_t *py, uint64_t *cols, size_t n, size_t ld, uint64_t *aux, size_t ald,
                              uint8_t *err, uint8_t *flags, int nthreads, int lockstep_group) {
    cp_job J;
    if (cp_job_init(&J, kind, curve, blind_x32, blind_y32)) return -1;
    J.msg = msg, J.r = r, J.s = s, J.px = px, J.py = py;
    if (lockstep_group > 0) {
        long rc = run_lockstep(&J, msg, r, s, px, py, cols, n, ld, aux, ald, err, flags, nthreads, lockstep_group);
        return rc == -1 ? -2 : rc;
    }
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t i = 0; i < n; i++) {
        walker w = {cols, ld, i, 0, 0, aux, ald, 0, NULL};
        uint8_t f = 0;
        walk_curve_program(&w, &J, i, &f);
        err[i] = w.err;
        if (flags) flags[i] = f;
    }
    return count_err(err, n);
}
long p2e_oracle_curve_program_num_cols(int kind, int curve, long *num_aux) {
    cp_job J;
    uint8_t zero[32] = {0}, gx[32], gy[32];
    if (cp_job_init(&J, kind, curve, zero, zero)) return -1;
    /* any point will do for counting: 2G as the blinding point, G as the input point, all-ones bytes elsewhere */
    apt g2;
    ecc_double(J.cv, &g2, &J.cv->g);
    J.blind = g2;
    words_to_bytes(J.cv->g.x, gx);
    words_to_bytes(J.cv->g.y, gy);
    uint8_t ones[32];
    memset(ones, 0x11, sizeof ones);
    J.msg = J.r = J.s = ones, J.px = gx, J.py = gy;
    walker w = {NULL, 0, 0, 0, 0, NULL, 0, 0, NULL};
    uint8_t f = 0;
    walk_curve_program(&w, &J, 0, &f);
    if (num_aux) *num_aux = (long)w.acol;
    return (long)w.col;
}
long p2e_oracle_glv_mul_witness(const uint8_t *px, const uint8_t *py, const uint8_t *k, uint64_t *cols,
                                size_t n, size_t ld, uint8_t *err, uint8_t *flags, int nthreads) {
    return p2e_oracle_glv_mul_witness_aux(px, py, k, cols, n, ld, NULL, 0, err, flags, nthreads);
}
long p2e_oracle_glv_mul_witness_aux(const uint8_t *px, const uint8_t *py, const uint8_t *k, uint64_t *cols,
                                    size_t n, size_t ld, uint64_t *aux, size_t ald, uint8_t *err, uint8_t *flags,
                                    int nthreads) {
    oracle_init();
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t i = 0; i < n; i++) {
        walker w = {cols, ld, i, 0, 0, aux, ald, 0, NULL};
        pt p;
        p.x = nn_from_bytes(px + 32 * i);
        p.y = nn_from_bytes(py + 32 * i);
        nn kk = nn_from_bytes(k + 32 * i);
        int ok = 1;
        (void)w_glv_mul(&w, &p, &kk, &ok);
        err[i] = w.err;
        if (flags) flags[i] = (uint8_t)ok;
    }
    return count_err(err, n);
}
void p2e_oracle_rando(uint8_t x32[32], uint8_t y32[32]) {
    oracle_init();
    words_to_bytes(RANDO.x, x32);
    words_to_bytes(RANDO.y, y32);
}
int p2e_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
