// Per-lane bodies of the stand-alone generator kernels: one reference generator per entry point,
// operands and results as 29-bit-limb Goldilocks columns  col[limb * ld + i]  (include/p2e.h).
// Unlike the fused pipeline these accept everything the reference accepts: arbitrary Goldilocks
// elements as limbs for add/sub/inv (value must be < 2^256), 29-bit limbs for mul (value < 2^261).
#pragma once
#include "wit.hpp"

namespace p2e {

// emitter into a local array (fully unrolled -> registers)
struct ArrEmit {
    u64* a;
    int k;
    P2E_HD void put(u64 v) { a[k++] = v; }
    P2E_HD void put_at(int d, u64 v) { a[k + d] = v; }
    P2E_HD void skip(int d) { k += d; }
    P2E_HD void put_wide(u64 v) { put(v); }
    P2E_HD void put_wide_at(int d, u64 v) { put_at(d, v); }
    P2E_HD void skip_wide(int d) { skip(d); }
};

// sum limb_k * 2^(29k) for arbitrary 64-bit limbs; false if the value is >= 2^256
// (reference get_biguint_target gadgets/biguint.rs:444-452 + from_noncanonical_biguint panic)
P2E_HD bool value_of_limbs(const u64* limbs, int nl, U256& out) {
    u32 acc[12];
    P2E_UNROLL
    for (int i = 0; i < 12; i++) acc[i] = 0;
    P2E_UNROLL
    for (int k = 0; k < NL; k++) {
        if (k < nl) {
            int bit = BITS * k;
            int wi = bit >> 5, sh = bit & 31;
            u64 lo = limbs[k] << sh;
            u32 hi = sh ? (u32)(limbs[k] >> (64 - sh)) : 0u;
            u64 c = (u64)acc[wi] + (u32)lo;
            acc[wi] = (u32)c;
            c = (c >> 32) + acc[wi + 1] + (u32)(lo >> 32);
            acc[wi + 1] = (u32)c;
            c = (c >> 32) + acc[wi + 2] + hi;
            acc[wi + 2] = (u32)c;
            c >>= 32;
            P2E_UNROLL
            for (int j = 3; j < 12; j++) {
                if (wi + j < 12) {
                    c += acc[wi + j];
                    acc[wi + j] = (u32)c;
                    c >>= 32;
                }
            }
        }
    }
    P2E_UNROLL
    for (int i = 0; i < 8; i++) out.w[i] = acc[i];
    return (acc[8] | acc[9] | acc[10] | acc[11]) == 0;
}

P2E_HD void load_col(const u64* base, size_t ld, size_t i, u64* out, int n) {
    P2E_UNROLL
    for (int k = 0; k < 17; k++)
        if (k < n) out[k] = base[(size_t)k * ld + i];
}
P2E_HD void store_col(u64* base, size_t ld, size_t i, const u64* in, int n) {
    P2E_UNROLL
    for (int k = 0; k < 17; k++)
        if (k < n) base[(size_t)k * ld + i] = in[k];
}

// gates/mul_nonnative.rs:249-324 + :513-531
template <class MOD>
P2E_HD uint8_t prim_mul(const u64* x, const u64* y, u64* r, u64* q, u64* cs, u64* b, size_t ld, size_t i) {
    u64 xl[NL], yl[NL];
    load_col(x, ld, i, xl, NL);
    load_col(y, ld, i, yl, NL);
    uint8_t err = 0;
    u32 x29[NL], y29[NL];
    P2E_UNROLL
    for (int k = 0; k < NL; k++) {
        if ((xl[k] >> BITS) | (yl[k] >> BITS)) err |= ERR_LIMB_RANGE;
        x29[k] = (u32)xl[k] & MASK29;
        y29[k] = (u32)yl[k] & MASK29;
    }
    u32 xw[9], yw[9], prod[18], qw[9];
    U256 rv;
    pack29_wide(x29, xw);
    pack29_wide(y29, yw);
    mul_wide<9, 9>(xw, yw, prod);
    if constexpr (MOD::kBarrett)
        reduce_barrett_wide<MOD, 10, true>(prod, rv.w, qw);
    else
        reduce_wide<MOD, 10, true>(prod, rv.w, qw);
    if (qw[8] >> 5) err |= ERR_QUOTIENT_RANGE;  // q does not fit the gate's 9 q wires
    u32 q29[NL], r29[NL];
    split29(rv, r29);
    split29_9(qw, q29);
    u64 tmp[51];
    ArrEmit e{tmp, 0};
    uint8_t e2 = 0;
    emit_mul_rows<MOD>(e, x29, y29, q29, r29, e2);
    if (!err) err |= e2;
    if (err) {
        P2E_UNROLL
        for (int k = 0; k < 51; k++) tmp[k] = 0;
    }
    store_col(r, ld, i, tmp, NL);
    store_col(q, ld, i, tmp + 9, NL);
    store_col(cs, ld, i, tmp + 18, 17);
    store_col(b, ld, i, tmp + 35, 16);
    return err;
}

// gates/mul_nonnative.rs:513-531 on arbitrary inputs: true Goldilocks division by 2^29
P2E_HD uint8_t prim_checksum(const u64* a, u64* b, size_t ld, size_t i) {
    // 2^-29 mod p_gl = 2^163 = -2^67 = p - 8*(2^32-1)   (2^96 = -1, 2^64 = 2^32-1)
    const u64 inv = P_GL - 8ull * 0xFFFFFFFFull;
    uint8_t err = 0;
    u64 last = 0;
    P2E_UNROLL
    for (int k = 0; k < 2 * NL - 2; k++) {
        u64 ak = a[(size_t)k * ld + i];
        if (ak >= P_GL) ak -= P_GL;
        u64 bi = gl_mul(gl_add(ak, last), inv);
        u64 v = gl_add(bi, 1ull << 33);
        if (v >> 34) err |= ERR_CARRY_RANGE;
        b[(size_t)k * ld + i] = v;
        last = bi;
    }
    return err;
}

template <class MOD, bool IS_SUB>
P2E_HD uint8_t prim_addsub(const u64* a, const u64* b, u64* out, u64* ov, size_t ld, size_t i) {
    u64 al[NL], bl[NL];
    load_col(a, ld, i, al, NL);
    load_col(b, ld, i, bl, NL);
    U256 av, bv;
    uint8_t err = 0;
    if (!value_of_limbs(al, NL, av)) err |= ERR_VALUE_GE_2_256;
    if (!value_of_limbs(bl, NL, bv)) err |= ERR_VALUE_GE_2_256;
    u64 tmp[10];
    ArrEmit e{tmp, 0};
    if (IS_SUB)
        (void)wit_sub<MOD>(e, av, bv);
    else
        (void)wit_add<MOD>(e, av, bv);
    if (err) {
        P2E_UNROLL
        for (int k = 0; k < 10; k++) tmp[k] = 0;
    }
    store_col(out, ld, i, tmp, NL);
    ov[i] = tmp[9];
    return err;
}

// gadgets/nonnative.rs:696-728, k summands laid out [k][9][ld]
template <class MOD>
P2E_HD uint8_t prim_add_many(const u64* summands, int k, u64* out, u64* ov, size_t ld, size_t i) {
    uint8_t err = 0;
    u32 acc[9];
    P2E_UNROLL
    for (int j = 0; j < 9; j++) acc[j] = 0;
    for (int t = 0; t < k; t++) {
        u64 l[NL];
        load_col(summands + (size_t)t * NL * ld, ld, i, l, NL);
        U256 v;
        if (!value_of_limbs(l, NL, v)) err |= ERR_VALUE_GE_2_256;
        v = fe_canon<MOD>(v);
        u64 c = 0;
        P2E_UNROLL
        for (int j = 0; j < 9; j++) {
            c += (u64)acc[j] + (j < 8 ? v.w[j] : 0u);
            acc[j] = (u32)c;
            c >>= 32;
        }
    }
    u32 o = 0;
    for (int t = 0; t < k; t++) {
        bool ge = acc[8] != 0 || geq_mod<MOD>(acc);
        if (ge) {
            u32 br = 0;
            P2E_UNROLL
            for (int j = 0; j < 9; j++) {
                u64 d = (u64)acc[j] - (j < 8 ? MOD::m(j) : 0u) - br;
                acc[j] = (u32)d;
                br = (u32)(d >> 63);
            }
            o++;
        }
    }
    U256 s;
    P2E_UNROLL
    for (int j = 0; j < 8; j++) s.w[j] = err ? 0u : acc[j];
    u32 sl[NL];
    split29(s, sl);
    P2E_UNROLL
    for (int j = 0; j < NL; j++) out[(size_t)j * ld + i] = sl[j];
    ov[i] = err ? 0u : o;
    return err;
}

// gadgets/nonnative.rs:857-872
template <class MOD>
P2E_HD uint8_t prim_inv(const u64* x, u64* inv, u64* div, size_t ld, size_t i) {
    u64 xl[NL];
    load_col(x, ld, i, xl, NL);
    U256 xv;
    uint8_t err = 0;
    if (!value_of_limbs(xl, NL, xv)) err |= ERR_VALUE_GE_2_256;
    xv = fe_canon<MOD>(xv);
    u64 tmp[18];
    ArrEmit e{tmp, 0};
    (void)wit_inv<MOD>(e, xv, err);
    if (err) {
        P2E_UNROLL
        for (int k = 0; k < 18; k++) tmp[k] = 0;
    }
    store_col(inv, ld, i, tmp, NL);
    store_col(div, ld, i, tmp + 9, NL);
    return err;
}

// gadgets/glv.rs:128-142
P2E_HD uint8_t prim_glv(const u64* k, u64* k1, u64* k2, u64* n1, u64* n2, size_t ld, size_t i) {
    u64 kl[NL];
    load_col(k, ld, i, kl, NL);
    U256 kv;
    uint8_t err = 0;
    if (!value_of_limbs(kl, NL, kv)) err |= ERR_VALUE_GE_2_256;
    GlvOut g = glv_decompose(kv);
    u32 l1[NL], l2[NL];
    split29(g.k1, l1);
    split29(g.k2, l2);
    if (l1[5] | l1[6] | l1[7] | l1[8] | l2[5] | l2[6] | l2[7] | l2[8]) err |= ERR_LIMB_RANGE;
    P2E_UNROLL
    for (int j = 0; j < 5; j++) {
        k1[(size_t)j * ld + i] = err ? 0u : l1[j];
        k2[(size_t)j * ld + i] = err ? 0u : l2[j];
    }
    n1[i] = err ? 0u : g.n1;
    n2[i] = err ? 0u : g.n2;
    return err;
}

// gadgets/biguint.rs:508-518 BigUintDivRemGenerator (the 8th and last generator of the crate; used by rem_biguint /
// reduce, gadgets/nonnative.rs:539-548): a (na <= 18 limbs) = div * b + rem, b (nb <= 9 limbs).  Limb counts of the
// outputs as div_rem_biguint allocates them (:391-397): nd = na - nb + 1 (0 if nb > na + 1), nb.  Restoring binary
// division, one quotient bit per step, everything in registers (static indices only); not a hot path.
// Flags: a limb >= 2^29 (deviation: the reference would sum arbitrary field elements), b == 0 (BigUint::div_rem
// panics), and a quotient that set_biguint_target rejects -- INCLUDING the reference's quirk that convert_base hands it
// floor(32 D / 29) limbs for a D-digit value, zero limbs included, so e.g. a 290-bit quotient "does not fit" 10 limbs.
P2E_HD uint8_t prim_div_rem(const u64* a, int na, const u64* b, int nb, u64* div, u64* rem, size_t ld, size_t i) {
    uint8_t err = 0;
    u32 aw[17], bw[9];
    P2E_UNROLL
    for (int k = 0; k < 17; k++) aw[k] = 0;
    P2E_UNROLL
    for (int k = 0; k < 9; k++) bw[k] = 0;
    P2E_UNROLL
    for (int k = 0; k < 18; k++) {
        if (k < na) {
            const u64 l = a[(size_t)k * ld + i];
            if (l >> BITS) err |= ERR_LIMB_RANGE;
            const int bit = BITS * k, wi = bit >> 5, sh = bit & 31;
            const u64 v = (l & MASK29) << sh;
            aw[wi] |= (u32)v;
            if (wi + 1 < 17) aw[wi + 1] |= (u32)(v >> 32);
        }
    }
    u32 any_b = 0;
    P2E_UNROLL
    for (int k = 0; k < 9; k++) {
        if (k < nb) {
            const u64 l = b[(size_t)k * ld + i];
            if (l >> BITS) err |= ERR_LIMB_RANGE;
            const int bit = BITS * k, wi = bit >> 5, sh = bit & 31;
            const u64 v = (l & MASK29) << sh;
            bw[wi] |= (u32)v;
            if (wi + 1 < 9) bw[wi + 1] |= (u32)(v >> 32);
            any_b |= (u32)(l & MASK29);
        }
    }
    if (!any_b) err |= ERR_DIVISION_BY_ZERO;
    u32 r[9], q[17];
    P2E_UNROLL
    for (int k = 0; k < 9; k++) r[k] = 0;
    P2E_UNROLL
    for (int w = 16; w >= 0; w--) {
        u32 word = aw[w], qw = 0;
        for (int j = 0; j < 32; j++) {
            u32 carry = word >> 31;
            word <<= 1;
            P2E_UNROLL
            for (int k = 0; k < 9; k++) {   // r = (r << 1) | next bit of a      (r < b < 2^261: no overflow)
                const u32 nk = (r[k] << 1) | carry;
                carry = r[k] >> 31;
                r[k] = nk;
            }
            u32 t[9], borrow = 0;
            P2E_UNROLL
            for (int k = 0; k < 9; k++) {   // t = r - b
                const u64 d = (u64)r[k] - bw[k] - borrow;
                t[k] = (u32)d;
                borrow = (u32)(d >> 63);
            }
            const bool ge = borrow == 0;
            P2E_UNROLL
            for (int k = 0; k < 9; k++) r[k] = ge ? t[k] : r[k];
            qw = (qw << 1) | (ge ? 1u : 0u);
        }
        q[w] = qw;
    }
    const int nd = nb > na + 1 ? 0 : na - nb + 1;
    // limbs set_biguint_target would be handed for the quotient (see the header comment)
    int digits = 0;
    u32 top = 0;
    P2E_UNROLL
    for (int k = 0; k < 17; k++)
        if (q[k]) {
            digits = k + 1;
            top = q[k];
        }
    int bitlen = 0;
    if (digits) {
        int lz = 0;
        while (!(top >> 31)) {
            top <<= 1;
            lz++;
        }
        bitlen = 32 * digits - lz;
    }
    const int l0 = (32 * digits) / BITS;
    if (!err && l0 + (bitlen > BITS * l0 ? 1 : 0) > nd) err |= ERR_LIMB_RANGE;   // (the quotient by zero is garbage)
    P2E_UNROLL
    for (int k = 0; k < 18; k++) {
        if (k < nd) {
            const int bit = BITS * k, wi = bit >> 5, sh = bit & 31;
            const u64 lo = q[wi], hi = wi + 1 < 17 ? q[wi + 1] : 0;
            div[(size_t)k * ld + i] = err ? 0 : (((lo | (hi << 32)) >> sh) & MASK29);
        }
    }
    P2E_UNROLL
    for (int k = 0; k < 9; k++) {
        if (k < nb) {
            const int bit = BITS * k, wi = bit >> 5, sh = bit & 31;
            const u64 lo = r[wi], hi = wi + 1 < 9 ? r[wi + 1] : 0;
            rem[(size_t)k * ld + i] = err ? 0 : (((lo | (hi << 32)) >> sh) & MASK29);
        }
    }
    return err;
}

// gadgets/biguint.rs:27-51,454-463: 256-bit packed LE -> nine 29-bit limb columns
P2E_HD void prim_split(const uint8_t* packed, u64* limbs, size_t ld, size_t i) {
    U256 v = *reinterpret_cast<const U256*>(packed + 32 * i);
    u32 l[NL];
    split29(v, l);
    P2E_UNROLL
    for (int k = 0; k < NL; k++) limbs[(size_t)k * ld + i] = l[k];
}
// inverse direction (gadgets/biguint.rs:444-452), flags limbs >= 2^29 and values >= 2^256
P2E_HD uint8_t prim_pack(const u64* limbs, uint8_t* packed, size_t ld, size_t i) {
    u64 l[NL];
    load_col(limbs, ld, i, l, NL);
    uint8_t err = 0;
    P2E_UNROLL
    for (int k = 0; k < NL; k++)
        if (l[k] >> BITS) err |= ERR_LIMB_RANGE;
    U256 v;
    if (!value_of_limbs(l, NL, v)) err |= ERR_VALUE_GE_2_256;
    if (err) v = u256_zero();
    *reinterpret_cast<U256*>(packed + 32 * i) = v;
    return err;
}

}  // namespace p2e
