// Constraint-block columns: the values of the targets that the plonky2_ux U29 gates fill
// inside the constraint blocks of the non-native gadgets -- what add_biguint / sub_biguint / mul_biguint
// (gadgets/biguint.rs:240-323) receive back from add_many_ux, sub_ux, mul_ux, add_uxs_with_carry, the
// mul_biguint_by_bool products modulus * overflow (gadgets/biguint.rs:360-374) and the cmp_biguint result of a
// range-checked gadget -- in builder-call order, generator by generator:
//   add_nonnative      gadgets/nonnative.rs:262-273   add_biguint(a, b) | modulus * overflow | add_biguint(sum, .) [| cmp]
//   sub_nonnative      gadgets/nonnative.rs:373-386   add_biguint(diff, b) | modulus * overflow | sub_biguint(., .) [| cmp]
//   add_many_nonnative gadgets/nonnative.rs:330-351   the add_biguint fold | mul_biguint(modulus, [overflow]) | add_biguint [| cmp]
//   inv_nonnative      gadgets/nonnative.rs:518-530   mul_biguint(x, inv) | mul_biguint(modulus, div) | add_biguint(., one) [| cmp]
//   mul_nonnative      gadgets/nonnative.rs:462-463   [cmp] only (its relation lives in the two custom gates)
// 249 385 values per verify.  Every value is a function of limbs that already sit in the witness matrix, the aux
// matrix, the inputs or the circuit constants, so this is one more streaming pass: a lane owns one (signature,
// generator), loads its operands through the schedule's wiring table (schedule.hpp GenOp) and emits its block.
// [upstream-from-memory] plonky2_ux is not in the container: the gates are modelled by what they constrain
// (limb = value mod 2^29, carry = value >> 29; borrow-out of sub_ux in {0, 1}; list_le_ux_circuit = [a <= b]);
// oracle/check_circuit.py holds the same model and is what the tests compare with ("parity unpinned").
#pragma once
#include "aux.hpp"

namespace p2e {

enum UxKind : uint8_t { UX_ADD = 0, UX_SUB = 1, UX_ADD_MANY = 2, UX_MUL = 3, UX_INV = 4 };
constexpr u32 UX_COLS_ADD = 45, UX_COLS_SUB = 47, UX_COLS_ADD_MANY = 144, UX_COLS_INV = 434;   // + 1 with range_check

struct UxItem {
    uint8_t kind, field, nops, range_check;
    uint8_t nl[4];
    u32 src[4];
    u32 res_col;   // the generator's first witness column
    u32 ux_col, ncols;
};
struct UxArgs {
    const u64* cols;
    size_t ld;
    const u64* aux;
    size_t ald;
    void* ux;   // u64 or u32 column matrix, by the emitter type
    size_t uld, n;
    const uint8_t* in[5];   // packed inputs by INPUT_* slot: pk.y, pk.x, msg (glv_mul: k), r, s
    const U256* consts;     // [NUM_CONSTV]: circuit constants by id (AUX_SRC_CONST | id)
    const UxItem* items;
    u32* err;
};

// the limbs of a target (zero beyond the limbs it has); flags limbs that are no U29 values
P2E_HD void ux_load(const UxArgs& A, u32 src, int nl, size_t i, u32* l, bool& bad) {
    const u32 kind = src & AUX_SRC_KIND_MASK;
    if (kind == AUX_SRC_CONST) {
        split29(A.consts[src & 0xFFFFu], l);
    } else if (kind == AUX_SRC_INPUT) {
        const u32* p = reinterpret_cast<const u32*>(A.in[src & 7u] + 32 * i);
        U256 v;
        P2E_UNROLL
        for (int k = 0; k < 8; k++) v.w[k] = p[k];
        split29(v, l);
    } else {
        const u64* base = kind == AUX_SRC_AUX ? A.aux + (size_t)(src & ~AUX_SRC_KIND_MASK) * A.ald : A.cols + (size_t)src * A.ld;
        const size_t ld = kind == AUX_SRC_AUX ? A.ald : A.ld;
        P2E_UNROLL
        for (int k = 0; k < NL; k++) {
            const u64 v = k < nl ? base[(size_t)k * ld + i] : 0;
            bad = bad || (v >> BITS) != 0;
            l[k] = (u32)v;
        }
    }
    P2E_UNROLL
    for (int k = 0; k < NL; k++)
        if (k >= nl) l[k] = 0;
}
template <class MOD>
P2E_HD u32 ux_m29(int k) {
    return k < NL ? MOD::m29(k) : 0u;
}

// add_biguint(a[NA], b[NB]) gadgets/biguint.rs:240-270: max(NA, NB) add_many_ux([carry, a_i, b_i]) -> (limb, carry);
// out gets max(NA, NB) + 1 limbs
template <int NA, int NB, class E>
P2E_HD void ux_add_biguint(E& e, const u32* a, const u32* b, u32* out) {
    constexpr int N = NA > NB ? NA : NB;
    u32 carry = 0;
    P2E_UNROLL
    for (int k = 0; k < N; k++) {
        const u32 s = carry + (k < NA ? a[k] : 0u) + (k < NB ? b[k] : 0u);
        out[k] = s & MASK29;
        carry = s >> BITS;
        e.put(out[k]);
        e.put(carry);
    }
    out[N] = carry;
}
// sub_biguint(a[N], b[N]) gadgets/biguint.rs:272-293: N sub_ux -> (limb, borrow)
template <int N, class E>
P2E_HD void ux_sub_biguint(E& e, const u32* a, const u32* b) {
    u32 borrow = 0;
    P2E_UNROLL
    for (int k = 0; k < N; k++) {
        const i64 d = (i64)a[k] - (i64)b[k] - (i64)borrow;
        borrow = d < 0 ? 1u : 0u;
        e.put((u64)(d + ((i64)borrow << BITS)));
        e.put(borrow);
    }
}
// mul_biguint(a[NA], b[NB]) gadgets/biguint.rs:295-323: NA * NB mul_ux -> (product, carry), then NA + NB
// add_uxs_with_carry over the columns; out gets NA + NB + 1 limbs
template <int NA, int NB, class E>
P2E_HD void ux_mul_biguint(E& e, const u32* a, const u32* b, u32* out) {
    u64 col[NA + NB];
    P2E_UNROLL
    for (int k = 0; k < NA + NB; k++) col[k] = 0;
    P2E_UNROLL
    for (int x = 0; x < NA; x++) {
        P2E_UNROLL
        for (int y = 0; y < NB; y++) {
            const u64 p = (u64)a[x] * b[y];
            const u32 lo = (u32)p & MASK29, hi = (u32)(p >> BITS);
            e.put(lo);
            e.put(hi);
            col[x + y] += lo;
            col[x + y + 1] += hi;
        }
    }
    u64 carry = 0;
    P2E_UNROLL
    for (int k = 0; k < NA + NB; k++) {
        const u64 s = col[k] + carry;
        out[k] = (u32)s & MASK29;
        carry = s >> BITS;
        e.put(out[k]);
        e.put(carry);
    }
    out[NA + NB] = (u32)carry;
}
// cmp_biguint(x, modulus) = list_le_ux_circuit: [x <= m] on 9 limbs, least significant first
template <class MOD>
P2E_HD u32 ux_le_modulus(const u32* x) {
    bool gt = false, eq = true;
    P2E_UNROLL
    for (int k = NL - 1; k >= 0; k--) {
        const u32 m = MOD::m29(k);
        gt = gt || (eq && x[k] > m);
        eq = eq && x[k] == m;
    }
    return gt ? 0u : 1u;
}

// the block's last value (the cmp_biguint result of a range-checked gadget) and the emitter's flush, inside every
// branch of ux_block: the paired-store emitters keep their cursor bookkeeping compile-time only along straight lines
template <class MOD, class E>
P2E_HD void ux_finish(E& e, const UxItem& it, const u32* res) {
    if (it.range_check) {
        e.put(ux_le_modulus<MOD>(res));
        e.flush();
    } else {
        e.flush();
    }
}
template <class MOD, class E>
P2E_HD void ux_block(E e, const UxArgs& A, const UxItem& it, size_t i, bool& bad) {
    u32 m[NL];
    P2E_UNROLL
    for (int k = 0; k < NL; k++) m[k] = MOD::m29(k);
    u32 res[NL];
    ux_load(A, it.res_col, NL, i, res, bad);
    if (it.kind == UX_ADD || it.kind == UX_SUB) {
        u32 a[NL], b[NL], t[NL + 2], mto[NL + 1];
        ux_load(A, it.src[0], it.nl[0], i, a, bad);
        ux_load(A, it.src[1], it.nl[1], i, b, bad);
        const u64 ov = A.cols[(size_t)(it.res_col + NL) * A.ld + i];
        if (it.kind == UX_ADD) {
            ux_add_biguint<NL, NL>(e, a, b, t);                      // sum_expected (limbs beyond an operand's count are zero_ux)
            P2E_UNROLL
            for (int k = 0; k < NL; k++) {                           // mul_biguint_by_bool(modulus, overflow): field products
                mto[k] = (u32)gl_mul(m[k], ov);
                e.put(mto[k]);
            }
            ux_add_biguint<NL, NL>(e, res, mto, t);                  // sum_actual
            ux_finish<MOD>(e, it, res);
        } else {
            ux_add_biguint<NL, NL>(e, res, b, t);                    // diff_plus_b: 10 limbs
            P2E_UNROLL
            for (int k = 0; k < NL; k++) {
                mto[k] = (u32)gl_mul(m[k], ov);
                e.put(mto[k]);
            }
            mto[NL] = 0;                                             // pad_biguints
            ux_sub_biguint<NL + 1>(e, t, mto);
            ux_finish<MOD>(e, it, res);
        }
    } else if (it.kind == UX_ADD_MANY) {
        // the fold over the summands from zero_biguint() (no limbs): accumulator grows by one limb per add
        u32 x0[NL], x1[NL], x2[NL], x3[NL], a1[NL + 1], a2[NL + 2], a3[NL + 3], a4[NL + 4], z0[1] = {0};
        ux_load(A, it.src[0], it.nl[0], i, x0, bad);
        ux_load(A, it.src[1], it.nl[1], i, x1, bad);
        ux_load(A, it.src[2], it.nl[2], i, x2, bad);
        ux_load(A, it.src[3], it.nl[3], i, x3, bad);
        ux_add_biguint<0, NL>(e, z0, x0, a1);
        ux_add_biguint<NL + 1, NL>(e, a1, x1, a2);
        ux_add_biguint<NL + 2, NL>(e, a2, x2, a3);
        ux_add_biguint<NL + 3, NL>(e, a3, x3, a4);
        u32 ovl[1] = {(u32)A.cols[(size_t)(it.res_col + NL) * A.ld + i]}, mto[NL + 2], t[NL + 3];
        ux_mul_biguint<NL, 1>(e, m, ovl, mto);                       // 11 limbs
        ux_add_biguint<NL, NL + 2>(e, res, mto, t);
        ux_finish<MOD>(e, it, res);
    } else if (it.kind == UX_INV) {
        u32 x[NL], div[NL], prod[2 * NL + 1], mtd[2 * NL + 1], one[1] = {1}, t[2 * NL + 2];
        ux_load(A, it.src[0], it.nl[0], i, x, bad);
        ux_load(A, it.res_col + NL, NL, i, div, bad);
        ux_mul_biguint<NL, NL>(e, x, res, prod);                     // x * inv
        ux_mul_biguint<NL, NL>(e, m, div, mtd);                      // modulus * div
        ux_add_biguint<2 * NL + 1, 1>(e, mtd, one, t);               // + 1
        ux_finish<MOD>(e, it, res);
    } else {                                                         // UX_MUL: biguint_to_nonnative(r, range_check)
        ux_finish<MOD>(e, it, res);
    }
}

template <class E>
P2E_HD void body_ux(const UxArgs& A, int item, size_t i) {
    const UxItem it = A.items[item];
    E e = E::at(static_cast<typename E::elem*>(A.ux), A.uld, i, it.ux_col);
    bool bad = false;
    if (it.field == 0)
        ux_block<ModP>(e, A, it, i, bad);
    else
        ux_block<ModN>(e, A, it, i, bad);
    if (bad) err_or(&A.err[i], ERR_LIMB_RANGE);
}

// curve programs (curves.hpp): field 0 / 1 = secp256k1 base / scalar, 2 / 3 = P-256 base / scalar; A.consts is the
// program's constant array indexed by source id (points at 2c / 2c + 1, scalar constants from AUX_GCONST_BASE)
template <class E>
P2E_HD void body_ux_cv(const UxArgs& A, int item, size_t i) {
    const UxItem it = A.items[item];
    E e = E::at(static_cast<typename E::elem*>(A.ux), A.uld, i, it.ux_col);
    bool bad = false;
    if (it.field == 0)
        ux_block<ModP>(e, A, it, i, bad);
    else if (it.field == 1)
        ux_block<ModN>(e, A, it, i, bad);
    else if (it.field == 2)
        ux_block<ModP256>(e, A, it, i, bad);
    else
        ux_block<ModN256>(e, A, it, i, bad);
    if (bad) err_or(&A.err[i], ERR_LIMB_RANGE);
}

}  // namespace p2e
