// Built-in-generator columns: the values of the targets that plonky2's OWN generators
// fill on the ECDSA path, i.e. everything between the hot-path generators that is not a constant or an input:
//   split_le_base bits of a scalar's limbs + the 4-/2-bit digits built from them   gadgets/split_nonnative.rs:25-72
//   is_equal(digit, zero), not(.), the random_access_curve_points selection        gadgets/curve_fixed_base.rs:56-61,
//                                                                                  gadgets/curve_msm.rs:67-71,
//                                                                                  gadgets/curve_windowed_mul.rs:74-118
//   not(b) and the four mul_biguint_by_bool products of curve_conditional_add      gadgets/curve.rs:225-243,
//                                                                                  gadgets/biguint.rs:360-374
//   not(b) and the two products of nonnative_conditional_neg                       gadgets/nonnative.rs:584-596
// They are emitted in the order the gadgets create them, as a second column-major matrix aux[col * ld + sig]
// (8 959 columns per verify, 4 738 per glv_mul).  Every value is a function of the hot-path columns, the
// constant tables and the caller's pk.y, so this is a streaming pass over the finished witness matrix: one
// (signature, item) per lane, coalesced column loads and stores, HBM-bound.  Wire placement of these targets
// (BaseSumGate / RandomAccessGate / ArithmeticGate rows) is the host's business: the plonky2 sources that fix it
// are not in the container.
#pragma once
#include "pipeline.hpp"

namespace p2e {

// where the limbs of a target live
constexpr u32 AUX_SRC_NONE = 0xFFFFFFFFu;
constexpr u32 AUX_SRC_CONST = 0x80000000u;     // | constant id: 2 * constant point id + (0: x, 1: y), or a CONSTV_* scalar constant
constexpr u32 AUX_SRC_INPUT = 0x40000000u;     // | INPUT_*: a caller's packed 32-byte input (a 9-limb virtual target)
constexpr u32 AUX_SRC_INPUT_PY = 0x40000000u;  // the caller's pk.y
constexpr u32 AUX_SRC_AUX = 0x20000000u;       // | column of the built-in-generator (aux) matrix
constexpr u32 AUX_SRC_KIND_MASK = 0xE0000000u;
constexpr u32 INPUT_PY = 0, INPUT_PX = 1, INPUT_MSG = 2, INPUT_R = 3, INPUT_S = 4;   // Buffers / p2e.h argument slots
// scalar constants the gadgets create (constant_nonnative): zero has NO limbs (Q5), B = 7 one
constexpr u32 CONSTV_ZERO = 6, CONSTV_B7 = 7, CONSTV_GLV_S = 8, CONSTV_GLV_BETA = 9, NUM_CONSTV = 10;
// columns of the result limbs inside the column block of one curve op (gadgets/curve.rs:160-243 emission order)
constexpr u32 COL_ADD_X3 = 150, COL_ADD_Y3 = 221, COL_DBL_X3 = 201, COL_DBL_Y3 = 272, COL_CADD_X = 231, COL_CADD_Y = 241;

// AUX_CP_*: the curve programs' gadgets (curves.hpp): a window of curve_scalar_mul_windowed, the bit split and one bit
// of curve_scalar_mul
enum AuxKind : uint8_t { AUX_SPLIT4 = 0, AUX_SPLIT2 = 1, AUX_FBWIN = 2, AUX_MSMDIG = 3, AUX_CNEG = 4, AUX_CP_WINDOW = 5, AUX_CP_BITS = 6,
                         AUX_CP_BIT = 7 };
// scalar constants of a curve program: source code AUX_SRC_CONST | (AUX_GCONST_BASE + j); below it 2c / 2c + 1 = x / y
// of constant point c
constexpr u32 AUX_GCONST_BASE = 64;

struct AuxItem {
    uint8_t kind;      // AuxKind
    uint8_t nlx, nly;  // limbs of p1.x / p1.y (window items), of x (AUX_CNEG), of the scalar (splits)
    uint8_t pad;
    u32 aux_col, ncols;  // this item's aux columns
    u32 a, b, c;         // splits: a = scalar limb column.  windows: a (and c) = digit scalars' limb columns,
                         // b = window / digit index.  AUX_CNEG: a = column of the bool
    u32 sumx, sumy;      // windows: limb columns of curve_add's result; AUX_CNEG: sumx = limb column of neg
    u32 p1x, p1y;        // limbs of p1 (AUX_CNEG: p1x = limbs of x): column or AUX_SRC_* code
};
struct AuxTables {
    u32 tabx[16], taby[16];  // curve_msm_circuit's precomputation[i]: limb columns (or AUX_SRC_CONST)
    u32 num_aux_cols;
};
struct AuxArgs {
    const u64* cols;  // finished witness matrix, or null: read the compact container's narrow matrix instead (every
    size_t ld;        // column this pass reads is a limb or a flag)
    void* aux;        // u64 or u32 column matrix, by the emitter type
    size_t ald, n;
    const uint8_t* py;  // pk.y, packed
    const Aff* cpts;
    const Aff* fbtab;
    const AuxItem* items;
    const AuxTables* tab;
    u32* err;
    const u32* nar;          // compact source: narrow matrix, its stride, wide columns before column c
    size_t ldn;
    const u32* wide_before;
    const uint8_t* in[5];    // curve programs: every packed input by INPUT_* slot (in[INPUT_PY] == py)
};

P2E_HD u64 aux_col(const AuxArgs& A, u32 c, size_t i) {
    if (A.cols) return A.cols[(size_t)c * A.ld + i];
    return A.nar[(size_t)(c - A.wide_before[c]) * A.ldn + i];
}
P2E_HD void aux_limbs_of(const U256& v, u64* l) {
    u32 s[NL];
    split29(v, s);
    P2E_UNROLL
    for (int k = 0; k < NL; k++) l[k] = s[k];
}
// the (zero padded) 9 limbs of a target
P2E_HD void aux_load_limbs(const AuxArgs& A, u32 src, size_t i, u64* l, int nl) {
    if (src & AUX_SRC_CONST) {
        const u32 id = src & 0xFFFFu;
        const Aff a = A.cpts[id >> 1];
        aux_limbs_of(u256_select((id & 1) != 0, a.y, a.x), l);   // (a struct-level ?: would live in scratch)
    } else if (src == AUX_SRC_INPUT_PY) {
        const u32* p = reinterpret_cast<const u32*>(A.py + 32 * i);
        U256 v;
        P2E_UNROLL
        for (int k = 0; k < 8; k++) v.w[k] = p[k];
        aux_limbs_of(v, l);
    } else {
        P2E_UNROLL
        for (int k = 0; k < NL; k++) l[k] = k < nl ? aux_col(A, src + (u32)k, i) : 0;
    }
}
// digit t (WB bits) of a scalar given as nl limb columns: bits of each 29-bit limb little-endian, limb-major,
// zero padded (gadgets/split_nonnative.rs:33-38,60-65); a digit may straddle two limbs
template <int WB>
P2E_HD u32 aux_digit(const AuxArgs& A, u32 col, int nl, int t, size_t i) {
    const int bit = WB * t, li = bit / BITS, sh = bit % BITS;
    u64 lo = li < nl ? aux_col(A, col + (u32)li, i) : 0;
    u32 d = (u32)(lo >> sh);
    if (sh + WB > BITS) {
        u64 hi = li + 1 < nl ? aux_col(A, col + (u32)li + 1, i) : 0;
        d |= (u32)hi << (BITS - sh);
    }
    return d & ((1u << WB) - 1);
}
// mul_biguint_by_bool gadgets/biguint.rs:360-374 (limbs are canonical Goldilocks values, b is 0 or 1)
template <class E>
P2E_HD void aux_put_select(E& e, const u64* l, int nl, u64 b) {
    P2E_UNROLL
    for (int k = 0; k < NL; k++)
        if (k < nl) e.put(b ? l[k] : 0);
}
// not(b), then the four products of curve_conditional_add gadgets/curve.rs:233-238.  nlx, nly are literal NL on
// the common path (only a constant p1 can be shorter), which keeps the paired-store bookkeeping compile-time.
template <class E>
P2E_HD void aux_put_cond_add(E& e, const AuxArgs& A, const AuxItem& it, size_t i, u64 b, int nlx, int nly) {
    const u64 not_b = 1 - b;
    e.put(not_b);
    u64 l[NL];
    aux_load_limbs(A, it.sumx, i, l, NL);
    aux_put_select(e, l, NL, b);
    aux_load_limbs(A, it.sumy, i, l, NL);
    aux_put_select(e, l, NL, b);
    aux_load_limbs(A, it.p1x, i, l, nlx);
    aux_put_select(e, l, nlx, not_b);
    aux_load_limbs(A, it.p1y, i, l, nly);
    aux_put_select(e, l, nly, not_b);
}
template <class E>
P2E_HD void aux_put_cond_add(E& e, const AuxArgs& A, const AuxItem& it, size_t i, u64 b) {
    if (it.nlx == NL && it.nly == NL)
        aux_put_cond_add(e, A, it, i, b, NL, NL);
    else
        aux_put_cond_add(e, A, it, i, b, (int)it.nlx, (int)it.nly);
}
// split_nonnative_to_{4,2}_bit_limbs of a scalar with NLS limbs (gadgets/split_nonnative.rs:25-72)
template <class E, int NLS, int WB>
P2E_HD void aux_put_split(E& e, const AuxArgs& A, const AuxItem& it, size_t i) {
    u64 l[NL];
    aux_load_limbs(A, it.a, i, l, NLS);
    bool bad = false;
    P2E_UNROLL
    for (int k = 0; k < NLS; k++) {
        bad = bad || (l[k] >> BITS) != 0;   // split_le_base(limb, 29) has no witness for a wider limb
        P2E_UNROLL
        for (int j = 0; j < BITS; j++) e.put((l[k] >> j) & 1);
    }
    if (bad) err_or(&A.err[i], ERR_LIMB_RANGE);
    constexpr int nbits = NLS * BITS;
    P2E_UNROLL
    for (int t = 0; WB * t < nbits; t++) {
        // digit t from the limbs already in registers (zero padded above the last limb)
        const int bit = WB * t, li = bit / BITS, sh = bit % BITS;
        u32 d = (u32)(l[li] >> sh);
        if (sh + WB > BITS && li + 1 < NLS) d |= (u32)l[li + 1] << (BITS - sh);
        d &= (1u << WB) - 1;
        if (WB == 4) {
            e.put(d & 3);    // lower = mul_add(b, two, a)
            e.put(d >> 2);   // upper = mul_add(d, two, c)
        }
        e.put(d);            // mul_add(upper, four, lower) / mul_add(b, two, a)
    }
}

// ---- gate-internal values of the built-in gates on the path (the rest of SURVEY.md 8(f) rank 1) ------------------
// [upstream-from-memory: plonky2 gadgets/arithmetic.rs is_equal + EqualityGenerator, gates/random_access.rs
// RandomAccessGenerator]  Per window, in call order:
//   is_equal(index, zero)   not_equal = 1 - equal, inv = index^-1 in Goldilocks (0 if index == 0), diff = index,
//                           not_equal_check = diff * inv, diff_normalized = diff * not_equal_check          (5 values)
//   random_access x 18      one RandomAccessGate op per selected limb (9 x, 9 y): the 4 bit wires of the access
//                           index, least significant first                                                   (72 values)
// fixed-base windows call is_equal first (gadgets/curve_fixed_base.rs:57-60), MSM digits random_access first
// (gadgets/curve_msm.rs:68-70).  Everything is a function of the window's index, which is an aux column.
constexpr u32 GATE_COLS_PER_WINDOW = 5 + 2 * NL * 4;
struct GateItem {
    u32 idx_col;    // aux column holding the window's index (4-bit digit / 4*m + n)
    u32 gate_col;   // first column of this window's block
    u32 ra_first;   // 1: random_access before is_equal (MSM digits)
};
struct GateArgs {
    const u64* aux;
    size_t ald;
    u64* gate;
    size_t gld, n;
    const GateItem* items;
    u64 inv16[16];  // Goldilocks inverses of 0..15 (inv16[0] = 0)
};
template <class E>
P2E_HD void gate_put_eq(E& e, u32 idx, const u64* inv16) {
    const u64 ne = idx != 0;
    e.put(ne);          // not(equal)
    e.put(inv16[idx]);  // EqualityGenerator inv
    e.put(idx);         // diff = x - zero
    e.put(ne);          // not_equal_check = diff * inv
    e.put(idx);         // diff_normalized = diff * not_equal_check
}
template <class E>
P2E_HD void gate_put_ra(E& e, u32 idx) {
    P2E_UNROLL
    for (int k = 0; k < 2 * NL; k++) {
        P2E_UNROLL
        for (int b = 0; b < 4; b++) e.put((idx >> b) & 1u);
    }
}
template <class E>
P2E_HD void body_gate(const GateArgs& A, int item, size_t i) {
    const GateItem it = A.items[item];
    const u32 idx = (u32)A.aux[(size_t)it.idx_col * A.ald + i] & 15u;
    // the 16 inverses: a register select chain would be long; they sit in the kernel arguments (scalar loads)
    E e = E::at(A.gate, A.gld, i, it.gate_col);
    if (it.ra_first) {
        gate_put_ra(e, idx);
        gate_put_eq(e, idx, A.inv16);
        e.flush();
    } else {
        gate_put_eq(e, idx, A.inv16);
        gate_put_ra(e, idx);
        e.flush();
    }
}

template <class E>
P2E_HD void body_aux(const AuxArgs& A, int item, size_t i) {
    const AuxItem it = A.items[item];
    E e = E::at(static_cast<typename E::elem*>(A.aux), A.ald, i, it.aux_col);
    if (it.kind == AUX_SPLIT4) {         // the scalar of fixed_base_curve_mul_circuit: 9 limbs
        aux_put_split<E, NL, 4>(e, A, it, i);
    } else if (it.kind == AUX_SPLIT2) {  // k1, k2 of the GLV decomposition: 5 limbs
        aux_put_split<E, 5, 2>(e, A, it, i);
    } else if (it.kind == AUX_FBWIN) {   // gadgets/curve_fixed_base.rs:56-61
        const u32 d = aux_digit<4>(A, it.a, NL, (int)it.b, i);
        const u64 is_zero = d == 0, should_add = 1 - is_zero;
        e.put(is_zero);
        e.put(should_add);
        const Aff r = A.fbtab[it.b * 16 + d];   // slot 0 holds a copy of slot 1 (:56)
        u64 l[NL];
        aux_limbs_of(r.x, l);
        aux_put_select(e, l, NL, 1);
        aux_limbs_of(r.y, l);
        aux_put_select(e, l, NL, 1);
        aux_put_cond_add(e, A, it, i, should_add);
    } else if (it.kind == AUX_MSMDIG) {  // gadgets/curve_msm.rs:67-71
        const u32 idx = 4 * aux_digit<2>(A, it.c, 5, (int)it.b, i) + aux_digit<2>(A, it.a, 5, (int)it.b, i);
        e.put(idx);
        u64 l[NL];
        aux_load_limbs(A, A.tab->tabx[idx], i, l, NL);
        aux_put_select(e, l, NL, 1);
        aux_load_limbs(A, A.tab->taby[idx], i, l, NL);
        aux_put_select(e, l, NL, 1);
        const u64 is_zero = idx == 0, should_add = 1 - is_zero;
        e.put(is_zero);
        e.put(should_add);
        aux_put_cond_add(e, A, it, i, should_add);
    } else {                             // AUX_CNEG gadgets/nonnative.rs:584-596
        const u64 b = aux_col(A, it.a, i), not_b = 1 - b;
        e.put(not_b);
        u64 l[NL];
        aux_load_limbs(A, it.sumx, i, l, NL);
        aux_put_select(e, l, NL, b);
        if (it.nlx == NL) {
            aux_load_limbs(A, it.p1x, i, l, NL);
            aux_put_select(e, l, NL, not_b);
        } else {
            aux_load_limbs(A, it.p1x, i, l, it.nlx);
            aux_put_select(e, l, it.nlx, not_b);
        }
    }
    e.flush();
}

// ---- curve programs (curves.hpp): targets may also be caller inputs other than pk.y (the stand-alone multiplications
// take their scalar as an input), tables are per program ----------------------------------------------------------------
P2E_HD void aux_load_limbs_cv(const AuxArgs& A, u32 src, size_t i, u64* l, int nl) {
    if ((src & AUX_SRC_KIND_MASK) == AUX_SRC_INPUT) {
        const u32* p = reinterpret_cast<const u32*>(A.in[src & 7u] + 32 * i);
        U256 v;
        P2E_UNROLL
        for (int k = 0; k < 8; k++) v.w[k] = p[k];
        aux_limbs_of(v, l);
        P2E_UNROLL
        for (int k = 0; k < NL; k++)
            if (k >= nl) l[k] = 0;
    } else if (src & AUX_SRC_CONST) {
        const u32 id = src & 0xFFFFu;
        const Aff a = A.cpts[id >> 1];
        aux_limbs_of(u256_select((id & 1) != 0, a.y, a.x), l);
        P2E_UNROLL
        for (int k = 0; k < NL; k++)
            if (k >= nl) l[k] = 0;   // (the virtual `result` of curve_scalar_mul carries the constant in 9 limbs: nl = 9)
    } else {
        P2E_UNROLL
        for (int k = 0; k < NL; k++) l[k] = k < nl ? aux_col(A, src + (u32)k, i) : 0;
    }
}
template <class E>
P2E_HD void aux_put_products_cv(E& e, const AuxArgs& A, u32 src, int nl, size_t i, u64 b) {
    u64 l[NL];
    aux_load_limbs_cv(A, src, i, l, nl);
    if (nl == NL)
        aux_put_select(e, l, NL, b);
    else
        aux_put_select(e, l, nl, b);
}
template <int WB>
P2E_HD u32 aux_digit_of_limbs(const u64* l, int t) {
    const int bit = WB * t, li = bit / BITS, sh = bit % BITS;
    u32 d = li < NL ? (u32)(l[li] >> sh) : 0u;
    if (sh + WB > BITS && li + 1 < NL) d |= (u32)l[li + 1] << (BITS - sh);
    return d & ((1u << WB) - 1);
}
template <class E>
P2E_HD void body_aux_cv(const AuxArgs& A, int item, size_t i) {
    const AuxItem it = A.items[item];
    E e = E::at(static_cast<typename E::elem*>(A.aux), A.ald, i, it.aux_col);
    u64 l[NL];
    if (it.kind == AUX_SPLIT4 || it.kind == AUX_CP_BITS) {   // gadgets/split_nonnative.rs:25-50 / gadgets/nonnative.rs:566-582
        aux_load_limbs_cv(A, it.a, i, l, NL);
        bool bad = false;
        P2E_UNROLL
        for (int k = 0; k < NL; k++) {
            bad = bad || (l[k] >> BITS) != 0;
            P2E_UNROLL
            for (int j = 0; j < BITS; j++) e.put((l[k] >> j) & 1);
        }
        if (bad) err_or(&A.err[i], ERR_LIMB_RANGE);
        if (it.kind == AUX_SPLIT4) {
            P2E_UNROLL
            for (int t = 0; 4 * t < NL * BITS; t++) {
                const u32 d = aux_digit_of_limbs<4>(l, t);
                e.put(d & 3);
                e.put(d >> 2);
                e.put(d);
            }
            e.flush();
        } else {
            e.flush();
        }
    } else if (it.kind == AUX_FBWIN) {   // gadgets/curve_fixed_base.rs:56-61 (the scalar is a witness target: u1)
        aux_load_limbs_cv(A, it.a, i, l, NL);
        const u32 d = aux_digit_of_limbs<4>(l, (int)it.b);
        const u64 is_zero = d == 0, should_add = 1 - is_zero;
        e.put(is_zero);
        e.put(should_add);
        const Aff r = A.fbtab[it.b * 16 + d];
        aux_limbs_of(r.x, l);
        aux_put_select(e, l, NL, 1);
        aux_limbs_of(r.y, l);
        aux_put_select(e, l, NL, 1);
        e.put(is_zero);   // not(should_add)
        aux_put_products_cv(e, A, it.sumx, NL, i, should_add);
        aux_put_products_cv(e, A, it.sumy, NL, i, should_add);
        aux_put_products_cv(e, A, it.p1x, (int)it.nlx, i, is_zero);
        aux_put_products_cv(e, A, it.p1y, (int)it.nly, i, is_zero);
        e.flush();
    } else if (it.kind == AUX_CP_WINDOW) {   // gadgets/curve_windowed_mul.rs:159-163: random access, is_equal, not, conditional add
        aux_load_limbs_cv(A, it.a, i, l, NL);
        const u32 d = aux_digit_of_limbs<4>(l, (int)it.b);
        aux_load_limbs_cv(A, A.tab->tabx[d], i, l, NL);
        aux_put_select(e, l, NL, 1);
        aux_load_limbs_cv(A, A.tab->taby[d], i, l, NL);
        aux_put_select(e, l, NL, 1);
        const u64 is_zero = d == 0, should_add = 1 - is_zero;
        e.put(is_zero);
        e.put(should_add);
        e.put(is_zero);   // not(should_add)
        aux_put_products_cv(e, A, it.sumx, NL, i, should_add);
        aux_put_products_cv(e, A, it.sumy, NL, i, should_add);
        aux_put_products_cv(e, A, it.p1x, (int)it.nlx, i, is_zero);
        aux_put_products_cv(e, A, it.p1y, (int)it.nly, i, is_zero);
        e.flush();
    } else {   // AUX_CP_BIT gadgets/curve.rs:258-265: not(bit), sum.x * bit, result.x * not_bit, sum.y * bit, result.y * not_bit
        aux_load_limbs_cv(A, it.a, i, l, NL);
        const u64 bit = aux_digit_of_limbs<1>(l, (int)it.b), not_bit = 1 - bit;
        e.put(not_bit);
        aux_put_products_cv(e, A, it.sumx, NL, i, bit);
        aux_put_products_cv(e, A, it.p1x, NL, i, not_bit);
        aux_put_products_cv(e, A, it.sumy, NL, i, bit);
        aux_put_products_cv(e, A, it.p1y, NL, i, not_bit);
        e.flush();
    }
}

}  // namespace p2e
