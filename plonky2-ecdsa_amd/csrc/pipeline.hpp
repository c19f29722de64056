// Per-lane bodies of the batch witness pipeline for verify_secp256k1_message_circuit and glv_mul
// (reference gadgets/ecdsa.rs:30-53, gadgets/glv.rs:87-104, gadgets/curve_msm.rs:21-79,
// gadgets/curve_fixed_base.rs:18-66, gadgets/curve.rs:123-243).
//
// The reference's schedule is a fixed list of "curve ops" (curve_add / curve_double /
// curve_conditional_add), built once on the host by schedule.hpp into an OpDesc table.  The pipeline
// runs four phases over a batch, every lane = one signature (64 consecutive signatures per wave):
//   S  scalar      s^-1, u1, u2, curve_assert_valid, GLV decomposition, window digits        (1 thread/sig)
//   A  chains      every intermediate point in Jacobian form, no inversions (ec.hpp)          (1 thread/sig/chain)
//   B  batch-inv   Montgomery batch inversion of all Z -> affine points + every v^-1          (1 thread/sig/chunk)
//   C  expand      the 231/282/251 witness columns of every curve op, fully parallel          (1 thread/sig/op)
// Phase C writes 98% of the output bytes and is the HBM-bound kernel the roofline is quoted on.
#pragma once
#include "ec29.hpp"
#include "wit.hpp"

namespace p2e {

// ---- schedule description shared by host and device ---------------------------------------------------
enum OpKind : uint8_t { OP_ADD = 0, OP_DBL = 1, OP_CADD = 2 };
// R_SELSLOT (curve programs, curves.hpp): operand 2 of a conditional add = the result of op (id & 0xFFF), selected by
// the bit in digit row (id >> 12) -- curve_scalar_mul's  result + 2^i p  (gadgets/curve.rs:257-270)
enum RefKind : uint32_t { R_SLOT = 0, R_CONST = 1, R_DYN = 2, R_FBTAB = 3, R_MSMTAB = 4, R_SELSLOT = 5 };
// F_NO_AFFINE: the op's result is only ever consumed inside an expansion run (run-walked in registers by
// k_expand_runs), so phase B skips its Jacobian -> affine conversion
enum OpFlags : uint8_t { F_Z1ONE = 1, F_Z2ONE = 2, F_CHECK_R = 4, F_NO_AFFINE = 8 };
P2E_HD constexpr u32 make_ref(u32 kind, u32 id) { return (kind << 24) | id; }
P2E_HD constexpr u32 ref_kind(u32 r) { return r >> 24; }
P2E_HD constexpr u32 ref_id(u32 r) { return r & 0xFFFFFFu; }
constexpr uint16_t DYN_CONST_BIT = 0x8000;
// encoding of src[]: bit 15 constant point, bit 14 fixed-base table entry (id = window * 16 + digit),
// bit 13 (operand 2 of a conditional add only) the select bit "digit != 0"
constexpr uint16_t SRC_FB_BIT = 0x4000, SRC_SEL_BIT = 0x2000, SRC_ID_MASK = 0x1FFF;

struct OpDesc {
    uint8_t kind;       // OpKind
    uint8_t flags;      // OpFlags
    uint16_t cadd_idx;  // for OP_CADD: row of the dyn[] array that receives the selected source
    u32 ref1, ref2;     // operands
    u32 col;            // first output column
};
static_assert(sizeof(OpDesc) == 16, "load_op reads a descriptor as one 16-byte word");
// One descriptor of the op table.  The table is written once at context creation and never while a kernel runs:
// on the device it is read through the CONSTANT address space, which lets a wave-uniform index become one
// s_load_dwordx4 -- through the plain pointer the compiler cannot rule out that the kernel's own scratch stores alias
// the table and reads every field with a separate vector load behind the stores (vmcnt is in-order on gfx9, so
// each op then waited for the previous op's stores to drain).
P2E_HD OpDesc load_op(const OpDesc* ops, int t) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef const u32x4 __attribute__((address_space(4))) * cptr;
    const u32x4 v = ((cptr)(uintptr_t)ops)[__builtin_amdgcn_readfirstlane(t)];   // every caller's t is wave-uniform
    u32 x = v.x, y = v.y, z = v.z, w = v.w;
    // keep it ONE scalar 16-byte load: without this the compiler narrows the access to the bytes a caller uses, and a
    // sub-dword load cannot be scalar on gfx9 -- it becomes a vector load whose wait drains everything in flight
    asm volatile("" : "+s"(x), "+s"(y), "+s"(z), "+s"(w));
    OpDesc d;
    d.kind = (uint8_t)(x & 0xFFu);
    d.flags = (uint8_t)((x >> 8) & 0xFFu);
    d.cadd_idx = (uint16_t)(x >> 16);
    d.ref1 = y;
    d.ref2 = z;
    d.col = w;
    return d;
#else
    return ops[t];
#endif
}
constexpr int CONST_RANDO = 0, CONST_NEG_RANDO = 1, CONST_NEG_RANDO_146 = 2, NUM_CONST_PTS = 3;
constexpr int FB_WINDOWS = 66, MSM_DIGITS = 73;
constexpr int MSM_TABLE_OPS = 23;  // curve_msm_circuit's precomputation: 8 + 6 + 9 adds (gadgets/curve_msm.rs:45-60)
constexpr int COLS_ADD = 231, COLS_DBL = 282, COLS_CADD = 251;

// column anchors of the non-curve-op part of the schedule (filled by the host walk)
struct ScalarCols {
    int32_t assert_valid;  // 224 columns, -1 if absent (glv_mul-only program)
    int32_t inv_s;         // 18
    int32_t u1, u2;        // 51 each
    int32_t glv;           // 12 + 20 + 20 + 51 + 10 = 113
    int32_t beta_x;        // 51
    int32_t neg_p, neg_sp; // 20 each
};

struct Program {
    int32_t num_ops;        // curve ops
    int32_t num_slots;      // num_ops + 2 (p_neg, sp_neg)
    int32_t slot_p, slot_sp;
    int32_t num_cadd;
    int32_t full_verify;    // 1: verify circuit, 0: glv_mul only
    int32_t num_cols;
    ScalarCols sc;
    u32 msm_tab[16];        // refs of precomputation[0..15]
    // sequentially dependent op ranges ("chains").  verify: 0 = MSM chain, 1 = fixed-base chain (independent of
    // chain 0, own stream), 2 = final add (needs both).  glv_mul: 0 = the whole schedule.
    int32_t num_chains;
    int32_t chain_begin[4], chain_end[4];
    // the MSM loop: msm_loop_iters iterations of (double, double, conditional add) starting at op msm_loop_begin
    int32_t msm_loop_begin, msm_loop_iters;
    // curve programs (curves.hpp; zero for the two built-in programs): kind, digit rows of the per-signature table /
    // bit selects, the column and constant of the one curve_neg(constant) generator, ops of the per-signature table
    int32_t cp_kind, cp_rows, cp_neg_col, cp_neg_const, cp_table_ops;
    // the fixed-base windows: fb_windows conditional adds starting at op fb_begin (-1: none), and the doublings per
    // iteration of the loop at msm_loop_begin (2: curve_msm_circuit, 4: curve_scalar_mul_windowed)
    int32_t fb_begin, fb_windows, loop_dbls;
};

struct Buffers {
    // inputs, packed 32-byte little-endian, element i at +32*i
    const uint8_t *msg, *r, *s, *pkx, *pky;  // glv_mul-only: pkx, pky, and k in `msg`
    Sink sink;       // where the witness values go (u64 column matrix, or the compact container)
    size_t n;
    u32* err;        // per-element error bits (never null; OR-ed atomically by phases B and C)
    uint8_t* valid;  // per-element "all connect constraints hold" (never null)
    // scratch, [slot][n]: Jacobian points + numerator of v^-1 (phase A), prefix products and affine points (phase B)
    U256 *PX, *PY, *PZ, *PW, *PREF, *AX, *AY;
    uint8_t* dig4;   // [66][n]
    uint8_t* dig2;   // [73][n]   4*m_d + n_d
    uint16_t* msrc;  // [73][n]   resolved source id of precomputation[4*m_d + n_d] (slot, or constant | DYN_CONST_BIT)
    uint16_t* dyn;   // [num_cadd][n]
    uint16_t* src;   // [2 * num_ops][n]: resolved operand ids of every op (written by phase A for phase C)
    // constants
    const Aff* cpts;   // [NUM_CONST_PTS]
    const Aff* fbtab;  // [66][16]
    const OpDesc* ops;
};

P2E_HD void err_or(u32* p, u32 v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicOr(p, v);
#else
    *p |= v;
#endif
}
P2E_HD U256 load_packed(const uint8_t* base, size_t i) {
    const u32* p = reinterpret_cast<const u32*>(base + 32 * i);
    U256 r;
    P2E_UNROLL
    for (int k = 0; k < 8; k++) r.w[k] = p[k];
    return r;
}
// digit t (width WB) of a value = bits [WB*t, WB*t+WB): identical to the reference's per-limb LE bit
// split + regroup (gadgets/split_nonnative.rs:25-72) because limbs are 29 contiguous bits each
template <int WB>
P2E_HD u32 digit_of(const U256& v, int t) {
    int bit = WB * t;
    if (bit >= 256) return 0;
    return (v.w[bit >> 5] >> (bit & 31)) & ((1u << WB) - 1);  // WB divides 32: never straddles
}


// ---- phase S --------------------------------------------------------------------------------------------
template <class E>
P2E_HD void body_scalar(const Program& G, const Buffers& B, size_t i) {
    uint8_t err = 0;
    bool ok = true;
    U256 px = load_packed(B.pkx, i), py = load_packed(B.pky, i);
    U256 k;  // scalar multiplying pk
    if (G.full_verify) {
        U256 msg = load_packed(B.msg, i), r = load_packed(B.r, i), s = load_packed(B.s, i);
        {  // curve_assert_valid gadgets/curve.rs:123-135
            E e = E::at(B.sink, i, (u32)G.sc.assert_valid);
            U256 y2 = wit_mul<ModP>(e, py, py, err);
            U256 x2 = wit_mul<ModP>(e, px, px, err);
            U256 x3 = wit_mul<ModP>(e, x2, px, err);
            U256 ax = wit_mul<ModP>(e, u256_zero(), px, err);
            U256 axb = wit_add<ModP>(e, ax, u256_small(7));
            U256 rhs = wit_add<ModP>(e, x3, axb);
            ok = ok && u256_eq(y2, rhs);
            e.flush();
        }
        E e = E::at(B.sink, i, (u32)G.sc.inv_s);
        U256 c = wit_inv<ModN>(e, s, err);        // gadgets/ecdsa.rs:40
        U256 u1 = wit_mul<ModN>(e, msg, c, err);  // :41
        k = wit_mul<ModN>(e, r, c, err);          // :42
        e.flush();
        for (int w = 0; w < FB_WINDOWS; w++) B.dig4[(size_t)w * B.n + i] = (uint8_t)digit_of<4>(u1, w);
    } else {
        k = load_packed(B.msg, i);
    }
    // decompose_secp256k1_scalar gadgets/glv.rs:53-85
    GlvOut g = glv_decompose(k);
    {
        E e = E::at(B.sink, i, (u32)G.sc.glv);
        u32 l[NL];
        split29(g.k1, l);
        emit_limbs(e, l, 5);
        if (l[5] | l[6] | l[7] | l[8]) err |= ERR_LIMB_RANGE;
        split29(g.k2, l);
        emit_limbs(e, l, 5);
        if (l[5] | l[6] | l[7] | l[8]) err |= ERR_LIMB_RANGE;
        e.put(g.n1);
        e.put(g.n2);
        U256 k1r = wit_cond_neg<ModN>(e, g.k1, g.n1);
        U256 k2r = wit_cond_neg<ModN>(e, g.k2, g.n2);
        U256 S;
        {
            const u64 sv[4] = {16069571880186789234ull, 1310022930574435960ull, 11900229862571533402ull,
                               6008836872998760672ull};
            P2E_UNROLL
            for (int q = 0; q < 4; q++) {
                S.w[2 * q] = (u32)sv[q];
                S.w[2 * q + 1] = (u32)(sv[q] >> 32);
            }
        }
        U256 sb = wit_mul<ModN>(e, S, k2r, err);
        sb = wit_add<ModN>(e, sb, k1r);
        // connect_nonnative(should_be_k, k): limb-wise equality with the (raw) k target
        ok = ok && u256_eq(sb, k);
        e.flush();
    }
    for (int d = 0; d < MSM_DIGITS; d++) {
        const u32 idx = 4 * digit_of<2>(g.k2, d) + digit_of<2>(g.k1, d);
        B.dig2[(size_t)d * B.n + i] = (uint8_t)idx;
        // the table entry's source id, resolved here once: phases A would otherwise chain digit -> msm_tab -> point loads
        const u32 tr = G.msm_tab[idx];
        B.msrc[(size_t)d * B.n + i] = ref_kind(tr) == R_CONST ? (uint16_t)(ref_id(tr) | DYN_CONST_BIT) : (uint16_t)ref_id(tr);
    }
    // glv_mul gadgets/glv.rs:87-104
    U256 beta;
    {
        const u64 bv[4] = {13923278643952681454ull, 11308619431505398165ull, 7954561588662645993ull,
                           8856726876819556112ull};
        P2E_UNROLL
        for (int q = 0; q < 4; q++) {
            beta.w[2 * q] = (u32)bv[q];
            beta.w[2 * q + 1] = (u32)(bv[q] >> 32);
        }
    }
    E e1 = E::at(B.sink, i, (u32)G.sc.beta_x);
    U256 bx = wit_mul<ModP>(e1, beta, px, err);
    e1.flush();
    E e2 = E::at(B.sink, i, (u32)G.sc.neg_p);
    U256 y1 = wit_cond_neg<ModP>(e2, py, g.n1);
    e2.flush();
    E e3 = E::at(B.sink, i, (u32)G.sc.neg_sp);
    U256 y2 = wit_cond_neg<ModP>(e3, py, g.n2);
    e3.flush();
    size_t sp = (size_t)G.slot_p * B.n + i, ssp = (size_t)G.slot_sp * B.n + i;
    U256 pxc = fe_canon<ModP>(px);  // only ever consumed through canonicalising generators
    B.PX[sp] = pxc;
    B.PY[sp] = y1;
    B.PX[ssp] = bx;
    B.PY[ssp] = y2;
    B.AX[sp] = pxc;
    B.AY[sp] = y1;
    B.AX[ssp] = bx;
    B.AY[ssp] = y2;
    B.err[i] = err;
    B.valid[i] = ok ? 1 : 0;
}

// ---- operand resolution -----------------------------------------------------------------------------------
// static part of a reference -> per-lane 16-bit source id (slot, or constant | DYN_CONST_BIT)
P2E_HD uint16_t resolve_src(const Program& G, const Buffers& B, size_t i, u32 ref) {
    u32 k = ref_kind(ref), id = ref_id(ref);
    if (k == R_SLOT) return (uint16_t)id;
    if (k == R_CONST) return (uint16_t)(id | DYN_CONST_BIT);
    if (k == R_DYN) return B.dyn[(size_t)id * B.n + i];
    if (k == R_MSMTAB) return B.msrc[(size_t)id * B.n + i];
    return 0;  // R_FBTAB never goes through here
}
P2E_HD Aff load_aff_src(const Buffers& B, size_t i, uint16_t src) {
    Aff a;
    if (src & DYN_CONST_BIT) {
        a = B.cpts[src & SRC_ID_MASK];
    } else {
        a.x = B.AX[(size_t)src * B.n + i];
        a.y = B.AY[(size_t)src * B.n + i];
    }
    return a;
}
P2E_HD Jac load_jac_src(const Buffers& B, size_t i, uint16_t src, bool z_one) {
    Jac j;
    if (src & DYN_CONST_BIT) {
        Aff a = B.cpts[src & 0x7FFF];
        j.X = a.x;
        j.Y = a.y;
        j.Z = u256_small(1);
    } else {
        size_t o = (size_t)src * B.n + i;
        j.X = B.PX[o];
        j.Y = B.PY[o];
        j.Z = z_one ? u256_small(1) : B.PZ[o];
    }
    return j;
}
// lds_fb / lds_w0: optional LDS copy of the table rows of windows lds_w0 ... (the north star's "LDS-staged precomputed
// window tables"; measured against the L2-resident gather in DESIGN.md section 2 -- build with -DP2E_LDS_FBTAB)
P2E_HD Aff load_fbtab(const Buffers& B, size_t i, u32 window, u32& digit, const Aff* lds_fb = nullptr, u32 lds_w0 = 0) {
    digit = B.dig4[(size_t)window * B.n + i];
    if (lds_fb) return lds_fb[(window - lds_w0) * 16 + digit];
    return B.fbtab[window * 16 + digit];
}

// ---- phase A: one op of a chain in Jacobian coordinates -----------------------------------------------
// State a lane carries from op to op of a chain range: the previous op's result and first operand (the next
// op's first operand is one of the two in 310 of 311 ops: a doubling after a doubling, the selected result
// of a conditional add) and the running product of the Z3 values (the forward half of the Montgomery
// batch inversion, so that phase B only runs the backward half).
struct ChainState {
    Jac out, p1;
    uint16_t out_id, p1_id;
    U256 acc;
};
P2E_HD Jac jac_select3(bool c1, const Jac& a, bool c2, const Jac& b, const Jac& c) {
    Jac r;
    r.X = u256_select(c1, a.X, u256_select(c2, b.X, c.X));
    r.Y = u256_select(c1, a.Y, u256_select(c2, b.Y, c.Y));
    r.Z = u256_select(c1, a.Z, u256_select(c2, b.Z, c.Z));
    return r;
}
// table_affine: the MSM window table has already been through phase B (its 23-op piece is inverted before
// the loop pieces start), so table operands are read in affine form and the 73 window additions are
// mixed additions (11 multiplications instead of 17).
template <class CV = Secp256k1, bool GENERIC = false>
P2E_HD void body_chain_op(const Program& G, const Buffers& B, size_t i, int t, bool table_affine, ChainState& st,
                          const Aff* lds_fb = nullptr, u32 lds_w0 = 0) {
    typedef typename CV::Fp F;
    const OpDesc op = load_op(B.ops, t);
    size_t o = (size_t)t * B.n + i;
    JacW res;
    uint16_t src1 = resolve_src(G, B, i, op.ref1);
    const bool from_out = src1 == st.out_id, from_p1 = src1 == st.p1_id;
    Jac p1 = st.out;
    if (!(from_out || from_p1)) p1 = load_jac_src(B, i, src1, (op.flags & F_Z1ONE) != 0);
    p1 = jac_select3(from_out, st.out, from_p1, st.p1, p1);
    B.src[(size_t)(2 * t) * B.n + i] = src1;
    if (op.kind == OP_DBL) {
        res = jac_dbl_cv<CV>(p1);
    } else {
        Jac p2;
        u32 digit = 1;
        uint16_t src2;
        bool z2one = (op.flags & F_Z2ONE) != 0;
        if (GENERIC && ref_kind(op.ref2) == R_SELSLOT) {
            digit = B.dig2[(size_t)(ref_id(op.ref2) >> 12) * B.n + i];
            src2 = (uint16_t)(ref_id(op.ref2) & 0xFFFu);
            p2 = load_jac_src(B, i, src2, z2one);
        } else if (ref_kind(op.ref2) == R_FBTAB) {
            Aff a = load_fbtab(B, i, ref_id(op.ref2), digit, lds_fb, lds_w0);
            p2 = jac_from_aff(a);
            src2 = (uint16_t)(SRC_FB_BIT | (ref_id(op.ref2) * 16 + digit));
        } else {
            const bool tab = ref_kind(op.ref2) == R_MSMTAB;
            if (tab) digit = B.dig2[(size_t)ref_id(op.ref2) * B.n + i];
            src2 = resolve_src(G, B, i, op.ref2);
            if (tab && table_affine) {
                p2 = jac_from_aff(load_aff_src(B, i, src2));
                z2one = true;
            } else {
                p2 = load_jac_src(B, i, src2, (op.flags & F_Z2ONE) != 0);
            }
        }
        B.src[(size_t)(2 * t + 1) * B.n + i] = (uint16_t)(src2 | (digit != 0 ? SRC_SEL_BIT : 0));
        // the Z-one specialisations only skip multiplications by one: pick by the host-known flags
        if ((op.flags & F_Z1ONE) && z2one)
            res = jac_add_cv<CV, true, true>(p1, p2);
        else if (z2one)
            res = jac_add_cv<CV, false, true>(p1, p2);
        else if (op.flags & F_Z1ONE)
            res = jac_add_cv<CV, true, false>(p1, p2);
        else
            res = jac_add_cv<CV, false, false>(p1, p2);
        if (op.kind == OP_CADD) B.dyn[(size_t)op.cadd_idx * B.n + i] = digit != 0 ? (uint16_t)t : src1;
    }
    // X, Y of a result are read back only by phase B's affine conversion and by the first op of a later chain
    // piece; pieces are cut at run boundaries, whose two candidate results always keep their affine form
    if (!(op.flags & F_NO_AFFINE)) {
        B.PX[o] = res.p.X;
        B.PY[o] = res.p.Y;
    }
    B.PZ[o] = res.p.Z;
    B.PW[o] = res.W;
    // forward half of the batch inversion of this range
    B.PREF[o] = st.acc;
    U256 z = res.p.Z;
    if (u256_is_zero(z)) {  // reference: inverse() of zero panics (gadgets/nonnative.rs:863)
        err_or(&B.err[i], ERR_INVERSE_OF_ZERO);
        z = u256_small(1);
    }
    st.acc = fe_mul<F>(st.acc, z);
    st.p1 = p1;
    st.p1_id = (op.flags & F_Z1ONE) ? (uint16_t)0xFFFF : src1;   // an affine operand has no Z to carry
    st.out = res.p;
    st.out_id = (uint16_t)t;
}
// body_chain_op for secp256k1 on lazy limbs (ec29.hpp): same operand resolution, same scratch outputs (canonical), the
// state carried from op to op stays on lazy limbs.
struct ChainState29 {
    JacL out, p1;
    uint16_t out_id, p1_id;
    F29 acc;
};
template <bool GENERIC = false>
P2E_HD void body_chain_op29(const Program& G, const Buffers& B, size_t i, int t, bool table_affine, ChainState29& st,
                            const Aff* lds_fb = nullptr, u32 lds_w0 = 0) {
    const OpDesc op = load_op(B.ops, t);
    size_t o = (size_t)t * B.n + i;
    JacWL res;
    uint16_t src1 = resolve_src(G, B, i, op.ref1);
    const bool from_out = src1 == st.out_id, from_p1 = src1 == st.p1_id;
    JacL p1 = st.out;
    if (!(from_out || from_p1)) p1 = jacl_from(load_jac_src(B, i, src1, (op.flags & F_Z1ONE) != 0));
    p1 = jacl_select3(from_out, st.out, from_p1, st.p1, p1);
    B.src[(size_t)(2 * t) * B.n + i] = src1;
    if (op.kind == OP_DBL) {
        res = jac_dbl29(p1);
    } else {
        Jac p2w;
        u32 digit = 1;
        uint16_t src2;
        bool z2one = (op.flags & F_Z2ONE) != 0;
        if (GENERIC && ref_kind(op.ref2) == R_SELSLOT) {
            digit = B.dig2[(size_t)(ref_id(op.ref2) >> 12) * B.n + i];
            src2 = (uint16_t)(ref_id(op.ref2) & 0xFFFu);
            p2w = load_jac_src(B, i, src2, z2one);
        } else if (ref_kind(op.ref2) == R_FBTAB) {
            Aff a = load_fbtab(B, i, ref_id(op.ref2), digit, lds_fb, lds_w0);
            p2w = jac_from_aff(a);
            src2 = (uint16_t)(SRC_FB_BIT | (ref_id(op.ref2) * 16 + digit));
        } else {
            const bool tab = ref_kind(op.ref2) == R_MSMTAB;
            if (tab) digit = B.dig2[(size_t)ref_id(op.ref2) * B.n + i];
            src2 = resolve_src(G, B, i, op.ref2);
            if (tab && table_affine) {
                p2w = jac_from_aff(load_aff_src(B, i, src2));
                z2one = true;
            } else {
                p2w = load_jac_src(B, i, src2, (op.flags & F_Z2ONE) != 0);
            }
        }
        B.src[(size_t)(2 * t + 1) * B.n + i] = (uint16_t)(src2 | (digit != 0 ? SRC_SEL_BIT : 0));
        const JacL p2 = jacl_from(p2w);
        if ((op.flags & F_Z1ONE) && z2one)
            res = jac_add29<true, true>(p1, p2);
        else if (z2one)
            res = jac_add29<false, true>(p1, p2);
        else if (op.flags & F_Z1ONE)
            res = jac_add29<true, false>(p1, p2);
        else
            res = jac_add29<false, false>(p1, p2);
        if (op.kind == OP_CADD) B.dyn[(size_t)op.cadd_idx * B.n + i] = digit != 0 ? (uint16_t)t : src1;
    }
    if (!(op.flags & F_NO_AFFINE)) {
        B.PX[o] = f29_canon_call(res.p.X);
        B.PY[o] = f29_canon_call(res.p.Y);
    }
    const U256 zc = f29_canon_call(res.p.Z);
    B.PZ[o] = zc;
    B.PW[o] = f29_canon_call(res.W);
    B.PREF[o] = f29_canon_call(st.acc);
    const bool z_zero = u256_is_zero(zc);
    if (z_zero) err_or(&B.err[i], ERR_INVERSE_OF_ZERO);   // reference: inverse() of zero panics (gadgets/nonnative.rs:863)
    st.acc = f29_mul_call(st.acc, f29_select(z_zero, f29_small(1), res.p.Z));
    st.p1 = p1;
    st.p1_id = (op.flags & F_Z1ONE) ? (uint16_t)0xFFFF : src1;   // an affine operand has no Z to carry
    st.out = res.p;
    st.out_id = (uint16_t)t;
}
P2E_HD ChainState29 chain_state29_init(const U256& acc) {
    ChainState29 st;
    st.out_id = st.p1_id = 0xFFFF;
    st.out.X = st.out.Y = st.out.Z = st.p1.X = st.p1.Y = st.p1.Z = f29_small(0);
    st.acc = f29_from_u256(acc);
    return st;
}
// Product of the Z3 values (zeros replaced by one) of ops [lo, hi) given the prefix array: prefix of the
// last op times its own Z3.
template <class CV = Secp256k1>
P2E_HD U256 range_product(const Buffers& B, size_t i, int last) {
    size_t o = (size_t)last * B.n + i;
    U256 z = B.PZ[o];
    if (u256_is_zero(z)) z = u256_small(1);
    return fe_mul<typename CV::Fp>(B.PREF[o], z);
}
// ops [lo, hi) of one chain, in order (the range may be a piece of a chain: everything a later piece needs
// lives in scratch).  continue_prefix: the range extends the inversion batch of the ops just before it.
template <class CV = Secp256k1, bool GENERIC = false>
P2E_HD void body_chain_range(const Program& G, const Buffers& B, size_t i, int lo, int hi, bool table_affine,
                             bool continue_prefix, const Aff* lds_fb = nullptr, u32 lds_w0 = 0) {
    if (LazyLimbs<CV>::available) {
        ChainState29 st = chain_state29_init(continue_prefix ? range_product<CV>(B, i, lo - 1) : u256_small(1));
        for (int t = lo; t < hi; t++) body_chain_op29<GENERIC>(G, B, i, t, table_affine, st, lds_fb, lds_w0);
        return;
    }
    ChainState st;
    st.out_id = st.p1_id = 0xFFFF;
    st.out.X = st.out.Y = st.out.Z = st.p1.X = st.p1.Y = st.p1.Z = u256_zero();
    st.acc = continue_prefix ? range_product<CV>(B, i, lo - 1) : u256_small(1);
    for (int t = lo; t < hi; t++) body_chain_op<CV, GENERIC>(G, B, i, t, table_affine, st, lds_fb, lds_w0);
}

// Ops lo + row + j * rows (j < count) of a piece whose `rows` interleaved sub-chains do not depend on each other.
// The MSM window table (gadgets/curve_msm.rs:45-60, 23 ops) is such a piece three times over: the two 4-op chains
// rando + i*p / rando + i*q (rows = 2, count = 4), the 6 adds that strip rando and the 9 cross sums (one op per
// row).  Walking the rows with different lanes shortens the latency-bound start of every call from 23 dependent
// ops to 6.  The prefix products of the piece's inversion batch are then left to phase B (have_prefix = false).
P2E_HD void body_chain_rows(const Program& G, const Buffers& B, size_t i, int lo, int rows, int row, int count) {
    ChainState29 st = chain_state29_init(u256_small(1));
    for (int j = 0; j < count; j++) body_chain_op29(G, B, i, lo + row + j * rows, false, st);
}

// The connect r == x of gadgets/ecdsa.rs:48-52 on the final add's JACOBIAN result (p2e_ecdsa_verify_batch: verdict
// only, no batch inversion): x = X / Z^2 is canonical, so x == r  <=>  r < p and X == r * Z^2.
template <class CV = Secp256k1>
P2E_HD void body_verify_check(const Program& G, const Buffers& B, size_t i, int last_op) {
    typedef typename CV::Fp F;
    const size_t o = (size_t)last_op * B.n + i;
    const U256 r = load_packed(B.r, i), X = B.PX[o], Z = B.PZ[o];
    const bool r_lt_p = !geq_mod<F>(r.w);
    if (!(r_lt_p && !u256_is_zero(Z) && u256_eq(fe_mul<F>(r, fe_sqr<F>(Z)), X))) B.valid[i] = 0;
}
P2E_HD void body_verify_check(const Program& G, const Buffers& B, size_t i) {
    body_verify_check<Secp256k1>(G, B, i, G.chain_end[2] - 1);
}

// body_batch_inv (below) for secp256k1 on lazy limbs (fe29.hpp): the same
// products in the same order -- v^-1 = W * (inv * prefix), inv *= Z, and for the ops that keep an affine form
// X * zi^2, Y * zi^3 -- with three differences that only change the time: the multiplications are fe29's, a value is
// made canonical only where it is stored, and the inputs of op t - 1 are requested before op t is computed (the loop
// as written below exposes one memory round trip per op to a wave that has nothing else to do).
// The loads are unconditional (an op without an affine form reads its Z twice more instead of X and Y: a load inside a
// divergent branch drags its wait to the join).
P2E_HD void body_batch_inv29(const Buffers& B, size_t i, int t0, int t1, bool have_prefix, bool uniform_t) {
    U256 accw;
    if (have_prefix) {
        accw = range_product<Secp256k1>(B, i, t1 - 1);
    } else {
        F29 acc = f29_small(1);
        for (int t = t0; t < t1; t++) {
            const size_t o = (size_t)t * B.n + i;
            U256 z = B.PZ[o];
            if (u256_is_zero(z)) z = u256_small(1);   // flagged by phase A
            B.PREF[o] = f29_canon_call(acc);
            acc = f29_mul_call(acc, f29_from_u256(z));
        }
        accw = f29_canon_call(acc);
    }
    F29 inv = f29_from_u256(fe_inv<ModP>(accw));
    struct In {
        U256 z, pref, w, x, y;
        uint8_t flags;
    };
    auto flags_of = [&](int t) { return uniform_t ? load_op(B.ops, t < t0 ? t0 : t).flags : B.ops[t < t0 ? t0 : t].flags; };
    auto fetch = [&](int t, uint8_t fl) {
        In r;
        const size_t o = (size_t)(t < t0 ? t0 : t) * B.n + i;
        const bool aff = !(fl & F_NO_AFFINE);
        r.z = B.PZ[o];
        r.pref = B.PREF[o];
        r.w = B.PW[o];
        r.x = *(aff ? &B.PX[o] : &B.PZ[o]);
        r.y = *(aff ? &B.PY[o] : &B.PZ[o]);
        r.flags = fl;
        return r;
    };
    uint8_t f_nxt = flags_of(t1 - 2);
    In cur = fetch(t1 - 1, flags_of(t1 - 1));
    for (int t = t1 - 1; t >= t0; t--) {
        const uint8_t f_n2 = flags_of(t - 2);
        const In nxt = fetch(t - 1, f_nxt);
        const size_t o = (size_t)t * B.n + i;
        const U256 z = u256_select(u256_is_zero(cur.z), u256_small(1), cur.z);
        const F29 zi = f29_mul_call(inv, f29_from_u256(cur.pref));
        inv = f29_mul_call(inv, f29_from_u256(z));
        B.PW[o] = f29_canon_call(f29_mul_call(f29_from_u256(cur.w), zi));   // v^-1 of op t
        if (!(cur.flags & F_NO_AFFINE)) {
            const F29 zi2 = f29_sqr_call(zi);
            const F29 zi3 = f29_mul_call(zi2, zi);
            B.AX[o] = f29_canon_call(f29_mul_call(f29_from_u256(cur.x), zi2));
            B.AY[o] = f29_canon_call(f29_mul_call(f29_from_u256(cur.y), zi3));
        }
        cur = nxt;
        f_nxt = f_n2;
    }
}
// ---- phase B: Montgomery batch inversion of Z over ops [t0, t1) of one signature ------------------------
// Reads the Jacobian results (left intact: later pieces of the chain still consume them), writes the
// affine points to AX/AY and v^-1 over W.  have_prefix: [t0, t1) is exactly one inversion batch of phase A,
// whose prefix products are already in PREF (the normal case); otherwise the forward pass runs here.
// uniform_t: every lane of the wave walks the same ops (so the descriptor can be a scalar load); false when lanes own
// different sub-ranges (body_batch_inv_split)
template <class CV = Secp256k1>
P2E_HD void body_batch_inv(const Program& G, const Buffers& B, size_t i, int t0, int t1, bool have_prefix, bool uniform_t = true) {
    (void)G;
    typedef typename CV::Fp F;
    if (LazyLimbs<CV>::available) {
        body_batch_inv29(B, i, t0, t1, have_prefix, uniform_t);
        return;
    }
    U256 acc;
    if (have_prefix) {
        acc = range_product<CV>(B, i, t1 - 1);
    } else {
        acc = u256_small(1);
        for (int t = t0; t < t1; t++) {
            size_t o = (size_t)t * B.n + i;
            U256 z = B.PZ[o];
            if (u256_is_zero(z)) z = u256_small(1);   // flagged by phase A
            B.PREF[o] = acc;
            acc = fe_mul<F>(acc, z);
        }
    }
    U256 inv = fe_inv<F>(acc);
    for (int t = t1 - 1; t >= t0; t--) {
        size_t o = (size_t)t * B.n + i;
        U256 z = B.PZ[o];
        if (u256_is_zero(z)) z = u256_small(1);
        U256 zi = fe_mul<F>(inv, B.PREF[o]);
        inv = fe_mul<F>(inv, z);
        B.PW[o] = fe_mul<F>(B.PW[o], zi);  // v^-1 of op t
        const uint8_t flags = uniform_t ? load_op(B.ops, t).flags : B.ops[t].flags;
        if (!(flags & F_NO_AFFINE)) {
            U256 zi2 = fe_sqr<F>(zi);
            U256 zi3 = fe_mul<F>(zi2, zi);
            B.AX[o] = fe_mul<F>(B.PX[o], zi2);
            B.AY[o] = fe_mul<F>(B.PY[o], zi3);
        }
    }
}

// ---- phase C: witness columns of one curve op ---------------------------------------------------------------
// gadgets/curve.rs:202-223
template <class E, class CV = Secp256k1>
P2E_HD Aff wit_curve_add(E& e, const Aff& p1, const Aff& p2, const U256& vinv, uint8_t& err) {
    U256 u = wit_sub<typename CV::Fp>(e, p2.y, p1.y);
    U256 v = wit_sub<typename CV::Fp>(e, p2.x, p1.x);
    wit_inv_given<typename CV::Fp>(e, v, vinv, err);
    U256 s = wit_mul<typename CV::Fp>(e, u, vinv, err);
    U256 s2 = wit_mul<typename CV::Fp>(e, s, s, err);
    U256 xs = wit_add<typename CV::Fp>(e, p2.x, p1.x);
    Aff r;
    r.x = wit_sub<typename CV::Fp>(e, s2, xs);
    U256 xd = wit_sub<typename CV::Fp>(e, p1.x, r.x);
    U256 pr = wit_mul<typename CV::Fp>(e, s, xd, err);
    r.y = wit_sub<typename CV::Fp>(e, pr, p1.y);
    return r;
}
// gadgets/curve.rs:160-185
template <class E, class CV = Secp256k1>
P2E_HD Aff wit_curve_double(E& e, const Aff& p, const U256& vinv, uint8_t& err) {
    U256 dy = wit_add<typename CV::Fp>(e, p.y, p.y);
    wit_inv_given<typename CV::Fp>(e, dy, vinv, err);
    U256 xx = wit_mul<typename CV::Fp>(e, p.x, p.x, err);
    U256 summ[4] = {xx, xx, xx, CV::a()};
    U256 t = wit_add_many<typename CV::Fp, 4>(e, summ);
    U256 l = wit_mul<typename CV::Fp>(e, t, vinv, err);
    U256 l2 = wit_mul<typename CV::Fp>(e, l, l, err);
    U256 xd2 = wit_add<typename CV::Fp>(e, p.x, p.x);
    Aff r;
    r.x = wit_sub<typename CV::Fp>(e, l2, xd2);
    U256 xdf = wit_sub<typename CV::Fp>(e, p.x, r.x);
    U256 lx = wit_mul<typename CV::Fp>(e, l, xdf, err);
    r.y = wit_sub<typename CV::Fp>(e, lx, p.y);
    return r;
}
template <class E, class CV = Secp256k1>
P2E_HD void body_expand(const Program& G, const Buffers& B, size_t i, int t, const Aff* lds_fb = nullptr) {
    const OpDesc op = load_op(B.ops, t);
    uint8_t err = 0;
    E e = E::at(B.sink, i, op.col);
    // operands were resolved by phase A: one level of index loads, then the points
    const uint16_t s1 = B.src[(size_t)(2 * t) * B.n + i];
    const uint16_t s2 = op.kind == OP_DBL ? (uint16_t)0 : B.src[(size_t)(2 * t + 1) * B.n + i];
    U256 vinv = B.PW[(size_t)t * B.n + i];
    Aff p1 = load_aff_src(B, i, s1);
    if (op.kind == OP_DBL) {
        (void)wit_curve_double<E, CV>(e, p1, vinv, err);
    } else {
        Aff p2 = (s2 & SRC_FB_BIT) ? (lds_fb ? lds_fb[s2 & 15u] : B.fbtab[s2 & SRC_ID_MASK]) : load_aff_src(B, i, (uint16_t)(s2 & (DYN_CONST_BIT | SRC_ID_MASK)));
        const bool sel = (s2 & SRC_SEL_BIT) != 0;
        Aff s = wit_curve_add<E, CV>(e, p1, p2, vinv, err);
        if (op.kind == OP_CADD) {  // gadgets/curve.rs:225-243: sum always computed (Q7), then selected
            bool b = sel;
            const U256 z = u256_zero();
            (void)wit_add<typename CV::Fp>(e, u256_select(b, s.x, z), u256_select(b, z, p1.x));
            (void)wit_add<typename CV::Fp>(e, u256_select(b, s.y, z), u256_select(b, z, p1.y));
        }
        if (op.flags & F_CHECK_R) {  // gadgets/ecdsa.rs:48-52 connect_nonnative(r, point.x)
            U256 r = load_packed(B.r, i);
            if (!u256_eq(s.x, r)) B.valid[i] = 0;
        }
    }
    e.flush();
    if (err) err_or(&B.err[i], err);
}

// Phase C over a RUN of MSM-loop iterations [it0, it1): the lane walks double, double, conditional add ...
// with the running point in registers (the affine results are by-products of the witness arithmetic), so per
// op it only reads v^-1 (32 B) and, per iteration, one table point.  Phase B therefore converts only the two
// candidates for the running point at the END of each run (F_NO_AFFINE on the rest).
// DBLS: doublings per iteration -- 2 in curve_msm_circuit (gadgets/curve_msm.rs:66-67), 4 in curve_scalar_mul_windowed
// (gadgets/curve_windowed_mul.rs:157)
template <class E, class CV = Secp256k1, int DBLS = 2>
P2E_HD void body_expand_run(const Program& G, const Buffers& B, size_t i, int it0, int it1) {
    uint8_t err = 0;
    int t = G.msm_loop_begin + (DBLS + 1) * it0;
    Aff p = load_aff_src(B, i, B.src[(size_t)(2 * t) * B.n + i]);
    for (int it = it0; it < it1; it++, t += DBLS + 1) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
        for (int k = 0; k < DBLS; k++) {  // curve_repeated_double(result, DBLS)
            E e = E::at(B.sink, i, load_op(B.ops, t + k).col);
            p = wit_curve_double<E, CV>(e, p, B.PW[(size_t)(t + k) * B.n + i], err);
            e.flush();
        }
        const int tc = t + DBLS;
        const uint16_t s2 = B.src[(size_t)(2 * tc + 1) * B.n + i];
        Aff p2 = load_aff_src(B, i, (uint16_t)(s2 & (DYN_CONST_BIT | SRC_ID_MASK)));
        const bool b = (s2 & SRC_SEL_BIT) != 0;
        E e = E::at(B.sink, i, load_op(B.ops, tc).col);
        Aff sm = wit_curve_add<E, CV>(e, p, p2, B.PW[(size_t)tc * B.n + i], err);
        const U256 z = u256_zero();
        Aff nx;
        nx.x = wit_add<typename CV::Fp>(e, u256_select(b, sm.x, z), u256_select(b, z, p.x));
        nx.y = wit_add<typename CV::Fp>(e, u256_select(b, sm.y, z), u256_select(b, z, p.y));
        e.flush();
        p = nx;
    }
    if (err) err_or(&B.err[i], err);
}

// Phase C over ALL fixed-base windows of one signature (gadgets/curve_fixed_base.rs:43-62): the lane walks the
// conditional adds against the constant table with the running point in registers, so per op it reads only v^-1 and the
// resolved table index; phase B then converts none of these results (F_NO_AFFINE) and phase A stores no X, Y for them.
// The running point after the last window is the first operand of op t_after (the unblinding add): it is left in the
// affine slot phase A resolved for that operand.
template <class E, class CV = Secp256k1>
P2E_HD void body_expand_fb_run(const Program& G, const Buffers& B, size_t i, int t_after) {
    uint8_t err = 0;
    int t = G.fb_begin;
    Aff p = load_aff_src(B, i, B.src[(size_t)(2 * t) * B.n + i]);
    for (int w = 0; w < G.fb_windows; w++, t++) {
        const uint16_t s2 = B.src[(size_t)(2 * t + 1) * B.n + i];
        const Aff p2 = B.fbtab[s2 & SRC_ID_MASK];
        const bool b = (s2 & SRC_SEL_BIT) != 0;
        E e = E::at(B.sink, i, load_op(B.ops, t).col);
        Aff sm = wit_curve_add<E, CV>(e, p, p2, B.PW[(size_t)t * B.n + i], err);
        const U256 z = u256_zero();
        Aff nx;
        nx.x = wit_add<typename CV::Fp>(e, u256_select(b, sm.x, z), u256_select(b, z, p.x));
        nx.y = wit_add<typename CV::Fp>(e, u256_select(b, sm.y, z), u256_select(b, z, p.y));
        e.flush();
        p = nx;
    }
    const uint16_t dst = B.src[(size_t)(2 * t_after) * B.n + i];
    if (!(dst & DYN_CONST_BIT)) {   // (every digit zero: the running point is still the constant it started from)
        B.AX[(size_t)dst * B.n + i] = p.x;
        B.AY[(size_t)dst * B.n + i] = p.y;
    }
    if (err) err_or(&B.err[i], err);
}

}  // namespace p2e
