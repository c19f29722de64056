// Host-side mirror of the reference's gadget API for the hot path (HOST ONLY, plain C++).
//
// The reference builds its circuit by calling CircuitBuilderNonNative / CircuitBuilderCurve /
// CircuitBuilderGlv methods; every call registers witness generators in a fixed order.  This builder
// offers the same method names with the same argument meaning, but instead of plonky2 gates it records
//   (1) the generator list  -> column map (one entry per reference generator, registration order),
//   (2) the curve-op table  -> OpDesc[] executed by the GPU pipeline (pipeline.hpp).
// Reference interface mirrored (paths relative to /root/reference/src):
//   gadgets/nonnative.rs:53-164   add/sub/mul/inv/add_many/neg/nonnative_conditional_neg
//   gadgets/curve.rs:34-94        curve_assert_valid/curve_add/curve_double/curve_repeated_double/
//                                 curve_conditional_add/curve_conditional_neg
//   gadgets/curve_windowed_mul.rs:74-118  random_access_curve_points
//   gadgets/curve_fixed_base.rs:18 fixed_base_curve_mul_circuit
//   gadgets/curve_msm.rs:21       curve_msm_circuit
//   gadgets/glv.rs:26-44          decompose_secp256k1_scalar / glv_mul
//   gadgets/ecdsa.rs:30           verify_secp256k1_message_circuit
// and, for the curve programs of curves.hpp:
//   gadgets/curve.rs:137-147,245-285          curve_neg / curve_scalar_mul
//   gadgets/curve_windowed_mul.rs:52-72,131-173  precompute_window / curve_scalar_mul_windowed
//   gadgets/ecdsa.rs:55-78                    verify_p256_message_circuit
#pragma once
#include <cassert>
#include <string>
#include <vector>

#include "aux.hpp"
#include "consts.hpp"
#include "ux.hpp"
#include "pipeline.hpp"
#include "curves.hpp"

namespace p2e {
namespace host {

enum GenKind : int { GEN_ADD = 0, GEN_SUB = 1, GEN_ADD_MANY = 2, GEN_MUL = 3, GEN_INV = 4, GEN_GLV = 5 };
enum Field : int { FIELD_BASE = 0, FIELD_SCALAR = 1 };

// One hot-path generator: its output columns AND where its operands live (the wiring the reference's gadgets fix when
// they pass targets from one call to the next).  src[k] is a source code of aux.hpp (a witness column, an aux
// column, a caller input, a constant), nl[k] the limbs that operand target really has.  range_check: the gadget's
// range_check flag (an extra cmp_biguint in the constraint block, gadgets/nonnative.rs:180-190).
struct GenOp {
    int kind, field;
    u32 col, ncols;
    std::string label;
    int nops = 0;
    u32 src[4] = {AUX_SRC_NONE, AUX_SRC_NONE, AUX_SRC_NONE, AUX_SRC_NONE};
    uint8_t nl[4] = {0, 0, 0, 0};
    bool range_check = false;
};

// symbolic NonNativeTarget: the hot path never needs its value on the host, only its identity and, for the
// built-in-generator columns (aux.hpp), where its limbs live: an output column, a constant, or a caller input
struct NonNativeTarget {
    int field;
    u32 col = AUX_SRC_NONE;  // first column of its limbs in the witness matrix, or an AUX_SRC_* code
    int nl = NL;             // limbs the target really has (constants and k1, k2 have fewer than 9)
};
// symbolic BoolTarget: the witness column holding it
struct BoolTarget {
    u32 col;
};
// symbolic AffinePointTarget: where the GPU finds the point
struct AffinePointTarget {
    u32 ref;
    bool z_one;  // known to be stored in affine form before phase B (constants, inputs)
    u32 xcol = AUX_SRC_NONE, ycol = AUX_SRC_NONE;  // limb columns of x and y (or AUX_SRC_* codes)
    int nlx = NL, nly = NL;
};
struct AuxGen {   // one entry of the built-in-generator column map (p2e_aux_describe)
    int kind;
    u32 col, ncols;
    std::string label;
};

class ScheduleBuilder {
public:
    std::vector<GenOp> gens;
    std::vector<OpDesc> ops;
    Program prog{};
    ScheduleBuilder() { prog.fb_begin = -1; }
    // constraint-block columns (ux.hpp): one item per generator with a non-empty block, and per
    // generator (same index as gens) its first ux column / column count
    std::vector<UxItem> ux_items;
    std::vector<u32> ux_first, ux_count;
    u32 num_ux_cols = 0;
    // built-in-generator columns: items for k_aux + their column map
    std::vector<AuxItem> aux_items;
    std::vector<GateItem> gate_items;   // gate-internal values, one block per window (aux.hpp body_gate)
    u32 num_gate_cols = 0;
    std::vector<AuxGen> aux_gens;
    AuxTables aux_tab{};

    // ---- CircuitBuilderNonNative ----  (every result target = the first 9 columns its generator writes)
    NonNativeTarget add_nonnative(NonNativeTarget a, NonNativeTarget b, bool range_check = false) {
        return {a.field, gen(GEN_ADD, a.field, 10, {a, b}, range_check), NL};
    }
    NonNativeTarget sub_nonnative(NonNativeTarget a, NonNativeTarget b, bool range_check = false) {
        return {a.field, gen(GEN_SUB, a.field, 10, {a, b}, range_check), NL};
    }
    NonNativeTarget add_many_nonnative(const std::vector<NonNativeTarget>& xs, bool range_check = false) {
        if (xs.size() == 1) return xs[0];
        return {xs[0].field, gen(GEN_ADD_MANY, xs[0].field, 10, xs, range_check), NL};
    }
    NonNativeTarget mul_nonnative(NonNativeTarget a, NonNativeTarget b, bool range_check = false) {
        // MulNonnativeGate r,q,check_sum (35) + CheckSumGate b (16); the product target = the r wires
        return {a.field, gen(GEN_MUL, a.field, 51, {a, b}, range_check), NL};
    }
    NonNativeTarget inv_nonnative(NonNativeTarget a, bool range_check = false) {
        return {a.field, gen(GEN_INV, a.field, 18, {a}, range_check), NL};
    }
    NonNativeTarget constant_nonnative(int field, u32 const_id) {   // constant_biguint: convert_base's limb count (Q5)
        return {field, AUX_SRC_CONST | const_id, const_num_limbs(cval(const_id))};
    }
    NonNativeTarget neg_nonnative(NonNativeTarget a, bool range_check = false) {   // gadgets/nonnative.rs:491-500
        return sub_nonnative(constant_nonnative(a.field, id_zero_), a, range_check);
    }
    // gadgets/nonnative.rs:584-596: not(b), neg, neg * b, x * not_b, add
    NonNativeTarget nonnative_conditional_neg(NonNativeTarget x, BoolTarget b, bool range_check = false) {
        AuxItem it{};
        it.kind = AUX_CNEG;
        it.a = b.col;
        const u32 base = aux_col_;   // [not_b, neg * b (9), x * not_b (x.nl)]
        NonNativeTarget neg = neg_nonnative(x);
        it.sumx = neg.col;
        it.p1x = x.col;
        it.nlx = (uint8_t)x.nl;
        aux(it, 1 + NL + (u32)x.nl);
        NonNativeTarget t{x.field, AUX_SRC_AUX | (base + 1), NL}, f{x.field, AUX_SRC_AUX | (base + 1 + NL), x.nl};
        return add_nonnative(t, f, range_check);
    }
    // gadgets/split_nonnative.rs:25-50 / :52-72: split_le_base bits of every limb, then the digits built from them
    void split_nonnative_to_4_bit_limbs(NonNativeTarget v) {
        split4_col_ = aux_col_;
        AuxItem it{};
        it.kind = AUX_SPLIT4;   // aux.hpp splits exactly the 9 limbs of a full scalar here ...
        assert(v.nl == NL);
        it.a = v.col;
        it.nlx = (uint8_t)v.nl;
        const u32 nbits = (u32)v.nl * BITS;
        aux(it, nbits + 3 * ((nbits + 3) / 4));
    }
    void split_nonnative_to_2_bit_limbs(NonNativeTarget v) {
        AuxItem it{};
        it.kind = AUX_SPLIT2;   // ... and the 5 limbs of a GLV half-scalar here
        assert(v.nl == 5);
        it.a = v.col;
        it.nlx = (uint8_t)v.nl;
        const u32 nbits = (u32)v.nl * BITS;
        aux(it, nbits + (nbits + 1) / 2);
    }

    // ---- CircuitBuilderCurve ----
    void curve_assert_valid(const AffinePointTarget& p) {   // gadgets/curve.rs:123-135
        Scope s(this, "assert_valid");
        const NonNativeTarget x = x_of(p), y = y_of(p);
        const NonNativeTarget a = constant_nonnative(FIELD_BASE, id_a_), b = constant_nonnative(FIELD_BASE, id_b_);
        mul_nonnative(y, y, true);
        NonNativeTarget x2 = mul_nonnative(x, x);
        NonNativeTarget x3 = mul_nonnative(x2, x);
        NonNativeTarget ax = mul_nonnative(a, x);
        NonNativeTarget axb = add_nonnative(ax, b);
        add_nonnative(x3, axb, true);
    }
    AffinePointTarget curve_add(AffinePointTarget p1, AffinePointTarget p2, bool range_check = false) {
        return curve_op(OP_ADD, p1, p2, range_check);
    }
    AffinePointTarget curve_double(AffinePointTarget p, bool range_check = false) { return curve_op(OP_DBL, p, p, range_check); }
    AffinePointTarget curve_repeated_double(AffinePointTarget p, int n, bool range_check = false) {   // gadgets/curve.rs:187-200
        for (int i = 0; i < n - 1; i++) p = curve_double(p, false);
        return curve_double(p, range_check);
    }
    // b is always "random-access index != 0" on the supported circuits; it is derived from p2's ref.
    // not_b_aux: aux column of not(b) -- the four mul_biguint_by_bool products follow it (gadgets/curve.rs:232-238)
    AffinePointTarget curve_conditional_add(AffinePointTarget p1, AffinePointTarget p2, u32 not_b_aux, bool range_check = false) {
        return curve_op(OP_CADD, p1, p2, range_check, not_b_aux);
    }
    AffinePointTarget constant_affine_point(int which) {
        return {make_ref(R_CONST, (u32)which), true, AUX_SRC_CONST | (u32)(2 * which), AUX_SRC_CONST | (u32)(2 * which + 1),
                const_num_limbs(cval((u32)(2 * which))), const_num_limbs(cval((u32)(2 * which + 1)))};
    }

    // ---- fixed_base_curve_mul_circuit(builder, G, scalar) ----
    AffinePointTarget fixed_base_curve_mul_circuit(NonNativeTarget scalar) {
        split_nonnative_to_4_bit_limbs(scalar);
        AffinePointTarget result = constant_affine_point(id_rando_);
        prog.fb_begin = (int32_t)ops.size();
        prog.fb_windows = FB_WINDOWS;
        for (int w = 0; w < FB_WINDOWS; w++) {
            Scope s(this, "win" + std::to_string(w));
            // is_equal(limb, zero), not, random_access_curve_points(limb, muls_point), curve_conditional_add
            // aux columns of this window: [is_zero, should_add, selected x (9), selected y (9), not_b, products ...]
            const u32 base = aux_col_;
            AffinePointTarget r{make_ref(R_FBTAB, (u32)w), true, AUX_SRC_AUX | (base + 2), AUX_SRC_AUX | (base + 2 + NL)};
            AuxItem it = select_item(AUX_FBWIN, result);
            it.a = scalar.col;
            it.b = (u32)w;
            // the window's digit = the third value of its (lower, upper, limb) triple behind the 261 split bits
            gate_items.push_back({split4_col_ + (u32)NL * BITS + 3 * (u32)w + 2, num_gate_cols, 0});
            num_gate_cols += GATE_COLS_PER_WINDOW;
            result = curve_conditional_add(result, r, base + 2 + 2 * NL);
            it.sumx = ops.back().col + COL_ADD_X3;
            it.sumy = ops.back().col + COL_ADD_Y3;
            aux(it, 2 + 2 * NL + 1 + 2 * NL + (u32)it.nlx + (u32)it.nly);
        }
        Scope s(this, "unblind");
        return curve_add(result, constant_affine_point(id_neg_rando_), true);
    }

    // ---- curve programs (curves.hpp) -----------------------------------------------------------------------
    // Constants of a curve program live in the builder: points gpts[c] (source codes 2c / 2c + 1 as for the built-in
    // ones), scalar constants gvals[j] (source code AUX_GCONST_BASE + j), the fixed-base table gfbtab of the curve's generator.
    std::vector<Aff> gpts, gfbtab;
    std::vector<U256> gvals;
    void begin_curve_program(int kind, int curve, const U256& a, const U256& b) {
        generic_ = true;
        curve_ = curve;
        prog.cp_kind = kind;
        id_zero_ = add_const_value(u256_zero());
        id_a_ = add_const_value(a);
        id_b_ = add_const_value(b);
    }
    u32 add_const_point(const Aff& a) {
        gpts.push_back(a);
        assert(2 * gpts.size() <= AUX_GCONST_BASE);
        return (u32)gpts.size() - 1;
    }
    u32 add_const_value(const U256& v) {
        gvals.push_back(v);
        return AUX_GCONST_BASE + (u32)gvals.size() - 1;
    }
    // the value behind AUX_SRC_CONST | id in THIS builder's program
    U256 cval(u32 id) const {
        if (!generic_) return const_value(id);
        if (id >= AUX_GCONST_BASE) return gvals[id - AUX_GCONST_BASE];
        return (id & 1) ? gpts[id >> 1].y : gpts[id >> 1].x;
    }
    // curve_neg of a CONSTANT point (gadgets/curve.rs:137-147): neg_nonnative(y) is a subtraction generator (10 columns,
    // filled by the scalar phase); the GPU reads the negated point itself from constant `which_negated`
    AffinePointTarget curve_neg_const(u32 which, u32 which_negated) {
        AffinePointTarget p = constant_affine_point((int)which);
        prog.cp_neg_col = (int32_t)col_;
        prog.cp_neg_const = (int32_t)which;
        NonNativeTarget ny = neg_nonnative(y_of(p));
        AffinePointTarget q = constant_affine_point((int)which_negated);
        q.xcol = p.xcol;
        q.nlx = p.nlx;
        q.ycol = ny.col;
        q.nly = NL;
        return q;
    }
    // gadgets/curve_windowed_mul.rs:52-72; g / neg_g: the rand() point and its negative (program constants)
    void precompute_window(const AffinePointTarget& p, u32 g, u32 neg_g, AffinePointTarget* multiples /*16*/) {
        AffinePointTarget neg = constant_affine_point((int)neg_g);
        multiples[0] = constant_affine_point((int)g);
        for (int i = 1; i < 16; i++) multiples[i] = curve_add(p, multiples[i - 1], true);
        for (int i = 1; i < 16; i++) multiples[i] = curve_add(neg, multiples[i], true);
    }
    // gadgets/curve_windowed_mul.rs:131-173.  Constants: g, -g, starting_point, 2^264 starting_point and its negative
    AffinePointTarget curve_scalar_mul_windowed(const AffinePointTarget& p, NonNativeTarget n, u32 g, u32 neg_g, u32 start, u32 spm,
                                                u32 neg_spm, bool range_check) {
        split_nonnative_to_4_bit_limbs(n);
        AffinePointTarget result = constant_affine_point((int)start);
        AffinePointTarget pre[16];
        {
            Scope s(this, "precompute");
            const int t0 = (int)ops.size();
            precompute_window(p, g, neg_g, pre);
            prog.cp_table_ops = (int32_t)ops.size() - t0;
        }
        for (int i = 0; i < 16; i++) {
            prog.msm_tab[i] = pre[i].ref;
            aux_tab.tabx[i] = pre[i].xcol;
            aux_tab.taby[i] = pre[i].ycol;
        }
        prog.cp_rows = CP_WINDOWS;
        loop_begin_ = (int32_t)ops.size();
        loop_iters_ = CP_WINDOWS;
        prog.loop_dbls = 4;
        for (int w = CP_WINDOWS - 1; w >= 0; w--) {
            Scope s(this, "window" + std::to_string(w));
            result = curve_repeated_double(result, 4, false);
            // aux columns of this window: [selected x (9), selected y (9), is_zero, should_add, not_b, sum.x*b (9),
            // sum.y*b (9), p1.x*not_b, p1.y*not_b]  (random_access first, then is_equal / not: :160-163)
            const u32 base = aux_col_;
            AffinePointTarget r{make_ref(R_MSMTAB, (u32)w), false, AUX_SRC_AUX | base, AUX_SRC_AUX | (base + NL)};
            AuxItem it = select_item(AUX_CP_WINDOW, result);
            it.a = n.col;
            it.b = (u32)w;
            // the window's index = the third value of its (lower, upper, limb) triple behind the split bits; the random
            // access comes before is_equal here (:159-161)
            gate_items.push_back({split4_col_ + (u32)NL * BITS + 3 * (u32)w + 2, num_gate_cols, 1});
            num_gate_cols += GATE_COLS_PER_WINDOW;
            result = curve_conditional_add(result, r, base + 2 * NL + 2, false);
            it.sumx = ops.back().col + COL_ADD_X3;
            it.sumy = ops.back().col + COL_ADD_Y3;
            aux(it, 2 * NL + 2 + 1 + 2 * NL + (u32)it.nlx + (u32)it.nly);
        }
        Scope s(this, "unblind");
        AffinePointTarget to_add = curve_neg_const(spm, neg_spm);
        return curve_add(result, to_add, range_check);
    }
    // gadgets/curve.rs:245-285.  Constants: rando (the rand() point) and its negative
    AffinePointTarget curve_scalar_mul(const AffinePointTarget& p, NonNativeTarget n, u32 rando, u32 neg_rando, bool range_check) {
        assert(n.nl == NL);
        {   // split_nonnative_to_bits gadgets/nonnative.rs:566-582
            AuxItem it{};
            it.kind = AUX_CP_BITS;
            it.a = n.col;
            it.nlx = (uint8_t)n.nl;
            aux(it, (u32)n.nl * BITS);
        }
        // result: a virtual point connected to the constant (9-limb targets that carry the constant's limbs)
        AffinePointTarget result = constant_affine_point((int)rando);
        result.nlx = result.nly = NL;
        AffinePointTarget tp = p;
        prog.cp_rows = CP_BITS;
        for (int i = 0; i < CP_BITS; i++) {
            Scope s(this, "bit" + std::to_string(i));
            // aux columns of this bit: [not_bit, sum.x*bit (9), result.x*not_bit (9), sum.y*bit (9), result.y*not_bit (9)]
            const u32 base = aux_col_;
            AffinePointTarget sel = tp;
            assert(ref_kind(tp.ref) == R_SLOT);
            const u32 slot = ref_id(tp.ref) == SLOT_P_PLACEHOLDER ? 0xFFFu : ref_id(tp.ref);
            assert(slot <= 0xFFFu);
            sel.ref = make_ref(R_SELSLOT, slot | ((u32)i << 12));
            AuxItem it = select_item(AUX_CP_BIT, result);
            it.a = n.col;
            it.b = (u32)i;
            result = curve_op(OP_CADD, result, sel, false, base, 1);
            it.sumx = ops.back().col + COL_ADD_X3;
            it.sumy = ops.back().col + COL_ADD_Y3;
            aux(it, 1 + 4 * NL);
            tp = curve_double(tp, false);
        }
        Scope s(this, "unblind");
        AffinePointTarget neg_r = curve_neg_const(rando, neg_rando);
        return curve_add(result, neg_r, range_check);
    }
    // the caller's point of a curve program: 9-limb virtual targets set by value
    AffinePointTarget input_point() { return {make_ref(R_SLOT, SLOT_P_PLACEHOLDER), true, AUX_SRC_INPUT | INPUT_PX, AUX_SRC_INPUT_PY}; }
    // stand-alone gadget circuits: point in (px, py), scalar in the msg slot
    void windowed_mul_circuit(u32 g, u32 neg_g, u32 start, u32 spm, u32 neg_spm) {
        curve_scalar_mul_windowed(input_point(), NonNativeTarget{FIELD_SCALAR, AUX_SRC_INPUT | INPUT_MSG, NL}, g, neg_g, start, spm, neg_spm, true);
        finish_curve_program();
    }
    void scalar_mul_circuit(u32 rando, u32 neg_rando) {
        curve_scalar_mul(input_point(), NonNativeTarget{FIELD_SCALAR, AUX_SRC_INPUT | INPUT_MSG, NL}, rando, neg_rando, true);
        finish_curve_program();
    }
    // gadgets/ecdsa.rs:55-78
    void verify_p256_message_circuit(u32 rando32, u32 neg_rando32, u32 g, u32 neg_g, u32 start, u32 spm, u32 neg_spm) {
        prog.full_verify = 1;
        id_rando_ = rando32;
        id_neg_rando_ = neg_rando32;
        prog.sc.assert_valid = (int32_t)col_;
        AffinePointTarget pk = input_point();
        curve_assert_valid(pk);
        NonNativeTarget s{FIELD_SCALAR, AUX_SRC_INPUT | INPUT_S, NL}, msg{FIELD_SCALAR, AUX_SRC_INPUT | INPUT_MSG, NL},
            r{FIELD_SCALAR, AUX_SRC_INPUT | INPUT_R, NL};
        prog.sc.inv_s = (int32_t)col_;
        NonNativeTarget c = inv_nonnative(s);
        prog.sc.u1 = (int32_t)col_;
        NonNativeTarget u1 = mul_nonnative(msg, c, true);
        prog.sc.u2 = (int32_t)col_;
        NonNativeTarget u2 = mul_nonnative(r, c, true);
        AffinePointTarget point1, point2;
        {
            Scope sc(this, "fixed_base");
            point1 = fixed_base_curve_mul_circuit(u1);
        }
        {
            Scope sc(this, "windowed_mul");
            point2 = curve_scalar_mul_windowed(pk, u2, g, neg_g, start, spm, neg_spm, true);
        }
        {
            Scope sc(this, "final_add");
            curve_add(point1, point2, true);
            ops.back().flags |= F_CHECK_R;
        }
        finish_curve_program();
    }

    // ---- curve_msm_circuit(builder, p, q, n, m) ----
    AffinePointTarget curve_msm_circuit(AffinePointTarget p, AffinePointTarget q, NonNativeTarget n, NonNativeTarget m) {
        split_nonnative_to_2_bit_limbs(n);
        split_nonnative_to_2_bit_limbs(m);
        AffinePointTarget rando = constant_affine_point(CONST_RANDO);
        AffinePointTarget neg_rando = constant_affine_point(CONST_NEG_RANDO);
        AffinePointTarget pre[16];
        for (auto& x : pre) x = p;
        {
            Scope s(this, "table");
            AffinePointTarget cur_p = rando, cur_q = rando;
            for (int i = 0; i < 4; i++) {
                pre[i] = cur_p;
                pre[4 * i] = cur_q;
                cur_p = curve_add(cur_p, p);
                cur_q = curve_add(cur_q, q);
            }
            for (int i = 1; i < 4; i++) {
                pre[i] = curve_add(pre[i], neg_rando);
                pre[4 * i] = curve_add(pre[4 * i], neg_rando);
            }
            for (int i = 1; i < 4; i++)
                for (int j = 1; j < 4; j++) pre[i + 4 * j] = curve_add(pre[i], pre[4 * j]);
        }
        for (int i = 0; i < 16; i++) {
            prog.msm_tab[i] = pre[i].ref;
            aux_tab.tabx[i] = pre[i].xcol;
            aux_tab.taby[i] = pre[i].ycol;
        }
        AffinePointTarget result = rando;
        prog.msm_loop_begin = (int32_t)ops.size();
        prog.msm_loop_iters = MSM_DIGITS;
        prog.loop_dbls = 2;
        for (int d = MSM_DIGITS - 1; d >= 0; d--) {
            Scope s(this, "digit" + std::to_string(d));
            result = curve_repeated_double(result, 2);
            // mul_add(four, limb_m, limb_n), random_access_curve_points(index, pre), is_equal, not, conditional add
            // aux columns of this digit: [index, selected x (9), selected y (9), is_zero, should_add, not_b, products ...]
            const u32 base = aux_col_;
            AffinePointTarget r{make_ref(R_MSMTAB, (u32)d), false, AUX_SRC_AUX | (base + 1), AUX_SRC_AUX | (base + 1 + NL)};
            AuxItem it = select_item(AUX_MSMDIG, result);
            it.a = n.col;
            it.c = m.col;
            it.b = (u32)d;
            gate_items.push_back({base, num_gate_cols, 1});   // the digit's index = mul_add(four, limb_m, limb_n): first aux value
            num_gate_cols += GATE_COLS_PER_WINDOW;
            result = curve_conditional_add(result, r, base + 1 + 2 * NL + 2);
            it.sumx = ops.back().col + COL_ADD_X3;
            it.sumy = ops.back().col + COL_ADD_Y3;
            aux(it, 1 + 2 * NL + 2 + 1 + 2 * NL + (u32)it.nlx + (u32)it.nly);
        }
        Scope s(this, "unblind");
        return curve_add(result, constant_affine_point(CONST_NEG_RANDO_146), true);
    }

    // ---- CircuitBuilderGlv ----
    void decompose_secp256k1_scalar(NonNativeTarget k) {
        Scope s(this, "decompose");
        prog.sc.glv = (int32_t)col_;
        // GLVDecompositionGenerator outputs: k1[5], k2[5], k1_neg, k2_neg
        const u32 g = gen(GEN_GLV, FIELD_SCALAR, 12, {k}, false);
        glv_k1_ = {FIELD_SCALAR, g, 5};
        glv_k2_ = {FIELD_SCALAR, g + 5, 5};
        glv_k1_neg_ = {g + 10};
        glv_k2_neg_ = {g + 11};
        NonNativeTarget k1 = nonnative_conditional_neg(glv_k1_, glv_k1_neg_);
        NonNativeTarget k2 = nonnative_conditional_neg(glv_k2_, glv_k2_neg_);
        NonNativeTarget sb = mul_nonnative(constant_nonnative(FIELD_SCALAR, CONSTV_GLV_S), k2);
        add_nonnative(sb, k1, true);
    }
    // p = (px, py): the caller's point (9-limb virtual targets); k: the scalar target
    AffinePointTarget glv_mul(NonNativeTarget k) {
        decompose_secp256k1_scalar(k);
        NonNativeTarget px{FIELD_BASE, AUX_SRC_INPUT | INPUT_PX, NL};
        NonNativeTarget py{FIELD_BASE, AUX_SRC_INPUT_PY, NL};   // the caller's pk.y target (9 limbs)
        prog.sc.beta_x = (int32_t)col_;
        NonNativeTarget beta_px = mul_nonnative(constant_nonnative(FIELD_BASE, CONSTV_GLV_BETA), px, true);
        prog.sc.neg_p = (int32_t)col_;
        NonNativeTarget y1 = nonnative_conditional_neg(py, glv_k1_neg_, true);  // curve_conditional_neg(p, k1_neg)
        prog.sc.neg_sp = (int32_t)col_;
        NonNativeTarget y2 = nonnative_conditional_neg(py, glv_k2_neg_, true);  // curve_conditional_neg(sp, k2_neg)
        Scope s(this, "msm");
        AffinePointTarget p{make_ref(R_SLOT, SLOT_P_PLACEHOLDER), true, px.col, y1.col};
        AffinePointTarget sp{make_ref(R_SLOT, SLOT_SP_PLACEHOLDER), true, beta_px.col, y2.col};
        return curve_msm_circuit(p, sp, glv_k1_, glv_k2_);
    }

    // ---- verify_secp256k1_message_circuit(builder, msg, sig, pk) ----
    void verify_secp256k1_message_circuit() {
        prog.full_verify = 1;
        prog.sc.assert_valid = (int32_t)col_;
        AffinePointTarget pk{0, true, AUX_SRC_INPUT | INPUT_PX, AUX_SRC_INPUT_PY};
        curve_assert_valid(pk);
        NonNativeTarget s{FIELD_SCALAR, AUX_SRC_INPUT | INPUT_S, NL}, msg{FIELD_SCALAR, AUX_SRC_INPUT | INPUT_MSG, NL},
            r{FIELD_SCALAR, AUX_SRC_INPUT | INPUT_R, NL};
        prog.sc.inv_s = (int32_t)col_;
        NonNativeTarget c = inv_nonnative(s);
        prog.sc.u1 = (int32_t)col_;
        NonNativeTarget u1 = mul_nonnative(msg, c, true);
        prog.sc.u2 = (int32_t)col_;
        NonNativeTarget u2 = mul_nonnative(r, c, true);
        int c0 = (int)ops.size();
        AffinePointTarget point1, point2;
        {
            Scope sc(this, "fixed_base");
            point1 = fixed_base_curve_mul_circuit(u1);
        }
        int c1 = (int)ops.size();
        {
            Scope sc(this, "glv_mul");
            point2 = glv_mul(u2);
        }
        int c2 = (int)ops.size();
        {
            Scope sc(this, "final_add");
            curve_add(point1, point2, true);
            ops.back().flags |= F_CHECK_R;
        }
        prog.num_chains = 3;
        prog.chain_begin[0] = c1;
        prog.chain_end[0] = c2;
        prog.chain_begin[1] = c0;
        prog.chain_end[1] = c1;
        prog.chain_begin[2] = c2;
        prog.chain_end[2] = (int)ops.size();
        finish();
    }
    // glv_mul(p, k) alone (BASELINE config 3)
    void glv_mul_circuit() {
        prog.full_verify = 0;
        prog.sc.assert_valid = prog.sc.inv_s = prog.sc.u1 = prog.sc.u2 = -1;
        glv_mul(NonNativeTarget{FIELD_SCALAR, AUX_SRC_INPUT | INPUT_MSG, NL});   // k travels in the msg slot (p2e_glv_mul_witness_batch)
        prog.num_chains = 1;
        prog.chain_begin[0] = 0;
        prog.chain_end[0] = (int)ops.size();
        finish();
    }

    // Expansion runs of `run_iters` loop iterations: inside a run only the last (double, conditional add) pair
    // can be the next run's starting point, every other result never needs affine coordinates in memory.
    void mark_runs(int run_iters, bool mark_fb = true) {
        for (auto& o : ops) o.flags &= (uint8_t)~F_NO_AFFINE;
        if (run_iters < 1) return;
        const int D = prog.loop_dbls;   // doublings per iteration; the conditional add follows them
        for (int it = 0; it < prog.msm_loop_iters; it++) {
            const bool last_of_run = (it % run_iters) == run_iters - 1 || it == prog.msm_loop_iters - 1;
            const int t = prog.msm_loop_begin + (D + 1) * it;
            for (int k = 0; k < D - 1; k++) ops[t + k].flags |= F_NO_AFFINE;
            if (!last_of_run) {
                ops[t + D - 1].flags |= F_NO_AFFINE;
                ops[t + D].flags |= F_NO_AFFINE;
            }
        }
        // the fixed-base windows are expanded as ONE run per signature (body_expand_fb_run): none of their results
        // needs affine coordinates in memory, provided the whole chain is one phase-A piece
        if (mark_fb && prog.fb_begin >= 0)
            for (int w = 0; w < prog.fb_windows; w++) ops[prog.fb_begin + w].flags |= F_NO_AFFINE;
    }

private:
    static constexpr u32 SLOT_P_PLACEHOLDER = 0xFFFF00, SLOT_SP_PLACEHOLDER = 0xFFFF01;
    u32 col_ = 0, aux_col_ = 0, split4_col_ = 0;
    bool generic_ = false;
    int curve_ = 0;   // 0 secp256k1, 1 P-256: UxItem::field = 2 * curve + field
    int32_t loop_begin_ = 0, loop_iters_ = 0;   // curve_scalar_mul_windowed's loop (a curve program's Program::msm_loop_*)
    u32 id_zero_ = CONSTV_ZERO, id_a_ = CONSTV_ZERO, id_b_ = CONSTV_B7;
    int id_rando_ = CONST_RANDO, id_neg_rando_ = CONST_NEG_RANDO;
    void finish_curve_program() {
        prog.num_chains = 1;
        prog.chain_begin[0] = 0;
        prog.chain_end[0] = (int)ops.size();
        prog.msm_loop_begin = loop_begin_;
        prog.msm_loop_iters = loop_iters_;
        finish();
        for (auto& o : ops)
            if (ref_kind(o.ref2) == R_SELSLOT && (ref_id(o.ref2) & 0xFFFu) == 0xFFFu)
                o.ref2 = make_ref(R_SELSLOT, (ref_id(o.ref2) & ~0xFFFu) | (u32)prog.slot_p);
    }
    int num_cadd_ = 0;
    NonNativeTarget glv_k1_{FIELD_SCALAR}, glv_k2_{FIELD_SCALAR};
    BoolTarget glv_k1_neg_{0}, glv_k2_neg_{0};
    std::vector<std::string> path_;

    struct Scope {
        ScheduleBuilder* b;
        Scope(ScheduleBuilder* b_, const std::string& n) : b(b_) { b->path_.push_back(n); }
        ~Scope() { b->path_.pop_back(); }
    };
    std::string label() const {
        std::string s;
        for (size_t i = 0; i < path_.size(); i++) s += (i ? "/" : "") + path_[i];
        return s;
    }
    u32 gen(int kind, int field, u32 ncols, const std::vector<NonNativeTarget>& operands, bool range_check) {
        GenOp g{kind, field, col_, ncols, label()};   // returns the generator's first column
        g.nops = (int)operands.size();
        assert(g.nops <= 4);
        for (int k = 0; k < g.nops; k++) {
            g.src[k] = operands[(size_t)k].col;
            g.nl[k] = (uint8_t)operands[(size_t)k].nl;
        }
        g.range_check = range_check;
        gens.push_back(g);
        col_ += ncols;
        return col_ - ncols;
    }
    static NonNativeTarget x_of(const AffinePointTarget& p) { return {FIELD_BASE, p.xcol, p.nlx}; }
    static NonNativeTarget y_of(const AffinePointTarget& p) { return {FIELD_BASE, p.ycol, p.nly}; }
    void aux(AuxItem it, u32 ncols) {
        it.aux_col = aux_col_;
        it.ncols = ncols;
        aux_items.push_back(it);
        aux_gens.push_back({(int)it.kind, aux_col_, ncols, label()});
        aux_col_ += ncols;
    }
    // the part of a window's item that comes from curve_conditional_add(p1, ...): where p1's limbs live
    static AuxItem select_item(AuxKind kind, const AffinePointTarget& p1) {
        AuxItem it{};
        it.kind = kind;
        it.p1x = p1.xcol;
        it.p1y = p1.ycol;
        it.nlx = (uint8_t)p1.nlx;
        it.nly = (uint8_t)p1.nly;
        return it;
    }
public:
    // the value behind a constant source code AUX_SRC_CONST | id (aux.hpp)
    static U256 const_value(u32 id) {
        const Consts& C = consts();
        if (id < 2 * NUM_CONST_PTS) return (id & 1) ? C.cpts[id >> 1].y : C.cpts[id >> 1].x;
        if (id == CONSTV_B7) return u256_small(7);                                   // curve/secp256k1.rs:16
        if (id == CONSTV_GLV_S)                                                      // curve/glv.rs:18-23
            return u256_from_u64(16069571880186789234ull, 1310022930574435960ull, 11900229862571533402ull, 6008836872998760672ull);
        if (id == CONSTV_GLV_BETA)                                                   // curve/glv.rs:11-16
            return u256_from_u64(13923278643952681454ull, 11308619431505398165ull, 7954561588662645993ull, 8856726876819556112ull);
        return u256_zero();                                                          // CONSTV_ZERO
    }
private:
    static int const_num_limbs(const U256& v) {   // constant_biguint gadgets/biguint.rs:165-175 via convert_base :27-51
        u32 l[NL];
        split29(v, l);
        int n = NL;
        while (n > 0 && l[n - 1] == 0) n--;
        return n;
    }
    // sel_layout (conditional adds): order of the four bool products behind not_b -- 0: sum.x, sum.y, p1.x, p1.y
    // (curve_conditional_add gadgets/curve.rs:234-238), 1: sum.x, p1.x, sum.y, p1.y (curve_scalar_mul :262-265)
    AffinePointTarget curve_op(OpKind kind, AffinePointTarget p1, AffinePointTarget p2, bool range_check, u32 not_b_aux = 0,
                               int sel_layout = 0) {
        OpDesc d{};
        d.kind = kind;
        d.flags = (uint8_t)((p1.z_one ? F_Z1ONE : 0) | ((kind != OP_DBL && p2.z_one) ? F_Z2ONE : 0));
        d.ref1 = p1.ref;
        d.ref2 = kind == OP_DBL ? 0 : p2.ref;
        d.col = col_;
        u32 t = (u32)ops.size();
        const NonNativeTarget x1 = x_of(p1), y1 = y_of(p1), x2 = x_of(p2), y2 = y_of(p2);
        NonNativeTarget x3, y3;
        if (kind == OP_DBL) {  // gadgets/curve.rs:160-185
            NonNativeTarget dy = add_nonnative(y1, y1);
            NonNativeTarget idy = inv_nonnative(dy);
            NonNativeTarget xx = mul_nonnative(x1, x1);
            NonNativeTarget tr = add_many_nonnative({xx, xx, xx, constant_nonnative(FIELD_BASE, id_a_)});
            NonNativeTarget lam = mul_nonnative(tr, idy);
            NonNativeTarget lam2 = mul_nonnative(lam, lam);
            NonNativeTarget xd = add_nonnative(x1, x1);
            x3 = sub_nonnative(lam2, xd, range_check);
            NonNativeTarget xdf = sub_nonnative(x1, x3);
            NonNativeTarget lx = mul_nonnative(lam, xdf);
            y3 = sub_nonnative(lx, y1, range_check);
        } else {  // gadgets/curve.rs:202-223; a conditional add computes the sum unchecked (:233) ...
            const bool rc = kind == OP_CADD ? false : range_check;
            NonNativeTarget u = sub_nonnative(y2, y1);
            NonNativeTarget v = sub_nonnative(x2, x1);
            NonNativeTarget vinv = inv_nonnative(v);
            NonNativeTarget sl = mul_nonnative(u, vinv);
            NonNativeTarget s2 = mul_nonnative(sl, sl);
            NonNativeTarget xs = add_nonnative(x2, x1);
            x3 = sub_nonnative(s2, xs, rc);
            NonNativeTarget xd = sub_nonnative(x1, x3);
            NonNativeTarget pr = mul_nonnative(sl, xd);
            y3 = sub_nonnative(pr, y1, rc);
            if (kind == OP_CADD) {  // ... then selects: add(sum.x * b, p1.x * not_b), add(sum.y * b, p1.y * not_b)  :234-240
                u32 xt = not_b_aux + 1, yt = xt + NL, xf = yt + NL, yf = xf + (u32)p1.nlx;
                if (sel_layout == 1) {
                    xf = xt + NL;
                    yt = xf + (u32)p1.nlx;
                    yf = yt + NL;
                }
                const int F = FIELD_BASE;
                add_nonnative({F, AUX_SRC_AUX | xt, NL}, {F, AUX_SRC_AUX | xf, p1.nlx}, range_check);
                add_nonnative({F, AUX_SRC_AUX | yt, NL}, {F, AUX_SRC_AUX | yf, p1.nly}, range_check);
            }
        }
        (void)x3;
        (void)y3;
        AffinePointTarget out;
        if (kind == OP_CADD) {
            d.cadd_idx = (uint16_t)num_cadd_;
            out = {make_ref(R_DYN, (u32)num_cadd_), false, d.col + COL_CADD_X, d.col + COL_CADD_Y};
            num_cadd_++;
        } else if (kind == OP_DBL) {
            out = {make_ref(R_SLOT, t), false, d.col + COL_DBL_X3, d.col + COL_DBL_Y3};
        } else {
            out = {make_ref(R_SLOT, t), false, d.col + COL_ADD_X3, d.col + COL_ADD_Y3};
        }
        ops.push_back(d);
        return out;
    }
    void finish() {
        prog.num_ops = (int32_t)ops.size();
        prog.slot_p = prog.num_ops;
        prog.slot_sp = prog.num_ops + 1;
        prog.num_slots = prog.num_ops + 2;
        prog.num_cadd = num_cadd_;
        prog.num_cols = (int32_t)col_;
        aux_tab.num_aux_cols = aux_col_;
        auto fix = [&](u32& r) {
            if (ref_kind(r) == R_SLOT && ref_id(r) == SLOT_P_PLACEHOLDER) r = make_ref(R_SLOT, (u32)prog.slot_p);
            if (ref_kind(r) == R_SLOT && ref_id(r) == SLOT_SP_PLACEHOLDER) r = make_ref(R_SLOT, (u32)prog.slot_sp);
        };
        for (auto& o : ops) {
            fix(o.ref1);
            fix(o.ref2);
        }
        for (auto& r : prog.msm_tab) fix(r);
        // constraint-block columns, builder-call order = generator order
        for (const GenOp& g : gens) {
            u32 nc = 0;
            UxItem it{};
            it.field = (uint8_t)(2 * curve_ + g.field);
            it.nops = (uint8_t)g.nops;
            it.range_check = g.range_check ? 1 : 0;
            for (int k = 0; k < 4; k++) {
                it.src[k] = g.src[k];
                it.nl[k] = g.nl[k];
            }
            it.res_col = g.col;
            switch (g.kind) {
                case GEN_ADD: it.kind = UX_ADD; nc = UX_COLS_ADD; break;
                case GEN_SUB: it.kind = UX_SUB; nc = UX_COLS_SUB; break;
                case GEN_ADD_MANY: it.kind = UX_ADD_MANY; nc = UX_COLS_ADD_MANY; assert(g.nops == 4); break;
                case GEN_INV: it.kind = UX_INV; nc = UX_COLS_INV; assert(g.nl[0] == NL); break;
                case GEN_MUL: it.kind = UX_MUL; nc = 0; break;
                default: nc = 0; it.range_check = 0; break;   // GLV decomposition: constrained through the ops after it
            }
            if (g.kind != GEN_GLV) nc += g.range_check ? 1u : 0u;
            ux_first.push_back(num_ux_cols);
            ux_count.push_back(nc);
            if (nc) {
                it.ux_col = num_ux_cols;
                it.ncols = nc;
                ux_items.push_back(it);
            }
            num_ux_cols += nc;
        }
    }
};

}  // namespace host
}  // namespace p2e
