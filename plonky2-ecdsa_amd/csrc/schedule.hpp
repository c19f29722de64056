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
#pragma once
#include <string>
#include <vector>

#include "pipeline.hpp"

namespace p2e {
namespace host {

enum GenKind : int { GEN_ADD = 0, GEN_SUB = 1, GEN_ADD_MANY = 2, GEN_MUL = 3, GEN_INV = 4, GEN_GLV = 5 };
enum Field : int { FIELD_BASE = 0, FIELD_SCALAR = 1 };

struct GenOp {
    int kind, field;
    u32 col, ncols;
    std::string label;
};

// symbolic NonNativeTarget: the hot path never needs its value on the host, only its identity
struct NonNativeTarget {
    int field;
};
// symbolic AffinePointTarget: where the GPU finds the point
struct AffinePointTarget {
    u32 ref;
    bool z_one;  // known to be stored in affine form before phase B (constants, inputs)
};

class ScheduleBuilder {
public:
    std::vector<GenOp> gens;
    std::vector<OpDesc> ops;
    Program prog{};

    // ---- CircuitBuilderNonNative ----
    NonNativeTarget add_nonnative(NonNativeTarget a, NonNativeTarget, bool /*range_check*/ = false) {
        gen(GEN_ADD, a.field, 10);
        return a;
    }
    NonNativeTarget sub_nonnative(NonNativeTarget a, NonNativeTarget, bool = false) {
        gen(GEN_SUB, a.field, 10);
        return a;
    }
    NonNativeTarget add_many_nonnative(const std::vector<NonNativeTarget>& xs, bool = false) {
        if (xs.size() == 1) return xs[0];
        gen(GEN_ADD_MANY, xs[0].field, 10);
        return xs[0];
    }
    NonNativeTarget mul_nonnative(NonNativeTarget a, NonNativeTarget, bool = false) {
        gen(GEN_MUL, a.field, 51);  // MulNonnativeGate r,q,check_sum (35) + CheckSumGate b (16)
        return a;
    }
    NonNativeTarget inv_nonnative(NonNativeTarget a, bool = false) {
        gen(GEN_INV, a.field, 18);
        return a;
    }
    NonNativeTarget neg_nonnative(NonNativeTarget a, bool = false) { return sub_nonnative(a, a); }
    NonNativeTarget nonnative_conditional_neg(NonNativeTarget x, bool = false) {
        NonNativeTarget neg = neg_nonnative(x);
        return add_nonnative(neg, x);
    }

    // ---- CircuitBuilderCurve ----
    void curve_assert_valid() {
        Scope s(this, "assert_valid");
        NonNativeTarget f{FIELD_BASE};
        mul_nonnative(f, f, true);
        mul_nonnative(f, f);
        mul_nonnative(f, f);
        mul_nonnative(f, f);
        add_nonnative(f, f);
        add_nonnative(f, f, true);
    }
    AffinePointTarget curve_add(AffinePointTarget p1, AffinePointTarget p2, bool = false) {
        return curve_op(OP_ADD, p1, p2);
    }
    AffinePointTarget curve_double(AffinePointTarget p, bool = false) { return curve_op(OP_DBL, p, p); }
    AffinePointTarget curve_repeated_double(AffinePointTarget p, int n, bool = false) {
        for (int i = 0; i < n; i++) p = curve_double(p);
        return p;
    }
    // b is always "random-access index != 0" on the supported circuits; it is derived from p2's ref
    AffinePointTarget curve_conditional_add(AffinePointTarget p1, AffinePointTarget p2, bool = false) {
        return curve_op(OP_CADD, p1, p2);
    }
    AffinePointTarget constant_affine_point(int which) { return {make_ref(R_CONST, (u32)which), true}; }

    // ---- fixed_base_curve_mul_circuit(builder, G, scalar) ----
    AffinePointTarget fixed_base_curve_mul_circuit() {
        AffinePointTarget result = constant_affine_point(CONST_RANDO);
        for (int w = 0; w < FB_WINDOWS; w++) {
            Scope s(this, "win" + std::to_string(w));
            AffinePointTarget r{make_ref(R_FBTAB, (u32)w), true};  // random_access_curve_points(limb, muls_point)
            result = curve_conditional_add(result, r);
        }
        Scope s(this, "unblind");
        return curve_add(result, constant_affine_point(CONST_NEG_RANDO), true);
    }

    // ---- curve_msm_circuit(builder, p, q, n, m) ----
    AffinePointTarget curve_msm_circuit(AffinePointTarget p, AffinePointTarget q) {
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
        for (int i = 0; i < 16; i++) prog.msm_tab[i] = pre[i].ref;
        AffinePointTarget result = rando;
        prog.msm_loop_begin = (int32_t)ops.size();
        prog.msm_loop_iters = MSM_DIGITS;
        for (int d = MSM_DIGITS - 1; d >= 0; d--) {
            Scope s(this, "digit" + std::to_string(d));
            result = curve_repeated_double(result, 2);
            AffinePointTarget r{make_ref(R_MSMTAB, (u32)d), false};  // random_access_curve_points(index, pre)
            result = curve_conditional_add(result, r);
        }
        Scope s(this, "unblind");
        return curve_add(result, constant_affine_point(CONST_NEG_RANDO_146), true);
    }

    // ---- CircuitBuilderGlv ----
    void decompose_secp256k1_scalar() {
        Scope s(this, "decompose");
        prog.sc.glv = (int32_t)col_;
        NonNativeTarget f{FIELD_SCALAR};
        gen(GEN_GLV, FIELD_SCALAR, 12);
        NonNativeTarget k1 = nonnative_conditional_neg(f);
        NonNativeTarget k2 = nonnative_conditional_neg(f);
        NonNativeTarget sb = mul_nonnative(f, k2);
        add_nonnative(sb, k1, true);
    }
    AffinePointTarget glv_mul(int chain_slot_base_hint = 0) {
        (void)chain_slot_base_hint;
        decompose_secp256k1_scalar();
        NonNativeTarget b{FIELD_BASE};
        prog.sc.beta_x = (int32_t)col_;
        mul_nonnative(b, b, true);
        prog.sc.neg_p = (int32_t)col_;
        nonnative_conditional_neg(b, true);  // curve_conditional_neg(p, k1_neg)
        prog.sc.neg_sp = (int32_t)col_;
        nonnative_conditional_neg(b, true);  // curve_conditional_neg(sp, k2_neg)
        Scope s(this, "msm");
        AffinePointTarget p{make_ref(R_SLOT, SLOT_P_PLACEHOLDER), true}, sp{make_ref(R_SLOT, SLOT_SP_PLACEHOLDER), true};
        return curve_msm_circuit(p, sp);
    }

    // ---- verify_secp256k1_message_circuit(builder, msg, sig, pk) ----
    void verify_secp256k1_message_circuit() {
        prog.full_verify = 1;
        prog.sc.assert_valid = (int32_t)col_;
        curve_assert_valid();
        NonNativeTarget s{FIELD_SCALAR};
        prog.sc.inv_s = (int32_t)col_;
        NonNativeTarget c = inv_nonnative(s);
        prog.sc.u1 = (int32_t)col_;
        mul_nonnative(s, c, true);
        prog.sc.u2 = (int32_t)col_;
        mul_nonnative(s, c, true);
        int c0 = (int)ops.size();
        AffinePointTarget point1, point2;
        {
            Scope sc(this, "fixed_base");
            point1 = fixed_base_curve_mul_circuit();
        }
        int c1 = (int)ops.size();
        {
            Scope sc(this, "glv_mul");
            point2 = glv_mul();
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
        glv_mul();
        prog.num_chains = 1;
        prog.chain_begin[0] = 0;
        prog.chain_end[0] = (int)ops.size();
        finish();
    }

    // Expansion runs of `run_iters` loop iterations: inside a run only the last (double, conditional add) pair
    // can be the next run's starting point, every other result never needs affine coordinates in memory.
    void mark_runs(int run_iters) {
        for (auto& o : ops) o.flags &= (uint8_t)~F_NO_AFFINE;
        if (run_iters < 1) return;
        for (int it = 0; it < prog.msm_loop_iters; it++) {
            const bool last_of_run = (it % run_iters) == run_iters - 1 || it == prog.msm_loop_iters - 1;
            const int t = prog.msm_loop_begin + 3 * it;
            ops[t].flags |= F_NO_AFFINE;
            if (!last_of_run) {
                ops[t + 1].flags |= F_NO_AFFINE;
                ops[t + 2].flags |= F_NO_AFFINE;
            }
        }
    }

private:
    static constexpr u32 SLOT_P_PLACEHOLDER = 0xFFFF00, SLOT_SP_PLACEHOLDER = 0xFFFF01;
    u32 col_ = 0;
    int num_cadd_ = 0;
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
    void gen(int kind, int field, u32 ncols) {
        gens.push_back({kind, field, col_, ncols, label()});
        col_ += ncols;
    }
    AffinePointTarget curve_op(OpKind kind, AffinePointTarget p1, AffinePointTarget p2) {
        OpDesc d{};
        d.kind = kind;
        d.flags = (uint8_t)((p1.z_one ? F_Z1ONE : 0) | ((kind != OP_DBL && p2.z_one) ? F_Z2ONE : 0));
        d.ref1 = p1.ref;
        d.ref2 = kind == OP_DBL ? 0 : p2.ref;
        d.col = col_;
        u32 t = (u32)ops.size();
        NonNativeTarget f{FIELD_BASE};
        if (kind == OP_DBL) {  // gadgets/curve.rs:160-185
            add_nonnative(f, f);
            inv_nonnative(f);
            mul_nonnative(f, f);
            add_many_nonnative({f, f, f, f});
            mul_nonnative(f, f);
            mul_nonnative(f, f);
            add_nonnative(f, f);
            sub_nonnative(f, f);
            sub_nonnative(f, f);
            mul_nonnative(f, f);
            sub_nonnative(f, f);
        } else {  // gadgets/curve.rs:202-223
            sub_nonnative(f, f);
            sub_nonnative(f, f);
            inv_nonnative(f);
            mul_nonnative(f, f);
            mul_nonnative(f, f);
            add_nonnative(f, f);
            sub_nonnative(f, f);
            sub_nonnative(f, f);
            mul_nonnative(f, f);
            sub_nonnative(f, f);
            if (kind == OP_CADD) {  // gadgets/curve.rs:225-243
                add_nonnative(f, f);
                add_nonnative(f, f);
            }
        }
        AffinePointTarget out;
        if (kind == OP_CADD) {
            d.cadd_idx = (uint16_t)num_cadd_;
            out = {make_ref(R_DYN, (u32)num_cadd_), false};
            num_cadd_++;
        } else {
            out = {make_ref(R_SLOT, t), false};
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
        auto fix = [&](u32& r) {
            if (ref_kind(r) == R_SLOT && ref_id(r) == SLOT_P_PLACEHOLDER) r = make_ref(R_SLOT, (u32)prog.slot_p);
            if (ref_kind(r) == R_SLOT && ref_id(r) == SLOT_SP_PLACEHOLDER) r = make_ref(R_SLOT, (u32)prog.slot_sp);
        };
        for (auto& o : ops) {
            fix(o.ref1);
            fix(o.ref2);
        }
        for (auto& r : prog.msm_tab) fix(r);
    }
};

}  // namespace host
}  // namespace p2e
