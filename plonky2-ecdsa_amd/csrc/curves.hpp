// Curve programs: the crate's other scalar-multiplication gadgets on either of its curves,
// run through the same four phases as the built-in programs (pipeline.hpp) with the curve as a template parameter:
//   CP_WINDOWED    curve_scalar_mul_windowed(p, n)         gadgets/curve_windowed_mul.rs:131-173 (+ :52-72 precompute_window)
//   CP_SCALAR_MUL  curve_scalar_mul(p, n)                  gadgets/curve.rs:245-285
//   CP_VERIFY      verify_p256_message_circuit             gadgets/ecdsa.rs:55-78 (fixed base + windowed + final add)
// Both gadgets blind with a point drawn by rand() while the circuit is built (curve_windowed_mul.rs:57, curve.rs:253):
// a program is therefore created once per circuit with that point as an argument (p2e_curve_program_create), which is
// what "the witness is per build" means at this boundary.
//
// This header holds the scalar phase of these programs; phases A, B, C are body_chain_range<CV, true>,
// body_batch_inv<CV> and body_expand<E, CV> of pipeline.hpp.
#pragma once
#include "pipeline.hpp"

namespace p2e {

enum CurveProgKind : int32_t { CP_NONE = 0, CP_WINDOWED = 1, CP_SCALAR_MUL = 2, CP_VERIFY = 3 };
constexpr int CP_WINDOWS = 66;       // split_nonnative_to_4_bit_limbs of a 9-limb scalar: 261 bits -> 264 -> 66 windows
constexpr int CP_BITS = NL * BITS;   // split_nonnative_to_bits: 261

// Phase S of a curve program.  Inputs as in Buffers (stand-alone multiplications: the scalar travels in `msg`).
// Writes: the scalar-field generators of the verifier, the digit rows (dig4: fixed-base windows of u1; dig2: windows /
// bits of the scalar that multiplies the caller's point; msrc: the per-signature table entry each window selects), the
// one curve_neg(constant) generator, and the caller's point in its scratch slot.
template <class CV, class E>
P2E_HD void body_cscalar(const Program& G, const Buffers& B, size_t i) {
    typedef typename CV::Fp Fp;
    typedef typename CV::Fn Fn;
    uint8_t err = 0;
    bool ok = true;
    const U256 px = load_packed(B.pkx, i), py = load_packed(B.pky, i);
    U256 k;
    if (G.cp_kind == CP_VERIFY) {
        const U256 msg = load_packed(B.msg, i), s = load_packed(B.s, i), r = load_packed(B.r, i);
        {   // curve_assert_valid gadgets/curve.rs:123-135
            E e = E::at(B.sink, i, (u32)G.sc.assert_valid);
            U256 y2 = wit_mul<Fp>(e, py, py, err);
            U256 x2 = wit_mul<Fp>(e, px, px, err);
            U256 x3 = wit_mul<Fp>(e, x2, px, err);
            U256 ax = wit_mul<Fp>(e, CV::a(), px, err);
            U256 axb = wit_add<Fp>(e, ax, CV::b());
            U256 rhs = wit_add<Fp>(e, x3, axb);
            ok = ok && u256_eq(y2, rhs);
            e.flush();
        }
        E e = E::at(B.sink, i, (u32)G.sc.inv_s);
        U256 c = wit_inv<Fn>(e, s, err);          // gadgets/ecdsa.rs:65
        U256 u1 = wit_mul<Fn>(e, msg, c, err);    // :66
        k = wit_mul<Fn>(e, r, c, err);            // :67
        e.flush();
        for (int w = 0; w < FB_WINDOWS; w++) B.dig4[(size_t)w * B.n + i] = (uint8_t)digit_of<4>(u1, w);
    } else {
        k = load_packed(B.msg, i);
    }
    if (G.cp_kind == CP_SCALAR_MUL) {
        for (int b = 0; b < CP_BITS; b++) B.dig2[(size_t)b * B.n + i] = (uint8_t)digit_of<1>(k, b);
    } else {
        for (int w = 0; w < CP_WINDOWS; w++) {
            const u32 d = digit_of<4>(k, w);
            B.dig2[(size_t)w * B.n + i] = (uint8_t)d;
            const u32 tr = G.msm_tab[d];
            B.msrc[(size_t)w * B.n + i] = ref_kind(tr) == R_CONST ? (uint16_t)(ref_id(tr) | DYN_CONST_BIT) : (uint16_t)ref_id(tr);
        }
    }
    {   // curve_neg(constant) = neg_nonnative(y) = sub(0, y)  (gadgets/curve.rs:137-147, gadgets/nonnative.rs:491-500)
        E e = E::at(B.sink, i, (u32)G.cp_neg_col);
        (void)wit_sub<Fp>(e, u256_zero(), B.cpts[G.cp_neg_const].y);
        e.flush();
    }
    // the caller's point: canonical coordinates for the Jacobian walk; the affine slot keeps the RAW limbs, because
    // mul_nonnative reads its operands' limbs as they are (quirk Q3: curve_scalar_mul's first doubling squares p.x)
    const size_t sp = (size_t)G.slot_p * B.n + i;
    B.PX[sp] = fe_canon<Fp>(px);
    B.PY[sp] = fe_canon<Fp>(py);
    B.AX[sp] = px;
    B.AY[sp] = py;
    B.err[i] = err;
    B.valid[i] = ok ? 1 : 0;
}

}  // namespace p2e
