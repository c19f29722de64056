// Host-side construction of a curve program (curves.hpp): its constants and its schedule (HOST ONLY).
//   blinding / starting points   gadgets/curve_windowed_mul.rs:57,140-152 ; gadgets/curve.rs:253 ; gadgets/curve_fixed_base.rs:34-38
//   fixed-base table of the curve's generator   gadgets/curve_fixed_base.rs:24-30,45-56
//   synthetic signatures on either curve        curve/ecdsa.rs:25-40 sign_message
#pragma once
#include "schedule.hpp"

namespace p2e {
namespace host {

template <class CV>
inline Aff generator_cv();
template <>
inline Aff generator_cv<Secp256k1>() { return generator(); }
template <>
inline Aff generator_cv<P256>() {   // curve/p256.rs:43-57
    Aff g;
    g.x = u256_from_u64(0xF4A13945D898C296ull, 0x77037D812DEB33A0ull, 0xF8BCE6E563A440F2ull, 0x6B17D1F2E12C4247ull);
    g.y = u256_from_u64(0xCBB6406837BF51F5ull, 0x2BCE33576B315ECEull, 0x8EE7EB4A7C0F9E16ull, 0x4FE342E2FE1A7F9Bull);
    return g;
}
template <class CV>
inline Aff jac_to_aff_cv(const Jac& j) {
    typedef typename CV::Fp F;
    U256 zi = fe_inv<F>(j.Z);
    U256 zi2 = fe_sqr<F>(zi);
    Aff a;
    a.x = fe_mul<F>(j.X, zi2);
    a.y = fe_mul<F>(j.Y, fe_mul<F>(zi2, zi));
    return a;
}
// y^2 == x^3 + a x + b in CV::Fp (canonical coordinates)
template <class CV>
inline bool aff_on_curve_cv(const Aff& p) {
    typedef typename CV::Fp F;
    const U256 lhs = fe_sqr<F>(p.y);
    const U256 rhs = fe_add<F>(fe_mul<F>(fe_add<F>(fe_sqr<F>(p.x), CV::a()), p.x), CV::b());
    return u256_eq(fe_canon<F>(lhs), fe_canon<F>(rhs));
}
// k*P for 0 < k < n, P of prime order (left-to-right double-and-add: the running point is never +-P before an add)
template <class CV>
inline Aff scalar_mul_cv(const U256& k, const Aff& p) {
    int top = 255;
    while (top >= 0 && !((k.w[top >> 5] >> (top & 31)) & 1)) top--;
    Jac acc = jac_from_aff(p);
    const Jac pj = jac_from_aff(p);
    for (int i = top - 1; i >= 0; i--) {
        acc = jac_dbl_cv<CV>(acc).p;
        if ((k.w[i >> 5] >> (i & 31)) & 1) acc = jac_add_cv<CV, false, true>(acc, pj).p;
    }
    return jac_to_aff_cv<CV>(acc);
}
// keccak256 of the 8-byte little-endian encoding of Goldilocks zero, first `nbytes` bytes read as a little-endian
// integer (KeccakHash::<N>::hash_no_pad(&[F::ZERO]); digest re-derived in tests/ with an independent Keccak)
inline U256 hash0_scalar(int nbytes) {
    static const uint8_t h[32] = {0x01, 0x1b, 0x4d, 0x03, 0xdd, 0x8c, 0x01, 0xf1, 0x04, 0x91, 0x43,
                                  0xcf, 0x9c, 0x4c, 0x81, 0x7e, 0x4b, 0x16, 0x7f, 0x1d, 0x1b, 0x83,
                                  0xe5, 0xc6, 0xf0, 0xf1, 0x0d, 0x89, 0xba, 0x1e, 0x7b, 0xce};
    U256 v = u256_zero();
    for (int i = 0; i < nbytes && i < 32; i++) v.w[i >> 2] |= (u32)h[i] << (8 * (i & 3));
    return v;
}
template <class CV>
inline std::vector<Aff> fixed_base_table_cv() {   // [66][16], slot 0 := slot 1 (gadgets/curve_fixed_base.rs:56)
    std::vector<Aff> t((size_t)FB_WINDOWS * 16);
    Aff base = generator_cv<CV>();
    for (int w = 0; w < FB_WINDOWS; w++) {
        Aff acc = base;
        for (int k = 1; k < 16; k++) {
            t[(size_t)w * 16 + k] = acc;
            if (k < 15) acc = (k == 1) ? aff_dbl_cv<CV>(acc) : aff_add_cv<CV>(acc, base);
        }
        t[(size_t)w * 16] = t[(size_t)w * 16 + 1];
        for (int i = 0; i < 4; i++) base = aff_dbl_cv<CV>(base);
    }
    return t;
}

struct CurveProgramHost {
    int kind = 0, curve = 0;
    ScheduleBuilder sb;
};

// blind: the point the gadget draws with rand() at circuit-build time (precompute_window's g / curve_scalar_mul's rando)
template <class CV>
inline void build_curve_program(CurveProgramHost& H, int kind, const Aff& blind) {
    ScheduleBuilder& sb = H.sb;
    sb.begin_curve_program(kind, H.curve, CV::a(), CV::b());
    const Aff G = generator_cv<CV>();
    if (kind == CP_SCALAR_MUL) {
        const u32 rando = sb.add_const_point(blind), neg = sb.add_const_point(aff_neg_cv<CV>(blind));
        sb.scalar_mul_circuit(rando, neg);
        return;
    }
    u32 rando32 = 0, neg_rando32 = 0;
    if (kind == CP_VERIFY) {
        const Aff r = scalar_mul_cv<CV>(hash0_scalar(32), G);
        rando32 = sb.add_const_point(r);
        neg_rando32 = sb.add_const_point(aff_neg_cv<CV>(r));
        sb.gfbtab = fixed_base_table_cv<CV>();
    }
    const Aff start = scalar_mul_cv<CV>(hash0_scalar(25), G);
    Aff spm = start;
    for (int i = 0; i < CP_WINDOWS * 4; i++) spm = aff_dbl_cv<CV>(spm);
    const u32 g = sb.add_const_point(blind), neg_g = sb.add_const_point(aff_neg_cv<CV>(blind));
    const u32 st = sb.add_const_point(start), sp = sb.add_const_point(spm), nsp = sb.add_const_point(aff_neg_cv<CV>(spm));
    if (kind == CP_VERIFY)
        sb.verify_p256_message_circuit(rando32, neg_rando32, g, neg_g, st, sp, nsp);
    else
        sb.windowed_mul_circuit(g, neg_g, st, sp, nsp);
}
// curve: 0 secp256k1, 1 P-256 (include/p2e.h P2E_CURVE_*).  The verifier program exists for P-256 only: secp256k1's
// verifier is the built-in program 0 (it multiplies by GLV, not by windows).
inline bool make_curve_program(CurveProgramHost& H, int kind, int curve, const Aff& blind) {
    if (kind != CP_WINDOWED && kind != CP_SCALAR_MUL && kind != CP_VERIFY) return false;
    if (curve != 0 && curve != 1) return false;
    if (kind == CP_VERIFY && curve != 1) return false;
    H.kind = kind;
    H.curve = curve;   // (read by build_curve_program)
    if (curve == 0)
        build_curve_program<Secp256k1>(H, kind, blind);
    else
        build_curve_program<P256>(H, kind, blind);
    return true;
}

// synthetic valid signatures on a curve of the crate (same splitmix64 stream layout as synth_signature)
template <class CV>
inline void synth_signature_cv(u64 seed, u64 index, U256& msg, U256& r, U256& s, Aff& pk) {
    typedef typename CV::Fn Fn;
    SplitMix64 rng{seed ^ (0x9E3779B97F4A7C15ull * (index + 1))};
    auto below_n = [&]() {
        for (;;) {
            u64 a = rng.next(), b = rng.next(), c = rng.next(), d = rng.next();
            U256 v = u256_from_u64(a, b, c, d);
            if (!u256_is_zero(v) && !geq_mod<Fn>(v.w)) return v;
        }
    };
    const Aff G = generator_cv<CV>();
    for (;;) {
        U256 sk = below_n();
        msg = below_n();
        U256 k = below_n();
        pk = scalar_mul_cv<CV>(sk, G);
        Aff rr = scalar_mul_cv<CV>(k, G);
        r = fe_canon<Fn>(rr.x);
        s = fe_mul<Fn>(fe_inv<Fn>(k), fe_add<Fn>(msg, fe_mul<Fn>(r, sk)));
        if (u256_is_zero(r) || u256_is_zero(s)) continue;
        return;
    }
}

}  // namespace host
}  // namespace p2e
