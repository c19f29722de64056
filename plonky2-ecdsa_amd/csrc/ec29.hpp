// ec.hpp's Jacobian formulas for secp256k1 on fe29.hpp's lazy 29-bit limbs: the same multiplications in the same
// roles (same X3, Y3, Z3 and W as field elements -- what is stored is their canonical form, bit for bit what ec.hpp
// yields), with the additions between them carry-free and a value brought back to tight limbs (f29_norm) only where a
// multiplication's 64-bit columns need it.  The chains are latency-bound at every batch size (one wave per SIMD at
// 2^16 signatures per call, fewer below): a multiplication costs a lone wave 348 instead of 648 ns, an addition 73
// instead of 161 ns (tools/ubench/fe29_latency.hip).  Curve formulas: reference curve/curve_types.rs:192-262 in
// Jacobian form, see ec.hpp for W.
#pragma once
#include "ec.hpp"
#include "fe29.hpp"

namespace p2e {

// which curves have a lazy-limb form of their base field (P-256's prime has no short fold: it keeps fe.hpp)
template <class CV>
struct LazyLimbs {
    static constexpr bool available = false;
};
template <>
struct LazyLimbs<Secp256k1> {
    static constexpr bool available = true;
};

struct JacL {
    F29 X, Y, Z;   // tight limbs
};
struct JacWL {
    JacL p;
    F29 W;
};
P2E_HD JacL jacl_from(const Jac& p) {
    JacL r;
    r.X = f29_from_u256(p.X);
    r.Y = f29_from_u256(p.Y);
    r.Z = f29_from_u256(p.Z);
    return r;
}
P2E_HD Jac jacl_canon(const JacL& p) {
    Jac r;
    r.X = f29_canon(p.X);
    r.Y = f29_canon(p.Y);
    r.Z = f29_canon(p.Z);
    return r;
}
P2E_HD JacL jacl_select3(bool c1, const JacL& a, bool c2, const JacL& b, const JacL& c) {
    JacL r;
    r.X = f29_select(c1, a.X, f29_select(c2, b.X, c.X));
    r.Y = f29_select(c1, a.Y, f29_select(c2, b.Y, c.Y));
    r.Z = f29_select(c1, a.Z, f29_select(c2, b.Z, c.Z));
    return r;
}

// jac_add_cv<Secp256k1, Z1ONE, Z2ONE> (ec.hpp)
template <bool Z1ONE, bool Z2ONE>
P2E_HD JacWL jac_add29(const JacL& p1, const JacL& p2) {
    F29 u1, u2, s1, s2, z1c, z2c;
    if (Z2ONE) {
        u1 = p1.X;
        s1 = p1.Y;
    } else {
        const F29 zz = f29_sqr_call(p2.Z);
        z2c = f29_mul_call(zz, p2.Z);
        u1 = f29_mul_call(p1.X, zz);
        s1 = f29_mul_call(p1.Y, z2c);
    }
    if (Z1ONE) {
        u2 = p2.X;
        s2 = p2.Y;
    } else {
        const F29 zz = f29_sqr_call(p1.Z);
        z1c = f29_mul_call(zz, p1.Z);
        u2 = f29_mul_call(p2.X, zz);
        s2 = f29_mul_call(p2.Y, z1c);
    }
    const F29 h = f29_norm(f29_sub<1>(u2, u1));
    const F29 r = f29_norm(f29_sub<1>(s2, s1));
    const F29 h2 = f29_sqr_call(h);
    const F29 h3 = f29_mul_call(h2, h);
    const F29 v = f29_mul_call(u1, h2);
    JacWL o;
    o.p.X = f29_norm(f29_sub<2>(f29_sub<1>(f29_sqr_call(r), h3), f29_times<2>(v)));
    o.p.Y = f29_norm(f29_sub<1>(f29_mul_call(r, f29_sub<1>(v, o.p.X)), f29_mul_call(s1, h3)));
    if (Z1ONE && Z2ONE) {
        o.p.Z = h;
        o.W = f29_small(1);
    } else if (Z1ONE) {
        o.p.Z = f29_mul_call(p2.Z, h);
        o.W = z2c;
    } else if (Z2ONE) {
        o.p.Z = f29_mul_call(p1.Z, h);
        o.W = z1c;
    } else {
        o.p.Z = f29_mul_call(f29_mul_call(p1.Z, p2.Z), h);
        o.W = f29_mul_call(z1c, z2c);
    }
    return o;
}
// jac_dbl_cv<Secp256k1> (ec.hpp; a = 0)
P2E_HD JacWL jac_dbl29(const JacL& p) {
    const F29 a = f29_sqr_call(p.X);
    const F29 b = f29_sqr_call(p.Y);
    const F29 c = f29_sqr_call(b);
    const F29 t = f29_norm(f29_sub<1>(f29_sub<1>(f29_sqr_call(f29_add(p.X, b)), a), c));
    const F29 d = f29_times<2>(t);
    const F29 e = f29_norm(f29_times<3>(a));
    const F29 f = f29_sqr_call(e);
    JacWL o;
    o.p.X = f29_norm(f29_sub<4>(f, f29_times<2>(d)));
    const F29 c8 = f29_times<2>(f29_norm(f29_times<4>(c)));
    o.p.Y = f29_norm(f29_sub<2>(f29_mul_call(e, f29_sub<1>(d, o.p.X)), c8));
    o.p.Z = f29_norm(f29_times<2>(f29_mul_call(p.Y, p.Z)));
    o.W = f29_sqr_call(f29_sqr_call(p.Z));
    return o;
}

}  // namespace p2e
