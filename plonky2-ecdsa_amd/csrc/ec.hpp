// Inversion-free curve arithmetic used to pre-compute every intermediate point of the reference's
// (affine, one-inversion-per-op) gadget schedule.
//
// The reference's curve_add / curve_double (gadgets/curve.rs:160-223) each need v^-1 with
// v = x2 - x1  (or 2y).  Doing 311 dependent Fermat inversions per signature would make the path
// ALU-bound by two orders of magnitude.  Instead the chain is walked once in Jacobian coordinates
// (x = X/Z^2, y = Y/Z^3) and for every op we also keep W such that   v^-1 = W / Z3 :
//     add:    Z3 = Z1*Z2*H,  H = X2*Z1^2 - X1*Z2^2 = v*(Z1*Z2)^2   =>  v^-1 = (Z1*Z2)^3 / Z3
//     double: Z3 = 2*Y1*Z1,  2y = 2*Y1/Z1^3                         =>  (2y)^-1 = Z1^4 / Z3
// One Montgomery batch inversion over all Z3 of a signature then yields every affine point AND every
// v^-1 the witness generators need.  The formulas are exact algebraic images of the affine ones (no
// use of the curve equation), so they agree with the reference even for off-curve inputs; the only
// failure mode, Z3 == 0, is exactly the reference's inverse-of-zero panic.
#pragma once
#include "fe.hpp"

namespace p2e {

struct Aff {
    U256 x, y;
};
struct Jac {
    U256 X, Y, Z;
};
struct JacW {
    Jac p;
    U256 W;
};

// The crate's two curves (curve/secp256k1.rs:15-38, curve/p256.rs:15-57): base field, scalar field, a and b.
// Only a = 0 and a = -3 occur; jac_dbl_cv uses exactly those two shapes.
struct Secp256k1 {
    typedef ModP Fp;
    typedef ModN Fn;
    static constexpr bool kAZero = true;
    P2E_HD static U256 a() { return u256_zero(); }
    P2E_HD static U256 b() { return u256_small(7); }
};
struct P256 {
    typedef ModP256 Fp;
    typedef ModN256 Fn;
    static constexpr bool kAZero = false;   // a = -3 = p - 3
    P2E_HD static U256 a() {
        U256 r;
        P2E_UNROLL
        for (int i = 0; i < 8; i++) r.w[i] = ModP256::m(i);
        r.w[0] -= 3u;   // low word of p is 0xFFFFFFFF: no borrow
        return r;
    }
    P2E_HD static U256 b() {   // 5AC635D8 AA3A93E7 B3EBBD55 769886BC 651D06B0 CC53B0F6 3BCE3C3E 27D2604B
        const u32 v[8] = {0x27D2604Bu, 0x3BCE3C3Eu, 0xCC53B0F6u, 0x651D06B0u, 0x769886BCu, 0xB3EBBD55u, 0xAA3A93E7u, 0x5AC635D8u};
        U256 r;
        P2E_UNROLL
        for (int i = 0; i < 8; i++) r.w[i] = v[i];
        return r;
    }
};

typedef ModP Fp;

P2E_HD U256 fp_mul(const U256& a, const U256& b) { return fe_mul<Fp>(a, b); }
P2E_HD U256 fp_sqr(const U256& a) { return fe_sqr<Fp>(a); }
P2E_HD U256 fp_add(const U256& a, const U256& b) { return fe_add<Fp>(a, b); }
P2E_HD U256 fp_sub(const U256& a, const U256& b) { return fe_sub<Fp>(a, b); }

// P3 = P1 + P2 (P1 != +-P2), with compile-time knowledge of Z1 == 1 / Z2 == 1
template <class CV, bool Z1ONE, bool Z2ONE>
P2E_HD JacW jac_add_cv(const Jac& p1, const Jac& p2) {
    typedef typename CV::Fp F;
    U256 u1, u2, s1, s2, z1c, z2c;
    if (Z2ONE) {
        u1 = p1.X;
        s1 = p1.Y;
    } else {
        U256 zz = fe_sqr<F>(p2.Z);
        z2c = fe_mul<F>(zz, p2.Z);
        u1 = fe_mul<F>(p1.X, zz);
        s1 = fe_mul<F>(p1.Y, z2c);
    }
    if (Z1ONE) {
        u2 = p2.X;
        s2 = p2.Y;
    } else {
        U256 zz = fe_sqr<F>(p1.Z);
        z1c = fe_mul<F>(zz, p1.Z);
        u2 = fe_mul<F>(p2.X, zz);
        s2 = fe_mul<F>(p2.Y, z1c);
    }
    U256 h = fe_sub<F>(u2, u1);
    U256 r = fe_sub<F>(s2, s1);
    U256 h2 = fe_sqr<F>(h);
    U256 h3 = fe_mul<F>(h2, h);
    U256 v = fe_mul<F>(u1, h2);
    JacW o;
    o.p.X = fe_sub<F>(fe_sub<F>(fe_sqr<F>(r), h3), fe_add<F>(v, v));
    o.p.Y = fe_sub<F>(fe_mul<F>(r, fe_sub<F>(v, o.p.X)), fe_mul<F>(s1, h3));
    if (Z1ONE && Z2ONE) {
        o.p.Z = h;
        o.W = u256_small(1);
    } else if (Z1ONE) {
        o.p.Z = fe_mul<F>(p2.Z, h);
        o.W = z2c;
    } else if (Z2ONE) {
        o.p.Z = fe_mul<F>(p1.Z, h);
        o.W = z1c;
    } else {
        o.p.Z = fe_mul<F>(fe_mul<F>(p1.Z, p2.Z), h);
        o.W = fe_mul<F>(z1c, z2c);
    }
    return o;
}
template <bool Z1ONE, bool Z2ONE>
P2E_HD JacW jac_add(const Jac& p1, const Jac& p2) {
    return jac_add_cv<Secp256k1, Z1ONE, Z2ONE>(p1, p2);
}

// P3 = 2*P1: the slope numerator is 3*X^2 + a*Z^4 = 3*X^2 (a = 0) or 3*(X^2 - Z^4) (a = -3); Z^4 is W anyway
template <class CV>
P2E_HD JacW jac_dbl_cv(const Jac& p) {
    typedef typename CV::Fp F;
    U256 a = fe_sqr<F>(p.X);
    U256 b = fe_sqr<F>(p.Y);
    U256 c = fe_sqr<F>(b);
    U256 t = fe_sub<F>(fe_sub<F>(fe_sqr<F>(fe_add<F>(p.X, b)), a), c);
    U256 d = fe_add<F>(t, t);
    JacW o;
    U256 e1 = a;
    if (!CV::kAZero) {
        o.W = fe_sqr<F>(fe_sqr<F>(p.Z));
        e1 = fe_sub<F>(a, o.W);
    }
    U256 e = fe_add<F>(fe_add<F>(e1, e1), e1);
    U256 f = fe_sqr<F>(e);
    o.p.X = fe_sub<F>(f, fe_add<F>(d, d));
    U256 c2 = fe_add<F>(c, c);
    U256 c4 = fe_add<F>(c2, c2);
    U256 c8 = fe_add<F>(c4, c4);
    o.p.Y = fe_sub<F>(fe_mul<F>(e, fe_sub<F>(d, o.p.X)), c8);
    U256 yz = fe_mul<F>(p.Y, p.Z);
    o.p.Z = fe_add<F>(yz, yz);
    if (CV::kAZero) {   // (same statement order as the a = 0 form the tuned kernels were built from)
        U256 zz = fe_sqr<F>(p.Z);
        o.W = fe_sqr<F>(zz);
    }
    return o;
}
P2E_HD JacW jac_dbl(const Jac& p) { return jac_dbl_cv<Secp256k1>(p); }

P2E_HD Jac jac_from_aff(const Aff& a) {
    Jac j;
    j.X = a.x;
    j.Y = a.y;
    j.Z = u256_small(1);
    return j;
}
// affine helpers for the HOST only (constant tables at context / program creation, synthetic inputs)
template <class CV>
P2E_HD Aff aff_add_cv(const Aff& p, const Aff& q) {
    typedef typename CV::Fp F;
    U256 l = fe_mul<F>(fe_sub<F>(q.y, p.y), fe_inv<F>(fe_sub<F>(q.x, p.x)));
    Aff r;
    r.x = fe_sub<F>(fe_sub<F>(fe_sqr<F>(l), p.x), q.x);
    r.y = fe_sub<F>(fe_mul<F>(l, fe_sub<F>(p.x, r.x)), p.y);
    return r;
}
template <class CV>
P2E_HD Aff aff_dbl_cv(const Aff& p) {
    typedef typename CV::Fp F;
    U256 xx = fe_sqr<F>(p.x);
    U256 num = fe_add<F>(fe_add<F>(fe_add<F>(xx, xx), xx), CV::a());
    U256 l = fe_mul<F>(num, fe_inv<F>(fe_add<F>(p.y, p.y)));
    Aff r;
    r.x = fe_sub<F>(fe_sub<F>(fe_sqr<F>(l), p.x), p.x);
    r.y = fe_sub<F>(fe_mul<F>(l, fe_sub<F>(p.x, r.x)), p.y);
    return r;
}
template <class CV>
P2E_HD Aff aff_neg_cv(const Aff& p) {
    Aff r;
    r.x = p.x;
    r.y = fe_neg<typename CV::Fp>(p.y);
    return r;
}
P2E_HD Aff aff_add(const Aff& p, const Aff& q) {
    U256 l = fp_mul(fp_sub(q.y, p.y), fe_inv_p(fp_sub(q.x, p.x)));
    Aff r;
    r.x = fp_sub(fp_sub(fp_sqr(l), p.x), q.x);
    r.y = fp_sub(fp_mul(l, fp_sub(p.x, r.x)), p.y);
    return r;
}
P2E_HD Aff aff_dbl(const Aff& p) {
    U256 xx = fp_sqr(p.x);
    U256 l = fp_mul(fp_add(fp_add(xx, xx), xx), fe_inv_p(fp_add(p.y, p.y)));
    Aff r;
    r.x = fp_sub(fp_sub(fp_sqr(l), p.x), p.x);
    r.y = fp_sub(fp_mul(l, fp_sub(p.x, r.x)), p.y);
    return r;
}
P2E_HD Aff aff_neg(const Aff& p) {
    Aff r;
    r.x = p.x;
    r.y = fe_neg<Fp>(p.y);
    return r;
}

}  // namespace p2e
