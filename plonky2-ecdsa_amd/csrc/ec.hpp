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

typedef ModP Fp;

P2E_HD U256 fp_mul(const U256& a, const U256& b) { return fe_mul<Fp>(a, b); }
P2E_HD U256 fp_sqr(const U256& a) { return fe_sqr<Fp>(a); }
P2E_HD U256 fp_add(const U256& a, const U256& b) { return fe_add<Fp>(a, b); }
P2E_HD U256 fp_sub(const U256& a, const U256& b) { return fe_sub<Fp>(a, b); }

// P3 = P1 + P2 (P1 != +-P2), with compile-time knowledge of Z1 == 1 / Z2 == 1
template <bool Z1ONE, bool Z2ONE>
P2E_HD JacW jac_add(const Jac& p1, const Jac& p2) {
    U256 u1, u2, s1, s2, z1c, z2c;
    if (Z2ONE) {
        u1 = p1.X;
        s1 = p1.Y;
    } else {
        U256 zz = fp_sqr(p2.Z);
        z2c = fp_mul(zz, p2.Z);
        u1 = fp_mul(p1.X, zz);
        s1 = fp_mul(p1.Y, z2c);
    }
    if (Z1ONE) {
        u2 = p2.X;
        s2 = p2.Y;
    } else {
        U256 zz = fp_sqr(p1.Z);
        z1c = fp_mul(zz, p1.Z);
        u2 = fp_mul(p2.X, zz);
        s2 = fp_mul(p2.Y, z1c);
    }
    U256 h = fp_sub(u2, u1);
    U256 r = fp_sub(s2, s1);
    U256 h2 = fp_sqr(h);
    U256 h3 = fp_mul(h2, h);
    U256 v = fp_mul(u1, h2);
    JacW o;
    o.p.X = fp_sub(fp_sub(fp_sqr(r), h3), fp_add(v, v));
    o.p.Y = fp_sub(fp_mul(r, fp_sub(v, o.p.X)), fp_mul(s1, h3));
    if (Z1ONE && Z2ONE) {
        o.p.Z = h;
        o.W = u256_small(1);
    } else if (Z1ONE) {
        o.p.Z = fp_mul(p2.Z, h);
        o.W = z2c;
    } else if (Z2ONE) {
        o.p.Z = fp_mul(p1.Z, h);
        o.W = z1c;
    } else {
        o.p.Z = fp_mul(fp_mul(p1.Z, p2.Z), h);
        o.W = fp_mul(z1c, z2c);
    }
    return o;
}

// P3 = 2*P1 (a = 0)
P2E_HD JacW jac_dbl(const Jac& p) {
    U256 a = fp_sqr(p.X);
    U256 b = fp_sqr(p.Y);
    U256 c = fp_sqr(b);
    U256 t = fp_sub(fp_sub(fp_sqr(fp_add(p.X, b)), a), c);
    U256 d = fp_add(t, t);
    U256 e = fp_add(fp_add(a, a), a);
    U256 f = fp_sqr(e);
    JacW o;
    o.p.X = fp_sub(f, fp_add(d, d));
    U256 c2 = fp_add(c, c);
    U256 c4 = fp_add(c2, c2);
    U256 c8 = fp_add(c4, c4);
    o.p.Y = fp_sub(fp_mul(e, fp_sub(d, o.p.X)), c8);
    U256 yz = fp_mul(p.Y, p.Z);
    o.p.Z = fp_add(yz, yz);
    U256 zz = fp_sqr(p.Z);
    o.W = fp_sqr(zz);
    return o;
}

P2E_HD Jac jac_from_aff(const Aff& a) {
    Jac j;
    j.X = a.x;
    j.Y = a.y;
    j.Z = u256_small(1);
    return j;
}
// affine helpers for the HOST only (constant tables at context creation, synthetic inputs)
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
