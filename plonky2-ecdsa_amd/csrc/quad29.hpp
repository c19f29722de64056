// The curve operations of quad.hpp (four lanes per signature, secp256k1) on fe29.hpp's lazy 29-bit limbs.
//
// Same levels, same products, same W / prefix-product conventions as jac_dbl_quad / jac_add_quad (quad.hpp) and so as
// ec.hpp: what reaches scratch memory is the canonical form of the same field elements, bit for bit.  What changes is
// the time between the multiplications: additions are nine independent adds, subtractions add a multiple of p whose
// limbs cover the subtrahend's, and a value is only brought back to tight limbs (f29_norm) where the next
// multiplication's 64-bit columns need it -- six times in a doubling, four in an addition.  Measured on one MI355X
// wave (tools/ubench/fe29_latency.hip, profiles/r03_fe29_latency.txt): multiplication 648 -> 348 ns, squaring
// 548 -> 278 ns, the arithmetic of one lane's doubling 3 593 -> 1 427 ns.
//
// Stores: an op hands scratch memory X3, Y3 (unless F_NO_AFFINE), Z3, W and the prefix product BEFORE the op, all
// canonical.  X3, the prefix, Z3 and W exist before the op's last level, so the four lanes canonicalise one of them
// each there (quad_store_form: role 0 X3, 1 prefix, 2 Z3, 3 W) -- one conversion's latency for four values -- and the
// lane that holds Z3 tells the others whether it is zero (the prefix product must skip a zero, and the reference's
// inverse() panics on it: gadgets/nonnative.rs:863).  Y3 only exists after the last level; the few ops that need it
// convert it there.
#pragma once
#include "pipeline.hpp"

namespace p2e {

// quad_level (quad.hpp) on lazy limbs.  Every pair must satisfy f29_mul's bound on its own: the emulation build
// multiplies all four and checks each.
template <int USED, bool SQR>
P2E_HD void quad_level29(int role, const F29& a0, const F29& b0, const F29& a1, const F29& b1, const F29& a2, const F29& b2,
                         const F29& a3, const F29& b3, F29& r0, F29& r1, F29& r2, F29& r3) {
#if defined(__HIP_DEVICE_COMPILE__)
    F29 A = a0, B = b0;
    if (USED > 1) {
        A = f29_select(role == 1, a1, A);
        if (!SQR) B = f29_select(role == 1, b1, B);
    }
    if (USED > 2) {
        A = f29_select(role == 2, a2, A);
        if (!SQR) B = f29_select(role == 2, b2, B);
    }
    if (USED > 3) {
        A = f29_select(role == 3, a3, A);
        if (!SQR) B = f29_select(role == 3, b3, B);
    }
    const F29 r = SQR ? f29_sqr(A) : f29_mul(A, B);
    r0 = f29_bcast<0>(r);
    if (USED > 1) r1 = f29_bcast<1>(r);
    if (USED > 2) r2 = f29_bcast<2>(r);
    if (USED > 3) r3 = f29_bcast<3>(r);
#else
    (void)role;
    r0 = SQR ? f29_sqr(a0) : f29_mul(a0, b0);
    if (USED > 1) r1 = SQR ? f29_sqr(a1) : f29_mul(a1, b1);
    if (USED > 2) r2 = SQR ? f29_sqr(a2) : f29_mul(a2, b2);
    if (USED > 3) r3 = SQR ? f29_sqr(a3) : f29_mul(a3, b3);
#endif
}

struct QuadRes29 {
    JacL p;       // the result, tight limbs
    F29 zz3;      // Z3^2
    F29 zz1;      // Z1^2 of the first operand
    F29 acc;      // prefix product through this op (zeros replaced by one)
    U256 mine;    // canonical form of what THIS lane stores: role 0 X3 (Z3 again for F_NO_AFFINE), 1 the prefix product
                  // before the op, 2 Z3, 3 W
    bool z3_zero;
};

// the canonicalisation shared by the four lanes; must run before the level that multiplies the prefix product
P2E_HD void quad_store_form(int role, bool no_affine, const F29& x3, const F29& acc_before, const F29& z3, const F29& w, U256& mine,
                            bool& z3_zero) {
    const F29 v = f29_select(role == 1, acc_before, f29_select(role == 3, w, f29_select(role == 0 && !no_affine, x3, z3)));
    mine = f29_canon(v);
#if defined(__HIP_DEVICE_COMPILE__)
    const int zf = u256_is_zero(mine) ? 1 : 0;   // (means something in the lane of role 2)
    z3_zero = __builtin_amdgcn_mov_dpp(zf, 2 * 0x55 /* quad_perm:[2,2,2,2] */, 0xF, 0xF, true) != 0;
#else
    z3_zero = u256_is_zero(f29_canon(z3));
#endif
}

// jac_dbl_quad on lazy limbs (a = 0; ec.hpp jac_dbl, reference curve/curve_types.rs:192-230 in Jacobian form)
P2E_HD QuadRes29 jac_dbl_quad29(int role, bool no_affine, const JacL& p, const F29& acc) {
    F29 a, b, yz, zz, c, t0, f, w, y0, d0;
    quad_level29<4, false>(role, p.X, p.X, p.Y, p.Y, p.Y, p.Z, p.Z, p.Z, a, b, yz, zz);
    const F29 z3 = f29_norm(f29_times<2>(yz));
    const F29 e = f29_norm(f29_times<3>(a));
    quad_level29<4, true>(role, b, b, f29_add(p.X, b), b, e, b, zz, b, c, t0, f, w);
    const F29 t = f29_norm(f29_sub<1>(f29_sub<1>(t0, a), c));
    const F29 d = f29_times<2>(t);
    QuadRes29 o;
    o.p.X = f29_norm(f29_sub<4>(f, f29_times<2>(d)));
    quad_store_form(role, no_affine, o.p.X, acc, z3, w, o.mine, o.z3_zero);
    const F29 zfix = f29_select(o.z3_zero, f29_small(1), z3);
    quad_level29<3, false>(role, e, f29_sub<1>(d, o.p.X), acc, zfix, z3, z3, z3, z3, y0, o.acc, o.zz3, d0);
    const F29 c8 = f29_times<2>(f29_norm(f29_times<4>(c)));
    o.p.Y = f29_norm(f29_sub<2>(y0, c8));
    o.p.Z = z3;
    o.zz1 = zz;
    return o;
}

// jac_add_quad<Z1ONE, Z2ONE> on lazy limbs (ec.hpp jac_add; the sum is always the general-case formula: quirk Q7)
template <bool Z1ONE, bool Z2ONE>
P2E_HD QuadRes29 jac_add_quad29(int role, bool no_affine, const JacL& p1, bool have_zz1, const F29& zz1_in, const JacL& p2,
                                const F29& acc) {
    F29 zz1 = zz1_in, zz2, z12, z1c, z2c, u1 = p1.X, u2 = p2.X, s1 = p1.Y, s2 = p2.Y, d0, d1;
    // level 1: squares of the Z's
    if (!Z1ONE && !Z2ONE) {
        quad_level29<3, false>(role, p1.Z, p1.Z, p2.Z, p2.Z, p1.Z, p2.Z, p1.Z, p1.Z, zz1, zz2, z12, d0);
    } else if (!Z1ONE) {
        if (!have_zz1) quad_level29<1, true>(role, p1.Z, p1.Z, p1.Z, p1.Z, p1.Z, p1.Z, p1.Z, p1.Z, zz1, d0, d1, d0);
    } else if (!Z2ONE) {
        quad_level29<1, true>(role, p2.Z, p2.Z, p2.Z, p2.Z, p2.Z, p2.Z, p2.Z, p2.Z, zz2, d0, d1, d0);
    }
    // level 2: cubes and the cross products of the x coordinates
    if (!Z1ONE && !Z2ONE) {
        quad_level29<4, false>(role, zz1, p1.Z, zz2, p2.Z, p1.X, zz2, p2.X, zz1, z1c, z2c, u1, u2);
    } else if (!Z1ONE) {
        quad_level29<2, false>(role, zz1, p1.Z, p2.X, zz1, zz1, zz1, zz1, zz1, z1c, u2, d0, d1);
    } else if (!Z2ONE) {
        quad_level29<2, false>(role, zz2, p2.Z, p1.X, zz2, zz2, zz2, zz2, zz2, z2c, u1, d0, d1);
    }
    const F29 h = f29_norm(f29_sub<1>(u2, u1));
    // level 3
    F29 h2, z3;
    if (!Z1ONE && !Z2ONE) {
        quad_level29<4, false>(role, p1.Y, z2c, p2.Y, z1c, h, h, z12, h, s1, s2, h2, z3);
    } else if (!Z1ONE) {
        quad_level29<3, false>(role, p2.Y, z1c, h, h, p1.Z, h, h, h, s2, h2, z3, d0);
    } else if (!Z2ONE) {
        quad_level29<3, false>(role, p1.Y, z2c, h, h, p2.Z, h, h, h, s1, h2, z3, d0);
    } else {
        quad_level29<1, true>(role, h, h, h, h, h, h, h, h, h2, d0, d1, d0);
        z3 = h;
    }
    const F29 r = f29_norm(f29_sub<1>(s2, s1));
    // level 4
    F29 h3, v, r2, w;
    if (!Z1ONE && !Z2ONE) {
        quad_level29<4, false>(role, h2, h, u1, h2, r, r, z1c, z2c, h3, v, r2, w);
    } else {
        quad_level29<3, false>(role, h2, h, u1, h2, r, r, r, r, h3, v, r2, d0);
        w = Z1ONE ? (Z2ONE ? f29_small(1) : z2c) : z1c;
    }
    QuadRes29 o;
    o.p.X = f29_norm(f29_sub<2>(f29_sub<1>(r2, h3), f29_times<2>(v)));
    quad_store_form(role, no_affine, o.p.X, acc, z3, w, o.mine, o.z3_zero);
    const F29 zfix = f29_select(o.z3_zero, f29_small(1), z3);
    // level 5
    F29 t1, t2;
    quad_level29<4, false>(role, r, f29_sub<1>(v, o.p.X), s1, h3, acc, zfix, z3, z3, t1, t2, o.acc, o.zz3);
    o.p.Y = f29_norm(f29_sub<1>(t1, t2));
    o.p.Z = z3;
    o.zz1 = zz1;
    return o;
}

}  // namespace p2e
