// Small-batch form of phases A and B: FOUR lanes per signature.
//
// The Jacobian chains are strictly sequential per signature (243 dependent curve ops on the MSM chain) and a lone
// wave needs ~1 750 cycles per field multiplication, so below ~2^15 signatures per call the whole fill is the
// LATENCY of that chain (2^13 signatures: 4.1 ms, of which 1.1 ms is HBM work) while most SIMDs idle -- and that is
// exactly the shape the metric's strong-scaled form lands on (2^16 signatures over 8 GPUs = 8 192 per GPU).
// A curve op is 10-17 field multiplications of depth 3-5: here the four lanes of a quad (lane & 3 = role) own one
// signature, every "level" each lane multiplies ONE operand pair picked by its role and the four products are
// exchanged with v_mov_b32 quad_perm DPP broadcasts (8 per value, no LDS, no barrier).  State is replicated in the
// four lanes, so everything between the multiplications (adds, selects, operand resolution) is unchanged code.
//   doubling            10 multiplications in 3 levels        (jac_dbl_quad)
//   mixed addition      12 multiplications in 4 levels        (jac_add_quad<false, true>, Z1^2 carried from the op before)
//   general addition    18 multiplications in 5 levels
// The formulas are those of ec.hpp (same W with v^-1 = W / Z3, same prefix products): phases B and C cannot tell
// which form of phase A ran, and tests compare both against the oracle.
// Phase B splits a piece's backward pass into S sub-ranges walked by S lanes (one inversion each, all S in flight at
// once): body_batch_inv_split.
// Round 3: the two hot loops (285 of the 311 ops) and phase B compute on lazy 29-bit limbs (quad29.hpp, fe29.hpp: a
// multiplication costs a lone wave 350 instead of 650 ns, an addition 74 instead of 162); the generic walker below and
// the curve programs' P-256 form keep the canonical words of fe.hpp.
#pragma once
#include "pipeline.hpp"
#include "quad29.hpp"

namespace p2e {

#if defined(__HIP_DEVICE_COMPILE__)
template <int J>
__device__ __forceinline__ U256 quad_bcast(const U256& v) {
    U256 r;
    P2E_UNROLL
    for (int k = 0; k < 8; k++)
        r.w[k] = (u32)__builtin_amdgcn_mov_dpp((int)v.w[k], J * 0x55 /* quad_perm:[J,J,J,J] */, 0xF, 0xF, true);
    return r;
}
#endif

// One level: products (a0*b0, a1*b1, a2*b2, a3*b3), the first USED of them wanted.  Device: this lane multiplies the
// pair of its role, then the quad exchanges; host (emulation): all of them, the role is irrelevant.  SQR: every
// pair is a square (b ignored).
template <int USED, bool SQR, class CV = Secp256k1>
P2E_HD void quad_level(int role, const U256& a0, const U256& b0, const U256& a1, const U256& b1, const U256& a2,
                       const U256& b2, const U256& a3, const U256& b3, U256& r0, U256& r1, U256& r2, U256& r3) {
#if defined(__HIP_DEVICE_COMPILE__)
    U256 A = a0, B = b0;
    if (USED > 1) {
        A = u256_select(role == 1, a1, A);
        if (!SQR) B = u256_select(role == 1, b1, B);
    }
    if (USED > 2) {
        A = u256_select(role == 2, a2, A);
        if (!SQR) B = u256_select(role == 2, b2, B);
    }
    if (USED > 3) {
        A = u256_select(role == 3, a3, A);
        if (!SQR) B = u256_select(role == 3, b3, B);
    }
    const U256 r = SQR ? fe_sqr<typename CV::Fp>(A) : fe_mul<typename CV::Fp>(A, B);
    r0 = quad_bcast<0>(r);
    if (USED > 1) r1 = quad_bcast<1>(r);
    if (USED > 2) r2 = quad_bcast<2>(r);
    if (USED > 3) r3 = quad_bcast<3>(r);
#else
    (void)role;
    typedef typename CV::Fp F_;
    r0 = SQR ? fe_sqr<F_>(a0) : fe_mul<F_>(a0, b0);
    if (USED > 1) r1 = SQR ? fe_sqr<F_>(a1) : fe_mul<F_>(a1, b1);
    if (USED > 2) r2 = SQR ? fe_sqr<F_>(a2) : fe_mul<F_>(a2, b2);
    if (USED > 3) r3 = SQR ? fe_sqr<F_>(a3) : fe_mul<F_>(a3, b3);
#endif
}

struct QuadRes {
    JacW res;
    U256 zz3;   // Z3^2: the first thing an addition that consumes this result would have to compute
    U256 zz1;   // Z1^2 of the first operand (it may be the next op's first operand again: a conditional add that adds nothing)
    U256 acc;   // running product of the Z3 values (zeros replaced by one)
    bool z3_zero;
};

// ec.hpp jac_dbl in three levels; acc * Z3 and Z3^2 ride in the free slots of the last one
P2E_HD QuadRes jac_dbl_quad(int role, const Jac& p, const U256& acc) {
    U256 a, b, yz, zz, c, t0, f, w, y0, d0;
    quad_level<4, false>(role, p.X, p.X, p.Y, p.Y, p.Y, p.Z, p.Z, p.Z, a, b, yz, zz);
    const U256 z3 = fp_add(yz, yz);
    const U256 e = fp_add(fp_add(a, a), a);
    quad_level<4, true>(role, b, b, fp_add(p.X, b), b, e, b, zz, b, c, t0, f, w);
    const U256 t = fp_sub(fp_sub(t0, a), c);
    const U256 d = fp_add(t, t);
    QuadRes o;
    o.res.p.X = fp_sub(f, fp_add(d, d));
    const U256 c2 = fp_add(c, c), c4 = fp_add(c2, c2), c8 = fp_add(c4, c4);
    o.z3_zero = u256_is_zero(z3);
    const U256 zfix = u256_select(o.z3_zero, u256_small(1), z3);
    quad_level<3, false>(role, e, fp_sub(d, o.res.p.X), acc, zfix, z3, z3, z3, z3, y0, o.acc, o.zz3, d0);
    o.res.p.Y = fp_sub(y0, c8);
    o.res.p.Z = z3;
    o.res.W = w;
    o.zz1 = zz;
    return o;
}

// ec.hpp jac_add<Z1ONE, Z2ONE> in at most five levels.  have_zz1: zz1_in = Z1^2 is already known (carried from the
// op that produced p1), which spares the mixed addition its first level.
template <bool Z1ONE, bool Z2ONE>
P2E_HD QuadRes jac_add_quad(int role, const Jac& p1, bool have_zz1, const U256& zz1_in, const Jac& p2, const U256& acc) {
    U256 zz1 = zz1_in, zz2, z12, z1c, z2c, u1 = p1.X, u2 = p2.X, s1 = p1.Y, s2 = p2.Y, d0, d1;
    // level 1: squares of the Z's
    if (!Z1ONE && !Z2ONE) {
        quad_level<3, false>(role, p1.Z, p1.Z, p2.Z, p2.Z, p1.Z, p2.Z, p1.Z, p1.Z, zz1, zz2, z12, d0);
    } else if (!Z1ONE) {
        if (!have_zz1) quad_level<1, true>(role, p1.Z, p1.Z, p1.Z, p1.Z, p1.Z, p1.Z, p1.Z, p1.Z, zz1, d0, d1, d0);
    } else if (!Z2ONE) {
        quad_level<1, true>(role, p2.Z, p2.Z, p2.Z, p2.Z, p2.Z, p2.Z, p2.Z, p2.Z, zz2, d0, d1, d0);
    }
    // level 2: cubes and the cross products of the x coordinates
    if (!Z1ONE && !Z2ONE) {
        quad_level<4, false>(role, zz1, p1.Z, zz2, p2.Z, p1.X, zz2, p2.X, zz1, z1c, z2c, u1, u2);
    } else if (!Z1ONE) {
        quad_level<2, false>(role, zz1, p1.Z, p2.X, zz1, zz1, zz1, zz1, zz1, z1c, u2, d0, d1);
    } else if (!Z2ONE) {
        quad_level<2, false>(role, zz2, p2.Z, p1.X, zz2, zz2, zz2, zz2, zz2, z2c, u1, d0, d1);
    }
    const U256 h = fp_sub(u2, u1);
    // level 3
    U256 h2, z3;
    if (!Z1ONE && !Z2ONE) {
        quad_level<4, false>(role, p1.Y, z2c, p2.Y, z1c, h, h, z12, h, s1, s2, h2, z3);
    } else if (!Z1ONE) {
        quad_level<3, false>(role, p2.Y, z1c, h, h, p1.Z, h, h, h, s2, h2, z3, d0);
    } else if (!Z2ONE) {
        quad_level<3, false>(role, p1.Y, z2c, h, h, p2.Z, h, h, h, s1, h2, z3, d0);
    } else {
        quad_level<1, true>(role, h, h, h, h, h, h, h, h, h2, d0, d1, d0);
        z3 = h;
    }
    const U256 r = fp_sub(s2, s1);
    // level 4
    U256 h3, v, r2;
    QuadRes o;
    if (!Z1ONE && !Z2ONE) {
        quad_level<4, false>(role, h2, h, u1, h2, r, r, z1c, z2c, h3, v, r2, o.res.W);
    } else {
        quad_level<3, false>(role, h2, h, u1, h2, r, r, r, r, h3, v, r2, d0);
        o.res.W = Z1ONE ? (Z2ONE ? u256_small(1) : z2c) : z1c;
    }
    o.res.p.X = fp_sub(fp_sub(r2, h3), fp_add(v, v));
    o.z3_zero = u256_is_zero(z3);
    const U256 zfix = u256_select(o.z3_zero, u256_small(1), z3);
    // level 5
    U256 t1, t2;
    quad_level<4, false>(role, r, fp_sub(v, o.res.p.X), s1, h3, acc, zfix, z3, z3, t1, t2, o.acc, o.zz3);
    o.res.p.Y = fp_sub(t1, t2);
    o.res.p.Z = z3;
    o.zz1 = zz1;
    return o;
}

// ---- the same levels over either curve of the crate (curve programs, curves.hpp) ------------------------------------
// a = 0: jac_dbl_quad above.  a = -3 (P-256): the slope numerator 3 (X^2 - Z^4) needs Z^4 = W first, so the doubling has
// four levels (ec.hpp jac_dbl_cv, same Z3 = 2 Y Z and W = Z^4):
//   1: X^2, Y^2, Y Z, Z^2      2: (Y^2)^2, (X + Y^2)^2, (Z^2)^2      3: e^2 [, acc * Z3, Z3^2]      4: e (d - X3)
template <class CV>
P2E_HD QuadRes jac_dbl_quad_cv(int role, const Jac& p, const U256& acc) {
    typedef typename CV::Fp F;
    if (CV::kAZero) return jac_dbl_quad(role, p, acc);
    U256 a, b, yz, zz, c, t0, w, d0, f, y0;
    quad_level<4, false, CV>(role, p.X, p.X, p.Y, p.Y, p.Y, p.Z, p.Z, p.Z, a, b, yz, zz);
    const U256 z3 = fe_add<F>(yz, yz);
    quad_level<3, true, CV>(role, b, b, fe_add<F>(p.X, b), b, zz, b, zz, b, c, t0, w, d0);
    const U256 t = fe_sub<F>(fe_sub<F>(t0, a), c);
    const U256 d = fe_add<F>(t, t);
    const U256 e1 = fe_sub<F>(a, w);
    const U256 e = fe_add<F>(fe_add<F>(e1, e1), e1);
    QuadRes o;
    o.z3_zero = u256_is_zero(z3);
    const U256 zfix = u256_select(o.z3_zero, u256_small(1), z3);
    quad_level<3, false, CV>(role, e, e, acc, zfix, z3, z3, z3, z3, f, o.acc, o.zz3, d0);
    o.res.p.X = fe_sub<F>(f, fe_add<F>(d, d));
    const U256 c2 = fe_add<F>(c, c), c4 = fe_add<F>(c2, c2), c8 = fe_add<F>(c4, c4);
    quad_level<1, false, CV>(role, e, fe_sub<F>(d, o.res.p.X), e, e, e, e, e, e, y0, d0, d0, d0);
    o.res.p.Y = fe_sub<F>(y0, c8);
    o.res.p.Z = z3;
    o.res.W = w;
    o.zz1 = zz;
    return o;
}
// jac_add_quad over CV::Fp (the formulas do not involve the curve's coefficients)
template <class CV, bool Z1ONE, bool Z2ONE>
P2E_HD QuadRes jac_add_quad_cv(int role, const Jac& p1, bool have_zz1, const U256& zz1_in, const Jac& p2, const U256& acc) {
    typedef typename CV::Fp F;
    if (CV::kAZero) return jac_add_quad<Z1ONE, Z2ONE>(role, p1, have_zz1, zz1_in, p2, acc);
    U256 zz1 = zz1_in, zz2, z12, z1c, z2c, u1 = p1.X, u2 = p2.X, s1 = p1.Y, s2 = p2.Y, d0, d1;
    if (!Z1ONE && !Z2ONE) {
        quad_level<3, false, CV>(role, p1.Z, p1.Z, p2.Z, p2.Z, p1.Z, p2.Z, p1.Z, p1.Z, zz1, zz2, z12, d0);
    } else if (!Z1ONE) {
        if (!have_zz1) quad_level<1, true, CV>(role, p1.Z, p1.Z, p1.Z, p1.Z, p1.Z, p1.Z, p1.Z, p1.Z, zz1, d0, d1, d0);
    } else if (!Z2ONE) {
        quad_level<1, true, CV>(role, p2.Z, p2.Z, p2.Z, p2.Z, p2.Z, p2.Z, p2.Z, p2.Z, zz2, d0, d1, d0);
    }
    if (!Z1ONE && !Z2ONE) {
        quad_level<4, false, CV>(role, zz1, p1.Z, zz2, p2.Z, p1.X, zz2, p2.X, zz1, z1c, z2c, u1, u2);
    } else if (!Z1ONE) {
        quad_level<2, false, CV>(role, zz1, p1.Z, p2.X, zz1, zz1, zz1, zz1, zz1, z1c, u2, d0, d1);
    } else if (!Z2ONE) {
        quad_level<2, false, CV>(role, zz2, p2.Z, p1.X, zz2, zz2, zz2, zz2, zz2, z2c, u1, d0, d1);
    }
    const U256 h = fe_sub<F>(u2, u1);
    U256 h2, z3;
    if (!Z1ONE && !Z2ONE) {
        quad_level<4, false, CV>(role, p1.Y, z2c, p2.Y, z1c, h, h, z12, h, s1, s2, h2, z3);
    } else if (!Z1ONE) {
        quad_level<3, false, CV>(role, p2.Y, z1c, h, h, p1.Z, h, h, h, s2, h2, z3, d0);
    } else if (!Z2ONE) {
        quad_level<3, false, CV>(role, p1.Y, z2c, h, h, p2.Z, h, h, h, s1, h2, z3, d0);
    } else {
        quad_level<1, true, CV>(role, h, h, h, h, h, h, h, h, h2, d0, d1, d0);
        z3 = h;
    }
    const U256 r = fe_sub<F>(s2, s1);
    U256 h3, v, r2;
    QuadRes o;
    if (!Z1ONE && !Z2ONE) {
        quad_level<4, false, CV>(role, h2, h, u1, h2, r, r, z1c, z2c, h3, v, r2, o.res.W);
    } else {
        quad_level<3, false, CV>(role, h2, h, u1, h2, r, r, r, r, h3, v, r2, d0);
        o.res.W = Z1ONE ? (Z2ONE ? u256_small(1) : z2c) : z1c;
    }
    o.res.p.X = fe_sub<F>(fe_sub<F>(r2, h3), fe_add<F>(v, v));
    o.z3_zero = u256_is_zero(z3);
    const U256 zfix = u256_select(o.z3_zero, u256_small(1), z3);
    U256 t1, t2;
    quad_level<4, false, CV>(role, r, fe_sub<F>(v, o.res.p.X), s1, h3, acc, zfix, z3, z3, t1, t2, o.acc, o.zz3);
    o.res.p.Y = fe_sub<F>(t1, t2);
    o.res.p.Z = z3;
    o.zz1 = zz1;
    return o;
}

struct ChainStateQ {
    Jac out, p1;
    U256 out_zz, p1_zz;     // Z^2 of the two carried points
    uint16_t out_id, p1_id;
    uint16_t dyn_idx, dyn_val;   // the last conditional add's selected source, kept in registers: the op right after it
                                 // names it as its first operand, and reading it back from dyn[] would wait for the store
    U256 acc;
};

// Loaded values that are only consumed after the branches rejoin make the compiler put its s_waitcnt at the JOIN,
// i.e. on every path -- and since vmcnt is in-order that wait also drains the operand fetches issued for the next
// op.  Touching the values inside the (rare) branch that loaded them keeps the wait in there.
P2E_HD void settle(U256& v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v.w[0]), "+v"(v.w[1]), "+v"(v.w[2]), "+v"(v.w[3]), "+v"(v.w[4]), "+v"(v.w[5]), "+v"(v.w[6]), "+v"(v.w[7]));
#else
    (void)v;
#endif
}
P2E_HD void settle(u32& v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v));
#else
    (void)v;
#endif
}

// Scratch outputs of one op, shared by the four lanes: role 0 stores X, 1 Y, 2 Z, 3 W; the prefix product by all four
// (same value, same address).  BRANCH-FREE on purpose: with the stores under "if (role == k)" every path through the
// op has a different number of stores, the compiler's s_waitcnt bookkeeping assumes the fewest, and the next
// iteration's first load then waits for (nearly) all of this one's stores to be acknowledged.
P2E_HD void quad_store_op(const Buffers& B, size_t i, int role, int t, uint8_t flags, const QuadRes& q, const U256& acc_before) {
    const size_t o = (size_t)t * B.n + i;
    // results that never need their affine form (F_NO_AFFINE) have no X / Y slot: those lanes repeat Z / W
    const int r = (flags & F_NO_AFFINE) ? (role | 2) : role;
    // (the four bases pinned in scalar registers: a plain select over the kernel-argument fields is turned into an
    // indexed LOAD of the pointer, whose wait again drains the stores in front of it)
    uintptr_t bx = (uintptr_t)B.PX, by = (uintptr_t)B.PY, bz = (uintptr_t)B.PZ, bw = (uintptr_t)B.PW;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(bx), "+s"(by), "+s"(bz), "+s"(bw));
#endif
    U256* const dst = (U256*)(r == 0 ? bx : r == 1 ? by : r == 2 ? bz : bw);
    const U256 val = u256_select(r == 0, q.res.p.X, u256_select(r == 1, q.res.p.Y, u256_select(r == 2, q.res.p.Z, q.res.W)));
    dst[o] = val;
    B.PREF[o] = acc_before;
    if (q.z3_zero) err_or(&B.err[i], ERR_INVERSE_OF_ZERO);   // reference: inverse() of zero panics (gadgets/nonnative.rs:863)
}

// The same for an op computed on lazy limbs (quad29.hpp): every lane stores the ONE value it canonicalised (role 0 X3 --
// Z3 again for F_NO_AFFINE --, 1 the prefix product before the op, 2 Z3, 3 W); Y3 exists only after the op's last level
// and is converted here for the ops that need their affine form (the flag is wave-uniform).
P2E_HD void quad_store_op29(const Buffers& B, size_t i, int role, int t, uint8_t flags, const QuadRes29& q) {
    const size_t o = (size_t)t * B.n + i;
    const bool no_affine = (flags & F_NO_AFFINE) != 0;
    uintptr_t bx = (uintptr_t)B.PX, bp = (uintptr_t)B.PREF, bz = (uintptr_t)B.PZ, bw = (uintptr_t)B.PW;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(bx), "+s"(bp), "+s"(bz), "+s"(bw));
#endif
    U256* const dst = (U256*)(role == 1 ? bp : role == 3 ? bw : (role == 0 && !no_affine) ? bx : bz);
    dst[o] = q.mine;
    if (!no_affine) B.PY[o] = f29_canon(q.p.Y);
    if (q.z3_zero) err_or(&B.err[i], ERR_INVERSE_OF_ZERO);   // reference: inverse() of zero panics (gadgets/nonnative.rs:863)
}

// first operand of op `op`: from the registers of the previous ops whenever the schedule allows (310 of 311 ops),
// from scratch otherwise (the first op of a chain piece)
struct P1Sel {
    uint16_t src1;
    Jac p1;
    bool have_zz1;
    U256 zz1;
};
P2E_HD P1Sel quad_first_operand(const Program& G, const Buffers& B, size_t i, const OpDesc& op, const ChainStateQ& st) {
    P1Sel r;
    u32 src1;
    if (ref_kind(op.ref1) == R_DYN && ref_id(op.ref1) == st.dyn_idx) {
        src1 = st.dyn_val;
    } else {
        src1 = resolve_src(G, B, i, op.ref1);
        settle(src1);
    }
    r.src1 = (uint16_t)src1;
    const bool from_out = r.src1 == st.out_id, from_p1 = r.src1 == st.p1_id;
    Jac p1 = st.out;
    if (!(from_out || from_p1)) {
        p1 = load_jac_src(B, i, r.src1, (op.flags & F_Z1ONE) != 0);
        settle(p1.X);
        settle(p1.Y);
        settle(p1.Z);
    }
    r.p1 = jac_select3(from_out, st.out, from_p1, st.p1, p1);
    r.have_zz1 = from_out || from_p1;
    r.zz1 = u256_select(from_out, st.out_zz, st.p1_zz);
    return r;
}

// body_chain_op (pipeline.hpp) for a quad: same operand resolution, same scratch outputs; the four lanes share the
// stores (quad_store_op).  This generic form serves the ops outside the two hot loops (window-table build, unblinding
// adds, the final add); its operand loads are settled where they are issued.
P2E_HD void body_chain_op_quad(const Program& G, const Buffers& B, size_t i, int role, int t, const OpDesc& op, bool table_affine,
                               const P1Sel& s1, ChainStateQ& st) {
    const uint16_t src1 = s1.src1;
    const Jac& p1 = s1.p1;
    if (role == 0) B.src[(size_t)(2 * t) * B.n + i] = src1;
    QuadRes q;
    if (op.kind == OP_DBL) {
        q = jac_dbl_quad(role, p1, st.acc);
    } else {
        Jac p2;
        u32 digit = 1;
        uint16_t src2;
        bool z2one = (op.flags & F_Z2ONE) != 0;
        {
            if (ref_kind(op.ref2) == R_FBTAB) {
                Aff a = load_fbtab(B, i, ref_id(op.ref2), digit);
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
            settle(p2.X);
            settle(p2.Y);
            settle(p2.Z);
            settle(digit);
        }
        if (role == 0) B.src[(size_t)(2 * t + 1) * B.n + i] = (uint16_t)(src2 | (digit != 0 ? SRC_SEL_BIT : 0));
        if ((op.flags & F_Z1ONE) && z2one)
            q = jac_add_quad<true, true>(role, p1, false, s1.zz1, p2, st.acc);
        else if (z2one)
            q = jac_add_quad<false, true>(role, p1, s1.have_zz1, s1.zz1, p2, st.acc);
        else if (op.flags & F_Z1ONE)
            q = jac_add_quad<true, false>(role, p1, false, s1.zz1, p2, st.acc);
        else
            q = jac_add_quad<false, false>(role, p1, false, s1.zz1, p2, st.acc);
        if (op.kind == OP_CADD) {
            st.dyn_idx = op.cadd_idx;
            st.dyn_val = digit != 0 ? (uint16_t)t : src1;
            if (role == 0) B.dyn[(size_t)op.cadd_idx * B.n + i] = st.dyn_val;
        }
    }
    quad_store_op(B, i, role, t, op.flags, q, st.acc);
    st.acc = q.acc;
    st.p1 = p1;
    st.p1_zz = q.zz1;
    st.p1_id = (op.flags & F_Z1ONE) ? (uint16_t)0xFFFF : src1;   // an affine operand has no Z to carry
    st.out = q.res.p;
    st.out_zz = q.zz3;
    st.out_id = (uint16_t)t;
}
// The two loops that make up 285 of the 311 chain ops, as STRAIGHT-LINE code: every load sits at a fixed place at the
// top of an iteration (window digit / table id two iterations ahead, the table point one iteration ahead), nothing
// loaded is consumed in the iteration that issued it, and there is no branch with a load in it.  (The generic walker
// resolves operands through data-dependent branches; the compiler then drains vmcnt at every join -- on gfx9 that
// also waits for the previous op's scratch stores -- which was 40 % of the chain kernel's cycles.)
//
// MSM loop, iterations [it0, it1) (gadgets/curve_msm.rs:66-73): double, double, conditional add of table[digit].
// `cur` = the running point (first operand of the iteration's first doubling) with its source id.  TA: the window
// table is read in affine form (its inversion batch is done), else in Jacobian form.
template <bool TA>
P2E_HD void body_msm_iters_quad(const Program& G, const Buffers& B, size_t i, int role, int it0, int it1, Jac& cur, uint16_t& cur_id,
                                U256& acc, U256& cur_zz, uint16_t& dyn_idx, uint16_t& dyn_val) {
    const int lb = G.msm_loop_begin;
    // digit index of iteration `it`: MSB first (gadgets/curve_msm.rs:66 .rev()); read off the op table once per call
    auto digit_row = [&](int it) { return (size_t)ref_id(load_op(B.ops, lb + 3 * it + 2).ref2); };
    auto clampi = [&](int it) { return it < it1 ? it : it1 - 1; };
    struct Sel {
        u32 digit;
        u32 src;
    };
    auto sel_of = [&](int it) {
        Sel s;
        const size_t row = digit_row(clampi(it));
        s.digit = B.dig2[row * B.n + i];
        s.src = B.msrc[row * B.n + i];
        return s;
    };
    // table entry `src` (slot id, or constant | DYN_CONST_BIT) -> operand.  Branch-free: the address is selected, the
    // loads are unconditional (a load inside a divergent branch drags its wait to the join)
    auto point_of = [&](u32 src) {
        const bool is_const = (src & DYN_CONST_BIT) != 0;
        const size_t o = (size_t)(src & SRC_ID_MASK) * B.n + i;
        const Aff* cp = &B.cpts[is_const ? (src & SRC_ID_MASK) : 0];
        Jac p;
        if (TA) {
            const U256* px = is_const ? &cp->x : &B.AX[o];
            const U256* py = is_const ? &cp->y : &B.AY[o];
            p.X = *px;
            p.Y = *py;
            p.Z = u256_small(1);
        } else {
            const U256* px = is_const ? &cp->x : &B.PX[o];
            const U256* py = is_const ? &cp->y : &B.PY[o];
            const U256* pz = is_const ? &cp->x : &B.PZ[o];   // (a constant has Z = 1: the loaded word is discarded)
            p.X = *px;
            p.Y = *py;
            p.Z = u256_select(is_const, u256_small(1), *pz);
        }
        return p;
    };
    Sel s_cur = sel_of(it0), s_nxt = sel_of(it0 + 1);
    Jac p2_cur = point_of(s_cur.src);   // prologue: exposed once per piece
    // nothing may be in flight when the loop is entered: the compiler sizes the waits at the loop head for the SHORTER
    // of the two ways in (this prologue, with a handful of loads behind the digit) and would wait for most of an
    // iteration's stores every time round
    settle(s_cur.digit);
    settle(s_cur.src);
    settle(s_nxt.digit);
    settle(s_nxt.src);
    settle(p2_cur.X);
    settle(p2_cur.Y);
    settle(p2_cur.Z);
    // the running point, Z^2 of it and the prefix product live on lazy limbs inside the loop (quad29.hpp)
    JacL c = jacl_from(cur);
    F29 c_zz = f29_from_u256(cur_zz), a29 = f29_from_u256(acc);
    for (int it = it0; it < it1; it++) {
        const int t = lb + 3 * it;
        // ---- loads of this iteration, all up front
        const Sel s_n2 = sel_of(it + 2);
        const Jac p2_nxt = point_of(s_nxt.src);
        const OpDesc o0 = load_op(B.ops, t), o1 = load_op(B.ops, t + 1), o2 = load_op(B.ops, t + 2);
        // ---- double, double
        for (int k = 0; k < 2; k++) {
            const uint8_t fl = k ? o1.flags : o0.flags;
            const QuadRes29 q = jac_dbl_quad29(role, (fl & F_NO_AFFINE) != 0, c, a29);
            B.src[(size_t)(2 * (t + k)) * B.n + i] = cur_id;   // (all four lanes: same value)
            quad_store_op29(B, i, role, t + k, fl, q);
            a29 = q.acc;
            c = q.p;
            c_zz = q.zz3;
            cur_id = (uint16_t)(t + k);
        }
        // ---- conditional add of the selected table entry (the sum is always computed: quirk Q7)
        const JacL p2 = jacl_from(p2_cur);
        const QuadRes29 q = TA ? jac_add_quad29<false, true>(role, (o2.flags & F_NO_AFFINE) != 0, c, true, c_zz, p2, a29)
                               : jac_add_quad29<false, false>(role, (o2.flags & F_NO_AFFINE) != 0, c, false, c_zz, p2, a29);
        const bool take = s_cur.digit != 0;
        dyn_idx = o2.cadd_idx;
        dyn_val = take ? (uint16_t)(t + 2) : cur_id;
        B.src[(size_t)(2 * (t + 2)) * B.n + i] = cur_id;
        B.src[(size_t)(2 * (t + 2) + 1) * B.n + i] = (uint16_t)(s_cur.src | (take ? SRC_SEL_BIT : 0));
        B.dyn[(size_t)o2.cadd_idx * B.n + i] = dyn_val;
        quad_store_op29(B, i, role, t + 2, o2.flags, q);
        a29 = q.acc;
        c.X = f29_select(take, q.p.X, c.X);
        c.Y = f29_select(take, q.p.Y, c.Y);
        c.Z = f29_select(take, q.p.Z, c.Z);
        c_zz = f29_select(take, q.zz3, q.zz1);
        cur_id = dyn_val;
        s_cur = s_nxt;
        s_nxt = s_n2;
        p2_cur = p2_nxt;
    }
    cur = jacl_canon(c);
    cur_zz = f29_canon(c_zz);
    acc = f29_canon(a29);
}
// fixed-base windows [t0, t0 + count): one conditional add of fbtab[window][digit] each (gadgets/curve_fixed_base.rs:43-62)
P2E_HD void body_fb_windows_quad(const Buffers& B, size_t i, int role, int t0, int count, Jac& cur, uint16_t& cur_id, U256& acc,
                                 U256& cur_zz, bool have_zz, uint16_t& dyn_idx, uint16_t& dyn_val) {
    const int t1 = t0 + count;
    auto window_of = [&](int t) { return ref_id(load_op(B.ops, t < t1 ? t : t1 - 1).ref2); };
    auto digit_of_op = [&](int t) { return (u32)B.dig4[(size_t)window_of(t) * B.n + i]; };
    u32 d_cur = digit_of_op(t0), d_nxt = digit_of_op(t0 + 1);
    Aff p2_cur = B.fbtab[window_of(t0) * 16 + d_cur];   // prologue
    settle(d_cur);
    settle(d_nxt);
    settle(p2_cur.x);
    settle(p2_cur.y);
    JacL c = jacl_from(cur);
    F29 c_zz = f29_from_u256(cur_zz), a29 = f29_from_u256(acc);
    for (int t = t0; t < t1; t++) {
        const u32 d_n2 = digit_of_op(t + 2);
        const Aff p2_nxt = B.fbtab[window_of(t + 1) * 16 + d_nxt];
        const OpDesc o = load_op(B.ops, t);
        JacL p2;
        p2.X = f29_from_u256(p2_cur.x);
        p2.Y = f29_from_u256(p2_cur.y);
        p2.Z = f29_small(1);
        const QuadRes29 q = jac_add_quad29<false, true>(role, (o.flags & F_NO_AFFINE) != 0, c, have_zz, c_zz, p2, a29);
        const bool take = d_cur != 0;
        dyn_idx = o.cadd_idx;
        dyn_val = take ? (uint16_t)t : cur_id;
        B.src[(size_t)(2 * t) * B.n + i] = cur_id;
        B.src[(size_t)(2 * t + 1) * B.n + i] = (uint16_t)(SRC_FB_BIT | (ref_id(o.ref2) * 16 + d_cur) | (take ? SRC_SEL_BIT : 0));
        B.dyn[(size_t)o.cadd_idx * B.n + i] = dyn_val;
        quad_store_op29(B, i, role, t, o.flags, q);
        a29 = q.acc;
        c.X = f29_select(take, q.p.X, c.X);
        c.Y = f29_select(take, q.p.Y, c.Y);
        c.Z = f29_select(take, q.p.Z, c.Z);
        c_zz = f29_select(take, q.zz3, q.zz1);
        have_zz = true;
        cur_id = dyn_val;
        d_cur = d_nxt;
        d_nxt = d_n2;
        p2_cur = p2_nxt;
    }
    cur = jacl_canon(c);
    cur_zz = f29_canon(c_zz);
    acc = f29_canon(a29);
}

// ops [lo, hi) of one chain for a quad: the two loops above wherever the range contains them, the generic op for
// the rest (window-table build, unblinding adds, the final add)
P2E_HD void body_chain_range_quad(const Program& G, const Buffers& B, size_t i, int role, int lo, int hi, bool table_affine,
                                  bool continue_prefix) {
    ChainStateQ st;
    st.out_id = st.p1_id = st.dyn_idx = st.dyn_val = 0xFFFF;
    st.out.X = st.out.Y = st.out.Z = st.p1.X = st.p1.Y = st.p1.Z = st.out_zz = st.p1_zz = u256_zero();
    st.acc = continue_prefix ? range_product(B, i, lo - 1) : u256_small(1);
    const int lb = G.msm_loop_begin, le = lb + 3 * G.msm_loop_iters;
    int t = lo;
    while (t < hi) {
        const OpDesc op = load_op(B.ops, t);
        const bool msm_iter = t >= lb && t + 3 <= le && t + 3 <= hi && (t - lb) % 3 == 0;
        const bool fb_window = op.kind == OP_CADD && ref_kind(op.ref2) == R_FBTAB;
        if (msm_iter || fb_window) {
            // the loop's running point = this op's first operand, from the registers of the ops before or from scratch
            const P1Sel s1 = quad_first_operand(G, B, i, op, st);
            Jac cur = s1.p1;
            uint16_t cur_id = s1.src1;
            U256 cur_zz = s1.zz1;
            int done;
            if (msm_iter) {
                const int it0 = (t - lb) / 3, hi_it = ((hi < le ? hi : le) - lb) / 3;
                if (table_affine)
                    body_msm_iters_quad<true>(G, B, i, role, it0, hi_it, cur, cur_id, st.acc, cur_zz, st.dyn_idx, st.dyn_val);
                else
                    body_msm_iters_quad<false>(G, B, i, role, it0, hi_it, cur, cur_id, st.acc, cur_zz, st.dyn_idx, st.dyn_val);
                done = 3 * (hi_it - it0);
            } else {
                int count = 1;   // consecutive windows (uniform: read off the op table)
                while (t + count < hi) {
                    const OpDesc nx = load_op(B.ops, t + count);
                    if (!(nx.kind == OP_CADD && ref_kind(nx.ref2) == R_FBTAB)) break;
                    count++;
                }
                body_fb_windows_quad(B, i, role, t, count, cur, cur_id, st.acc, cur_zz, s1.have_zz1, st.dyn_idx, st.dyn_val);
                done = count;
            }
            // hand the running point back to the generic walker: it is what the last conditional add selected
            st.out = cur;
            st.out_zz = cur_zz;
            st.out_id = cur_id;
            st.p1_id = 0xFFFF;
            t += done;
        } else {
            const P1Sel s1 = quad_first_operand(G, B, i, op, st);
            body_chain_op_quad(G, B, i, role, t, op, table_affine, s1, st);
            t++;
        }
    }
}

// Phase B of ops [t0, t1) with the backward pass cut into S sub-ranges; this lane walks sub-range q.  With phase A's
// cumulative prefix products (have_prefix) a sub-range starts from the inverse of the product THROUGH its last op,
// which is one inversion of its own: S inversions run side by side instead of one followed by a t1 - t0 long walk.
template <class CV = Secp256k1>
P2E_HD void body_batch_inv_split(const Program& G, const Buffers& B, size_t i, int t0, int t1, bool have_prefix, int q, int S) {
    const int len = t1 - t0;
    const int a = t0 + (int)(((long long)len * q) / S), b = t0 + (int)(((long long)len * (q + 1)) / S);
    if (a >= b) return;
    // (have_prefix: PREF[t] is the product from the piece's first op, so [a, b) needs no forward pass of its own)
    body_batch_inv<CV>(G, B, i, a, b, have_prefix, false);   // lanes of one wave walk different ops
}

// ---- curve programs (curves.hpp): any op list of either curve, four lanes per signature ------------------------------
// body_chain_op<CV, true> (pipeline.hpp) for a quad: the generic walker only -- the windowed loop is not written out as
// straight-line code the way the built-in MSM loop is, so operand loads are settled where they are issued.
template <class CV>
P2E_HD void body_chain_op_quad_cv(const Program& G, const Buffers& B, size_t i, int role, int t, const OpDesc& op, bool table_affine,
                                  const P1Sel& s1, ChainStateQ& st) {
    const uint16_t src1 = s1.src1;
    const Jac& p1 = s1.p1;
    if (role == 0) B.src[(size_t)(2 * t) * B.n + i] = src1;
    QuadRes q;
    if (op.kind == OP_DBL) {
        q = jac_dbl_quad_cv<CV>(role, p1, st.acc);
    } else {
        Jac p2;
        u32 digit = 1;
        uint16_t src2;
        bool z2one = (op.flags & F_Z2ONE) != 0;
        if (ref_kind(op.ref2) == R_SELSLOT) {   // curve_scalar_mul: result + 2^i p, selected by bit i
            digit = B.dig2[(size_t)(ref_id(op.ref2) >> 12) * B.n + i];
            src2 = (uint16_t)(ref_id(op.ref2) & 0xFFFu);
            p2 = load_jac_src(B, i, src2, z2one);
        } else if (ref_kind(op.ref2) == R_FBTAB) {
            Aff a = load_fbtab(B, i, ref_id(op.ref2), digit);
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
        settle(p2.X);
        settle(p2.Y);
        settle(p2.Z);
        settle(digit);
        if (role == 0) B.src[(size_t)(2 * t + 1) * B.n + i] = (uint16_t)(src2 | (digit != 0 ? SRC_SEL_BIT : 0));
        if ((op.flags & F_Z1ONE) && z2one)
            q = jac_add_quad_cv<CV, true, true>(role, p1, false, s1.zz1, p2, st.acc);
        else if (z2one)
            q = jac_add_quad_cv<CV, false, true>(role, p1, s1.have_zz1, s1.zz1, p2, st.acc);
        else if (op.flags & F_Z1ONE)
            q = jac_add_quad_cv<CV, true, false>(role, p1, false, s1.zz1, p2, st.acc);
        else
            q = jac_add_quad_cv<CV, false, false>(role, p1, false, s1.zz1, p2, st.acc);
        if (op.kind == OP_CADD) {
            st.dyn_idx = op.cadd_idx;
            st.dyn_val = digit != 0 ? (uint16_t)t : src1;
            if (role == 0) B.dyn[(size_t)op.cadd_idx * B.n + i] = st.dyn_val;
        }
    }
    quad_store_op(B, i, role, t, op.flags, q, st.acc);
    st.acc = q.acc;
    st.p1 = p1;
    st.p1_zz = q.zz1;
    st.p1_id = (op.flags & F_Z1ONE) ? (uint16_t)0xFFFF : src1;
    st.out = q.res.p;
    st.out_zz = q.zz3;
    st.out_id = (uint16_t)t;
}
// ops [lo, hi) of a curve program's chain for a quad (its own inversion batch: the prefix starts at one)
template <class CV>
P2E_HD void body_chain_range_quad_cv(const Program& G, const Buffers& B, size_t i, int role, int lo, int hi, bool table_affine) {
    ChainStateQ st;
    st.out_id = st.p1_id = st.dyn_idx = st.dyn_val = 0xFFFF;
    st.out.X = st.out.Y = st.out.Z = st.p1.X = st.p1.Y = st.p1.Z = st.out_zz = st.p1_zz = u256_zero();
    st.acc = u256_small(1);
    for (int t = lo; t < hi; t++) {
        const OpDesc op = load_op(B.ops, t);
        const P1Sel s1 = quad_first_operand(G, B, i, op, st);
        body_chain_op_quad_cv<CV>(G, B, i, role, t, op, table_affine, s1, st);
    }
}

}  // namespace p2e
