// Host-side constant tables and synthetic-input generator (HOST ONLY).
//   rando / -rando / -(2^146 rando): blinding points of gadgets/curve_fixed_base.rs:34-38,64 and
//       gadgets/curve_msm.rs:33-39,74-75
//   fixed-base window table t*16^i*G, t=1..15 (slot 0 := slot 1): gadgets/curve_fixed_base.rs:24-30,45-56
//   synthetic signatures: curve/ecdsa.rs:25-40 sign_message
#pragma once
#include <vector>

#include "pipeline.hpp"

namespace p2e {
namespace host {

inline U256 u256_from_u64(u64 a, u64 b, u64 c, u64 d) {
    U256 r;
    u64 v[4] = {a, b, c, d};
    for (int i = 0; i < 4; i++) {
        r.w[2 * i] = (u32)v[i];
        r.w[2 * i + 1] = (u32)(v[i] >> 32);
    }
    return r;
}
inline Aff generator() {  // curve/secp256k1.rs:25-38
    Aff g;
    g.x = u256_from_u64(0x59F2815B16F81798ull, 0x029BFCDB2DCE28D9ull, 0x55A06295CE870B07ull, 0x79BE667EF9DCBBACull);
    g.y = u256_from_u64(0x9C47D08FFB10D4B8ull, 0xFD17B448A6855419ull, 0x5DA4FBFC0E1108A8ull, 0x483ADA7726A3C465ull);
    return g;
}
inline Aff jac_to_aff(const Jac& j) {
    U256 zi = fe_inv_p(j.Z);
    U256 zi2 = fp_sqr(zi);
    Aff a;
    a.x = fp_mul(j.X, zi2);
    a.y = fp_mul(j.Y, fp_mul(zi2, zi));
    return a;
}
// k*P for k != 0 (mod n), P of prime order: plain left-to-right double-and-add in Jacobian coordinates.
// The running point is never +-P before an add (k < n), so the incomplete formulas are safe.
inline Aff scalar_mul(const U256& k, const Aff& p) {
    int top = 255;
    while (top >= 0 && !((k.w[top >> 5] >> (top & 31)) & 1)) top--;
    Jac acc = jac_from_aff(p);
    Jac pj = jac_from_aff(p);
    for (int i = top - 1; i >= 0; i--) {
        acc = jac_dbl(acc).p;
        if ((k.w[i >> 5] >> (i & 31)) & 1) acc = jac_add<false, true>(acc, pj).p;
    }
    return jac_to_aff(acc);
}

struct Consts {
    Aff cpts[NUM_CONST_PTS];
    std::vector<Aff> fbtab;  // [66][16]
};
inline const Consts& consts() {
    static const Consts C = [] {
        Consts c;
        // keccak256(0u64 little-endian) read as a little-endian integer, NOT reduced
        // (gadgets/curve_fixed_base.rs:34-37; digest re-derived in tests/ with an independent Keccak)
        static const uint8_t h[32] = {0x01, 0x1b, 0x4d, 0x03, 0xdd, 0x8c, 0x01, 0xf1, 0x04, 0x91, 0x43,
                                      0xcf, 0x9c, 0x4c, 0x81, 0x7e, 0x4b, 0x16, 0x7f, 0x1d, 0x1b, 0x83,
                                      0xe5, 0xc6, 0xf0, 0xf1, 0x0d, 0x89, 0xba, 0x1e, 0x7b, 0xce};
        U256 hs;
        for (int i = 0; i < 8; i++)
            hs.w[i] = (u32)h[4 * i] | ((u32)h[4 * i + 1] << 8) | ((u32)h[4 * i + 2] << 16) | ((u32)h[4 * i + 3] << 24);
        Aff g = generator();
        Aff rando = scalar_mul(hs, g);
        c.cpts[CONST_RANDO] = rando;
        c.cpts[CONST_NEG_RANDO] = aff_neg(rando);
        Aff d = rando;
        for (int i = 0; i < 2 * MSM_DIGITS; i++) d = aff_dbl(d);
        c.cpts[CONST_NEG_RANDO_146] = aff_neg(d);
        c.fbtab.resize(FB_WINDOWS * 16);
        Aff base = g;
        for (int w = 0; w < FB_WINDOWS; w++) {
            Aff acc = base;
            for (int t = 1; t < 16; t++) {
                c.fbtab[w * 16 + t] = acc;
                if (t < 15) acc = (t == 1) ? aff_dbl(acc) : aff_add(acc, base);
            }
            c.fbtab[w * 16 + 0] = c.fbtab[w * 16 + 1];
            for (int i = 0; i < 4; i++) base = aff_dbl(base);
        }
        return c;
    }();
    return C;
}

// ---- synthetic signatures --------------------------------------------------------------------------
struct SplitMix64 {
    u64 s;
    u64 next() {
        s += 0x9E3779B97F4A7C15ull;
        u64 z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    U256 below_n() {  // uniform in [1, n)
        for (;;) {
            u64 a = next(), b = next(), c = next(), d = next();
            U256 v = u256_from_u64(a, b, c, d);
            if (!u256_is_zero(v) && !geq_mod<ModN>(v.w)) return v;
        }
    }
};
// k*G through the fixed-base table (64 mixed additions)
inline Aff base_mul(const U256& k) {
    const Consts& C = consts();
    Jac acc{};
    bool have = false;
    for (int w = 0; w < 64; w++) {
        u32 d = (k.w[w >> 3] >> ((w & 7) * 4)) & 15;
        if (!d) continue;
        const Aff& a = C.fbtab[w * 16 + d];
        if (!have) {
            acc = jac_from_aff(a);
            have = true;
        } else {
            acc = jac_add<false, true>(acc, jac_from_aff(a)).p;
        }
    }
    return jac_to_aff(acc);
}
inline void synth_signature(u64 seed, u64 index, U256& msg, U256& r, U256& s, Aff& pk) {
    SplitMix64 rng{seed ^ (0x9E3779B97F4A7C15ull * (index + 1))};
    for (;;) {
        U256 sk = rng.below_n();
        msg = rng.below_n();
        U256 k = rng.below_n();
        pk = base_mul(sk);
        Aff rr = base_mul(k);
        if (u256_is_zero(rr.x)) continue;
        r = fe_canon<ModN>(rr.x);  // base_to_scalar curve/curve_types.rs:280-282
        s = fe_mul<ModN>(fe_inv_n(k), fe_add<ModN>(msg, fe_mul<ModN>(r, sk)));
        if (u256_is_zero(r) || u256_is_zero(s)) continue;
        return;
    }
}

}  // namespace host
}  // namespace p2e
