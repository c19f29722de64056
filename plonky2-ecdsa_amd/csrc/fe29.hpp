// secp256k1 base-field arithmetic on UNSATURATED 29-bit limbs (9 x u32 per value), for the latency-bound chains.
//
// fe.hpp keeps every value canonical in 8 x 32-bit words: one multiplication is 64 v_mad_u64_u32 each followed by a
// v_addc for the third accumulator word, and every addition / subtraction is two or three 8-long carry chains through
// VCC (on gfx950 the hazard recognizer puts two wait states between the links).  A lone wave -- the small-batch plan
// has at most one chain wave per SIMD -- pays the full latency of all of it: ~1 750 cycles per multiplication, ~200 per
// addition, and the additions between the multiplications are a third of a curve op.
//
// Here a value is sum l[k] 2^(29 k) with limbs allowed to exceed 29 bits ("lazy"): additions and subtractions are nine
// independent 32-bit operations without carries, a 9 x 9 product column sums up in 64 bits without a carry word
// (81 v_mad_u64_u32 and nothing between them), and reduction folds the high half through
//     2^261 = 2^5 (2^32 + 977) = 31 264 + 2^8 2^29   (mod p),
// i.e. one small multiplication and one shift per limb.  The price is bookkeeping: every function states the limb
// bounds it accepts and guarantees, the CPU emulation build (tests/emu, -DP2E_F29_BOUNDS) carries a worst-case bound
// beside every limb through exactly the code the GPU runs and aborts on the first operation whose precondition
// could be violated, and tests/test_host.py drives every formula variant through it.
// Values are only ever COMPARED or STORED in canonical form (f29_canon): what reaches memory is bit-identical to fe.hpp's.
//
// Used by quad.hpp (four lanes per signature, secp256k1).  Same formulas as ec.hpp / reference curve/curve_types.rs.
#pragma once
#include "fe.hpp"

namespace p2e {

constexpr u32 F29_M = 0x1FFFFFFFu;
constexpr u32 F29_R0 = 31264u;   // 2^261 mod p = R0 + 2^F29_R1S * 2^29
constexpr int F29_R1S = 8;
// "tight": what f29_mul / f29_sqr / f29_norm / f29_from_u256 return.  Classes below are multiples of it.
constexpr u64 F29_T = (1ull << 29) + (1ull << 20);

#if defined(P2E_F29_BOUNDS) && !defined(__HIP_DEVICE_COMPILE__)
#define P2E_F29_TRACK 1
#else
#define P2E_F29_TRACK 0
#endif

struct F29 {
    u32 l[9];
#if P2E_F29_TRACK
    u64 ub[9];   // emulation build only: an upper bound of every limb that holds for ALL inputs reaching this point
#endif
};

#if P2E_F29_TRACK
}  // namespace p2e
#include <cstdio>
#include <cstdlib>
namespace p2e {
[[noreturn]] inline void f29_bound_fail(const char* what) {
    fprintf(stderr, "fe29.hpp: limb bound violated in %s\n", what);
    abort();
}
inline void f29_need(bool ok, const char* what) {
    if (!ok) f29_bound_fail(what);
}
typedef unsigned __int128 u128_;
#endif

// multiples of p whose limbs all lie in [m T, m T + 2^29): a - b is computed as a + (F29_SUBC[m] - b) for b of class m
// (limbs <= m T); the result is of class (class of a) + m + 1
P2E_HD u32 f29_subc(int m, int k) {
    // K p for K = 33, 65, 97, 129 written with borrowed limbs (tools/f29_constants.py prints and checks them)
    const u32 c0[4] = {0x3fff820fu, 0x5fff07efu, 0x7ffe8dcfu, 0x9ffe13afu};
    const u32 c1[4] = {0x3ffffef6u, 0x5ffffdf5u, 0x7ffffcf4u, 0x9ffffbf3u};
    const u32 cm[4] = {0x3ffffffeu, 0x5ffffffdu, 0x7ffffffcu, 0x9ffffffbu};
    const u32 c8[4] = {0x20fffffeu, 0x40fffffdu, 0x60fffffcu, 0x80fffffbu};
    return k == 0 ? c0[m - 1] : k == 1 ? c1[m - 1] : k == 8 ? c8[m - 1] : cm[m - 1];
}

P2E_HD F29 f29_from_u256(const U256& a) {
    F29 r;
    split29(a, r.l);
#if P2E_F29_TRACK
    for (int k = 0; k < 9; k++) r.ub[k] = k == 8 ? 0xFFFFFFu : F29_M;
#endif
    return r;
}
P2E_HD F29 f29_small(u32 v) {   // v < 2^29
    F29 r;
    P2E_UNROLL
    for (int k = 0; k < 9; k++) r.l[k] = k ? 0 : v;
#if P2E_F29_TRACK
    for (int k = 0; k < 9; k++) r.ub[k] = k ? 0 : v;
#endif
    return r;
}
P2E_HD F29 f29_select(bool c, const F29& t, const F29& f) {
    F29 r;
    P2E_UNROLL
    for (int k = 0; k < 9; k++) r.l[k] = c ? t.l[k] : f.l[k];
#if P2E_F29_TRACK
    for (int k = 0; k < 9; k++) r.ub[k] = t.ub[k] > f.ub[k] ? t.ub[k] : f.ub[k];   // (either may be taken)
#endif
    return r;
}

// a + b, limb by limb
P2E_HD F29 f29_add(const F29& a, const F29& b) {
    F29 r;
    P2E_UNROLL
    for (int k = 0; k < 9; k++) r.l[k] = a.l[k] + b.l[k];
#if P2E_F29_TRACK
    for (int k = 0; k < 9; k++) {
        r.ub[k] = a.ub[k] + b.ub[k];
        f29_need(r.ub[k] <= 0xFFFFFFFFull, "f29_add");
    }
#endif
    return r;
}
// K a for a small constant K
template <u32 K>
P2E_HD F29 f29_times(const F29& a) {
    F29 r;
    P2E_UNROLL
    for (int k = 0; k < 9; k++) r.l[k] = a.l[k] * K;
#if P2E_F29_TRACK
    for (int k = 0; k < 9; k++) {
        r.ub[k] = a.ub[k] * K;
        f29_need(r.ub[k] <= 0xFFFFFFFFull, "f29_times");
    }
#endif
    return r;
}
// a - b for b of class M (every limb of b <= the matching limb of the constant)
template <int M>
P2E_HD F29 f29_sub(const F29& a, const F29& b) {
    static_assert(M >= 1 && M <= 4, "classes 1..4");
    F29 r;
    P2E_UNROLL
    for (int k = 0; k < 9; k++) r.l[k] = a.l[k] + (f29_subc(M, k) - b.l[k]);
#if P2E_F29_TRACK
    for (int k = 0; k < 9; k++) {
        f29_need(b.ub[k] <= f29_subc(M, k), "f29_sub: subtrahend above its class");
        r.ub[k] = a.ub[k] + f29_subc(M, k);
        f29_need(r.ub[k] <= 0xFFFFFFFFull, "f29_sub");
    }
#endif
    return r;
}

// weak normalisation: every limb keeps its low 29 bits and takes the overflow of the limb below, all nine at once (no
// chain); the top limb's overflow comes back through 2^261 = R0 + 2^8 2^29.  Any limbs in -> tight limbs out.
P2E_HD F29 f29_norm(const F29& a) {
    F29 r;
    const u32 top = a.l[8] >> 29;
    r.l[0] = (a.l[0] & F29_M) + top * F29_R0;
    r.l[1] = (a.l[1] & F29_M) + (a.l[0] >> 29) + (top << F29_R1S);
    P2E_UNROLL
    for (int k = 2; k < 9; k++) r.l[k] = (a.l[k] & F29_M) + (a.l[k - 1] >> 29);
#if P2E_F29_TRACK
    const u64 topb = a.ub[8] >> 29;
    r.ub[0] = F29_M + topb * F29_R0;
    r.ub[1] = F29_M + (a.ub[0] >> 29) + (topb << F29_R1S);
    for (int k = 2; k < 9; k++) r.ub[k] = F29_M + (a.ub[k - 1] >> 29);
    for (int k = 0; k < 9; k++) f29_need(r.ub[k] <= F29_T, "f29_norm: result not tight");
#endif
    return r;
}

// shared tail of f29_mul / f29_sqr: c[0..16] = the 17 column sums (each < 2^64 - 2^48) -> tight limbs.
// The high columns are split into 29-bit limbs h[0..8] first (their carry chain does not need the low half), then the
// low columns take  R0 h[k] + 2^8 h[k-1]  on the way through their own carry chain; what leaves limb 8 (< 2^41) is
// folded once more into limbs 0..3.
#if P2E_F29_TRACK
inline void f29_reduce_bounds(const u128_* cb, F29& r, const char* what) {
    u128_ t = 0, hb[9];
    for (int k = 9; k <= 16; k++) {
        t = (t >> 29) + cb[k];
        f29_need(t < ((u128_)1 << 64), what);
        hb[k - 9] = F29_M;
    }
    hb[8] = t >> 29;
    f29_need(hb[8] <= 0xFFFFFFFFull, what);
    t = 0;
    for (int k = 0; k <= 8; k++) {
        t = (t >> 29) + cb[k] + hb[k] * F29_R0 + (k ? hb[k - 1] << F29_R1S : 0);
        f29_need(t < ((u128_)1 << 64), what);
    }
    const u128_ e9 = (t >> 29) + (hb[8] << F29_R1S);
    f29_need(e9 * F29_R0 + F29_M < ((u128_)1 << 64), what);
    for (int k = 0; k < 9; k++) r.ub[k] = k == 3 ? F29_M + 1 : F29_M;
}
#endif

P2E_HD F29 f29_mul(const F29& a, const F29& b) {
    F29 r;
    u32 h[9];
    u64 t = 0;
    P2E_UNROLL
    for (int k = 9; k <= 16; k++) {
        t >>= 29;
        P2E_UNROLL
        for (int i = k - 8; i <= 8; i++) t += (u64)a.l[i] * b.l[k - i];
        h[k - 9] = (u32)t & F29_M;
    }
    h[8] = (u32)(t >> 29);
    t = 0;
    P2E_UNROLL
    for (int k = 0; k <= 8; k++) {
        t >>= 29;
        P2E_UNROLL
        for (int i = 0; i <= k; i++) t += (u64)a.l[i] * b.l[k - i];
        t += (u64)h[k] * F29_R0;
        if (k) t += (u64)h[k - 1] << F29_R1S;
        r.l[k] = (u32)t & F29_M;
    }
    const u64 e9 = (t >> 29) + ((u64)h[8] << F29_R1S);
    t = r.l[0] + e9 * F29_R0;
    r.l[0] = (u32)t & F29_M;
    t = (t >> 29) + r.l[1] + (e9 << F29_R1S);
    r.l[1] = (u32)t & F29_M;
    t = (t >> 29) + r.l[2];
    r.l[2] = (u32)t & F29_M;
    r.l[3] += (u32)(t >> 29);
#if P2E_F29_TRACK
    u128_ cb[17];
    for (int k = 0; k <= 16; k++) {
        cb[k] = 0;
        for (int i = 0; i <= 8; i++)
            if (k - i >= 0 && k - i <= 8) cb[k] += (u128_)a.ub[i] * b.ub[k - i];
    }
    f29_reduce_bounds(cb, r, "f29_mul: operands too large for 64-bit columns");
#endif
    return r;
}

// a^2: the 36 cross products once against the doubled operand, the 9 squares on the even columns (45 multiplications)
P2E_HD F29 f29_sqr(const F29& a) {
    F29 r;
    u32 d[9], h[9];
    P2E_UNROLL
    for (int i = 0; i < 9; i++) d[i] = a.l[i] << 1;
    u64 t = 0;
    P2E_UNROLL
    for (int k = 9; k <= 16; k++) {
        t >>= 29;
        P2E_UNROLL
        for (int i = k - 8; 2 * i < k; i++) t += (u64)d[i] * a.l[k - i];
        if (k % 2 == 0) t += (u64)a.l[k / 2] * a.l[k / 2];
        h[k - 9] = (u32)t & F29_M;
    }
    h[8] = (u32)(t >> 29);
    t = 0;
    P2E_UNROLL
    for (int k = 0; k <= 8; k++) {
        t >>= 29;
        P2E_UNROLL
        for (int i = 0; 2 * i < k; i++) t += (u64)d[i] * a.l[k - i];
        if (k % 2 == 0) t += (u64)a.l[k / 2] * a.l[k / 2];
        t += (u64)h[k] * F29_R0;
        if (k) t += (u64)h[k - 1] << F29_R1S;
        r.l[k] = (u32)t & F29_M;
    }
    const u64 e9 = (t >> 29) + ((u64)h[8] << F29_R1S);
    t = r.l[0] + e9 * F29_R0;
    r.l[0] = (u32)t & F29_M;
    t = (t >> 29) + r.l[1] + (e9 << F29_R1S);
    r.l[1] = (u32)t & F29_M;
    t = (t >> 29) + r.l[2];
    r.l[2] = (u32)t & F29_M;
    r.l[3] += (u32)(t >> 29);
#if P2E_F29_TRACK
    u128_ cb[17];
    for (int i = 0; i < 9; i++) f29_need(a.ub[i] <= 0x7FFFFFFFull, "f29_sqr: operand cannot be doubled");
    for (int k = 0; k <= 16; k++) {
        cb[k] = 0;
        for (int i = 0; i <= 8; i++)
            if (k - i >= 0 && k - i <= 8) cb[k] += (u128_)a.ub[i] * a.ub[k - i];
    }
    f29_reduce_bounds(cb, r, "f29_sqr: operand too large for 64-bit columns");
#endif
    return r;
}

// Out-of-line forms for the kernels whose walkers would otherwise inline a hundred multiplications (lane per signature:
// every variant of the addition and the doubling, ~80 KB of code against a 64 KB instruction cache shared by two CUs
// and by the expansion kernels running beside the chains).  Arguments and result travel in VGPRs, as for fe_mul_call.
// (eighteen scalars: clang passes at most 16 registers' worth of aggregates per call and would hand the second operand
// over in scratch memory)
template <int UNUSED = 0>
P2E_HD_NOINLINE F29 f29_mul_call_regs(u32 a0, u32 a1, u32 a2, u32 a3, u32 a4, u32 a5, u32 a6, u32 a7, u32 a8, u32 b0, u32 b1, u32 b2,
                                      u32 b3, u32 b4, u32 b5, u32 b6, u32 b7, u32 b8) {
    F29 a, b;
    a.l[0] = a0, a.l[1] = a1, a.l[2] = a2, a.l[3] = a3, a.l[4] = a4, a.l[5] = a5, a.l[6] = a6, a.l[7] = a7, a.l[8] = a8;
    b.l[0] = b0, b.l[1] = b1, b.l[2] = b2, b.l[3] = b3, b.l[4] = b4, b.l[5] = b5, b.l[6] = b6, b.l[7] = b7, b.l[8] = b8;
    return f29_mul(a, b);
}
P2E_HD F29 f29_mul_call(const F29& a, const F29& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return f29_mul_call_regs(a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], a.l[6], a.l[7], a.l[8], b.l[0], b.l[1], b.l[2], b.l[3],
                             b.l[4], b.l[5], b.l[6], b.l[7], b.l[8]);
#else
    return f29_mul(a, b);   // (the emulation build keeps the bounds that ride beside the limbs)
#endif
}
template <int UNUSED = 0>
P2E_HD_NOINLINE F29 f29_sqr_call(F29 a) {
    return f29_sqr(a);
}

// The canonical 8-word form (what fe.hpp computes with, what scratch memory and the witness hold): fold the bits above
// 2^256 through 2^256 = 2^32 + 977, one strict carry pass, repack to 32-bit words, one conditional subtraction of p.
// Any limbs below 2^31 in.
P2E_HD U256 f29_canon(const F29& a) {
#if P2E_F29_TRACK
    for (int k = 0; k < 9; k++) f29_need(a.ub[k] < (1ull << 31), "f29_canon");
#endif
    const u32 hi = a.l[8] >> 24;
    u32 s[9];
    u32 t = a.l[0] + hi * 977u;
    s[0] = t & F29_M;
    t = (t >> 29) + a.l[1] + (hi << 3);
    s[1] = t & F29_M;
    P2E_UNROLL
    for (int k = 2; k < 8; k++) {
        t = (t >> 29) + a.l[k];
        s[k] = t & F29_M;
    }
    s[8] = (t >> 29) + (a.l[8] & 0xFFFFFFu);   // <= 2^24 + 3: bit 24 is 2^256
    U256 x;
    x.w[0] = s[0] | (s[1] << 29);
    x.w[1] = (s[1] >> 3) | (s[2] << 26);
    x.w[2] = (s[2] >> 6) | (s[3] << 23);
    x.w[3] = (s[3] >> 9) | (s[4] << 20);
    x.w[4] = (s[4] >> 12) | (s[5] << 17);
    x.w[5] = (s[5] >> 15) | (s[6] << 14);
    x.w[6] = (s[6] >> 18) | (s[7] << 11);
    x.w[7] = (s[7] >> 21) | (s[8] << 8);
    const u32 over = s[8] >> 24;
    // x + 2^256 over < 2 p:  subtract p (= add 2^32 + 977 and drop 2^256) if that does not go negative
    U256 y;
    u32 c = 0;
    y.w[0] = addc32(x.w[0], 977u, c);
    y.w[1] = addc32(x.w[1], 1u, c);
    P2E_UNROLL
    for (int k = 2; k < 8; k++) y.w[k] = addc32(x.w[k], 0u, c);
    return u256_select((c | over) != 0, y, x);
}
template <int UNUSED = 0>
P2E_HD_NOINLINE U256 f29_canon_call(F29 a) {
    return f29_canon(a);
}
P2E_HD bool f29_is_zero(const F29& a) { return u256_is_zero(f29_canon(a)); }

#if defined(__HIP_DEVICE_COMPILE__)
template <int J>
__device__ __forceinline__ F29 f29_bcast(const F29& v) {
    F29 r;
    P2E_UNROLL
    for (int k = 0; k < 9; k++)
        r.l[k] = (u32)__builtin_amdgcn_mov_dpp((int)v.l[k], J * 0x55 /* quad_perm:[J,J,J,J] */, 0xF, 0xF, true);
    return r;
}
#endif

}  // namespace p2e
