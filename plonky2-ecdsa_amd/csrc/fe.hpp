// 256-bit prime-field arithmetic for secp256k1 (base field p and scalar field n), written for the
// CDNA4 VALU: 8 x 32-bit limbs per value so every partial product is one v_mad_u64_u32, both moduli
// are of the form 2^256 - C so reduction is a short "fold the high half times C" chain (C = 2^32+977
// for p, a 129-bit constant for n) that also yields the exact integer quotient the plonky2
// MulNonnative / Inverse witness generators need.  No MFMA: integer carry chains.
//
// Everything here is __host__ __device__: the same code builds the constant tables on the host at
// context-create time and is exercised by the CPU emulation harness under tests/emu (test-only).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define P2E_HD __host__ __device__ __forceinline__
#define P2E_HD_NOINLINE __host__ __device__ __noinline__
#define P2E_UNROLL _Pragma("unroll")
#else
#define P2E_HD inline
#define P2E_HD_NOINLINE inline
#define P2E_UNROLL
#endif

namespace p2e {

typedef uint32_t u32;
typedef uint64_t u64;
typedef int64_t i64;

struct alignas(16) U256 {
    u32 w[8];
};

constexpr int BITS = 29;   // reference gadgets/nonnative.rs:32
constexpr int NL = 9;      // ceil(256/29), reference gates/mul_nonnative.rs:37-39
constexpr u32 MASK29 = (1u << BITS) - 1;
constexpr u64 P_GL = 0xFFFFFFFF00000001ull;  // Goldilocks

// error bits, mirrored in include/p2e.h
constexpr uint8_t ERR_LIMB_RANGE = 1, ERR_VALUE_GE_2_256 = 2, ERR_INVERSE_OF_ZERO = 4, ERR_CARRY_RANGE = 8,
                  ERR_QUOTIENT_RANGE = 16, ERR_DIVISION_BY_ZERO = 32;

// ------------------------------------------------------------------------------------------------
// moduli: m = 2^256 - C
// ------------------------------------------------------------------------------------------------
struct ModP {  // Secp256K1Base
    static constexpr int NC = 2;
    static constexpr bool kFourFolds = false;
    static constexpr bool kBarrett = false;
    P2E_HD static u32 k527(int i) {   // 2^-527 mod m (fe_inv_bingcd)
        constexpr u32 v[8] = {0xe02a12f7u, 0x77735922u, 0x2a6654feu, 0x518f6c84u, 0xc27a180au, 0x8c412c0du, 0x7a893ee2u, 0x81526b84u};
        return v[i];
    }
    P2E_HD static u32 c(int i) {
        constexpr u32 v[2] = {0x000003D1u, 0x00000001u};
        return v[i];
    }
    P2E_HD static u32 m(int i) {
        constexpr u32 v[8] = {0xFFFFFC2Fu, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu,
                              0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        return v[i];
    }
    // 29-bit limbs of m (convert_base(m, 32, 29))
    P2E_HD static u32 m29(int i) {
        constexpr u32 v[9] = {0x1FFFFC2Fu, 0x1FFFFFF7u, 0x1FFFFFFFu, 0x1FFFFFFFu, 0x1FFFFFFFu,
                              0x1FFFFFFFu, 0x1FFFFFFFu, 0x1FFFFFFFu, 0x00FFFFFFu};
        return v[i];
    }
};
struct ModN {  // Secp256K1Scalar
    static constexpr int NC = 5;
    static constexpr bool kFourFolds = true;
    static constexpr bool kBarrett = false;
    P2E_HD static u32 k527(int i) {
        constexpr u32 v[8] = {0x26886774u, 0x4608c517u, 0xb48257cbu, 0xed12990bu, 0xde6b7db0u, 0x51477db0u, 0x95d85510u, 0xd3235b67u};
        return v[i];
    }
    P2E_HD static u32 c(int i) {
        constexpr u32 v[5] = {0x2FC9BEBFu, 0x402DA173u, 0x50B75FC4u, 0x45512319u, 0x00000001u};
        return v[i];
    }
    P2E_HD static u32 m(int i) {
        constexpr u32 v[8] = {0xD0364141u, 0xBFD25E8Cu, 0xAF48A03Bu, 0xBAAEDCE6u,
                              0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        return v[i];
    }
    P2E_HD static u32 m29(int i) {
        constexpr u32 v[9] = {0x10364141u, 0x1E92F466u, 0x12280EEFu, 0x1DB9CD5Eu, 0x1FFFEBAAu,
                              0x1FFFFFFFu, 0x1FFFFFFFu, 0x1FFFFFFFu, 0x00FFFFFFu};
        return v[i];
    }
};

// NIST P-256 (reference field/p256_base.rs:14-17, field/p256_scalar.rs:5).  Neither modulus is
// 2^256 - small (C would be 224 bits), so these two reduce by Barrett (reduce_barrett below): mu = floor(2^512 / m).
struct ModP256 {  // P256Base
    static constexpr bool kBarrett = true;
    P2E_HD static u32 m(int i) {
        constexpr u32 v[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000001u, 0xFFFFFFFFu};
        return v[i];
    }
    P2E_HD static u32 m29(int i) {
        constexpr u32 v[9] = {0x1FFFFFFFu, 0x1FFFFFFFu, 0x1FFFFFFFu, 0x000001FFu, 0x00000000u, 0x00000000u, 0x00040000u, 0x1FE00000u, 0x00FFFFFFu};
        return v[i];
    }
    P2E_HD static u32 mu(int i) {
        constexpr u32 v[9] = {0x00000003u, 0x00000000u, 0xFFFFFFFFu, 0xFFFFFFFEu, 0xFFFFFFFEu, 0xFFFFFFFEu, 0xFFFFFFFFu, 0x00000000u, 0x00000001u};
        return v[i];
    }
    P2E_HD static u32 k527(int i) {
        constexpr u32 v[8] = {0x0019FFFFu, 0x000E0000u, 0xFFDE0000u, 0x00200000u, 0xFFF60000u, 0xFFEDFFFFu, 0x001E0000u, 0xFFEDFFFFu};
        return v[i];
    }
};
struct ModN256 {  // P256Scalar
    static constexpr bool kBarrett = true;
    P2E_HD static u32 m(int i) {
        constexpr u32 v[8] = {0xFC632551u, 0xF3B9CAC2u, 0xA7179E84u, 0xBCE6FAADu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x00000000u, 0xFFFFFFFFu};
        return v[i];
    }
    P2E_HD static u32 m29(int i) {
        constexpr u32 v[9] = {0x1C632551u, 0x1DCE5617u, 0x05E7A13Cu, 0x0DF55B4Eu, 0x1FFFFBCEu, 0x1FFFFFFFu, 0x0003FFFFu, 0x1FE00000u, 0x00FFFFFFu};
        return v[i];
    }
    P2E_HD static u32 mu(int i) {
        constexpr u32 v[9] = {0xEEDF9BFEu, 0x012FFD85u, 0xDF1A6C21u, 0x43190552u, 0xFFFFFFFFu, 0xFFFFFFFEu, 0xFFFFFFFFu, 0x00000000u, 0x00000001u};
        return v[i];
    }
    P2E_HD static u32 k527(int i) {
        constexpr u32 v[8] = {0x1C04824Bu, 0xD4C529B9u, 0x5C057EEDu, 0x2F26EFCEu, 0xEEB19D0Au, 0x88F0E491u, 0x33DB97B2u, 0x02F57AA6u};
        return v[i];
    }
};

// add / subtract with carry, 32-bit words.  clang lowers __builtin_addc/__builtin_subc chains to
// v_add_co_u32 / v_addc_co_u32 (one instruction per word, carry in VCC, no operand shuffling).
P2E_HD u32 addc32(u32 a, u32 b, u32& carry) {
#if defined(__clang__)
    unsigned co;
    u32 r = __builtin_addc(a, b, carry, &co);
    carry = co;
    return r;
#else
    u64 t = (u64)a + b + carry;
    carry = (u32)(t >> 32);
    return (u32)t;
#endif
}
P2E_HD u32 subb32(u32 a, u32 b, u32& borrow) {
#if defined(__clang__)
    unsigned bo;
    u32 r = __builtin_subc(a, b, borrow, &bo);
    borrow = bo;
    return r;
#else
    u64 t = (u64)a - b - borrow;
    borrow = (u32)(t >> 63);
    return (u32)t;
#endif
}

// ------------------------------------------------------------------------------------------------
// multi-word helpers (fully unrolled so the arrays live in VGPRs)
// ------------------------------------------------------------------------------------------------
// r[NA+NB] = a[NA] * b[NB].
// Device form: product scanning (column by column) with a 96-bit accumulator; every partial product is
//   v_mad_u64_u32 acc, vcc, a_i, b_j, acc      (64-bit accumulate, carry-out in VCC)
//   v_addc_co_u32 acc2, vcc, 0, acc2, vcc      (third accumulator word)
// i.e. two instructions per 32x32 product and no register shuffling (the compiler's own lowering of the
// portable loop below spends ~4 extra v_mov/v_lshl_add_u64 per product building 64-bit operand pairs).
// VCC written by the mad is consumed as carry-in by the very next instruction: the same back-to-back
// pattern hipcc emits for 64-bit adds, no wait states needed.
template <int NA, int NB>
P2E_HD void mul_wide(const u32* a, const u32* b, u32* r) {
#if defined(__HIP_DEVICE_COMPILE__)
    u64 acc = 0;
    u32 acc2 = 0;
    P2E_UNROLL
    for (int k = 0; k < NA + NB - 1; k++) {
        P2E_UNROLL
        for (int i = 0; i < NA; i++) {
            const int j = k - i;
            if (j >= 0 && j < NB) {
                asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc"
                    : "+v"(acc), "+v"(acc2)
                    : "v"(a[i]), "v"(b[j])
                    : "vcc");
            }
        }
        r[k] = (u32)acc;
        acc = (acc >> 32) | ((u64)acc2 << 32);
        acc2 = 0;
    }
    r[NA + NB - 1] = (u32)acc;
#else
    P2E_UNROLL
    for (int i = 0; i < NA + NB; i++) r[i] = 0;
    P2E_UNROLL
    for (int i = 0; i < NA; i++) {
        u64 c = 0;
        P2E_UNROLL
        for (int j = 0; j < NB; j++) {
            u64 t = (u64)a[i] * b[j] + r[i + j] + c;
            r[i + j] = (u32)t;
            c = t >> 32;
        }
        r[i + NB] = (u32)c;
    }
#endif
}

// r[16] = a[8]^2: the 28 cross products once (96-bit column accumulator as in mul_wide), the whole
// 512-bit cross sum doubled with one funnel-shift per word, then the 8 squares added on the even columns.
// 36 v_mad_u64_u32 instead of 64.
P2E_HD void sqr_wide8(const u32* a, u32* r) {
#if defined(__HIP_DEVICE_COMPILE__)
    u32 x[16];
    u64 acc = 0;
    u32 acc2 = 0;
    x[0] = 0;
    P2E_UNROLL
    for (int k = 1; k < 15; k++) {
        P2E_UNROLL
        for (int i = 0; i < 8; i++) {
            const int j = k - i;
            if (j > i && j < 8) {
                asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc"
                    : "+v"(acc), "+v"(acc2)
                    : "v"(a[i]), "v"(a[j])
                    : "vcc");
            }
        }
        x[k] = (u32)acc;
        acc = (acc >> 32) | ((u64)acc2 << 32);
        acc2 = 0;
    }
    x[15] = (u32)acc;
    // double (the cross sum is < 2^511, nothing is shifted out)
    P2E_UNROLL
    for (int k = 15; k > 0; k--) x[k] = (x[k] << 1) | (x[k - 1] >> 31);
    x[0] = 0;
    // + sum_i a_i^2 * 2^(64 i)
    u32 carry = 0;
    P2E_UNROLL
    for (int i = 0; i < 8; i++) {
        u64 sq = (u64)a[i] * a[i];
        r[2 * i] = addc32(x[2 * i], (u32)sq, carry);
        r[2 * i + 1] = addc32(x[2 * i + 1], (u32)(sq >> 32), carry);
    }
#else
    mul_wide<8, 8>(a, a, r);
#endif
}

template <int N>
P2E_HD u32 add_n(u32* r, const u32* a, const u32* b) {
    u32 c = 0;
    P2E_UNROLL
    for (int i = 0; i < N; i++) r[i] = addc32(a[i], b[i], c);
    return c;
}
template <int N>
P2E_HD u32 sub_n(u32* r, const u32* a, const u32* b) {  // returns borrow
    u32 br = 0;
    P2E_UNROLL
    for (int i = 0; i < N; i++) r[i] = subb32(a[i], b[i], br);
    return br;
}
template <int N>
P2E_HD bool geq_n(const u32* a, const u32* b) {  // a >= b
    u32 br = 0;
    P2E_UNROLL
    for (int i = 0; i < N; i++) (void)subb32(a[i], b[i], br);
    return br == 0;
}
template <int N>
P2E_HD bool is_zero_n(const u32* a) {
    u32 o = 0;
    P2E_UNROLL
    for (int i = 0; i < N; i++) o |= a[i];
    return o == 0;
}
template <class MOD>
P2E_HD bool geq_mod(const u32* a) {
    u32 br = 0;
    P2E_UNROLL
    for (int i = 0; i < 8; i++) (void)subb32(a[i], MOD::m(i), br);
    return br == 0;
}
template <class MOD>
P2E_HD void sub_mod_raw(u32* a) {  // a -= m
    u32 br = 0;
    P2E_UNROLL
    for (int i = 0; i < 8; i++) a[i] = subb32(a[i], MOD::m(i), br);
}
template <class MOD>
P2E_HD void add_mod_raw(u32* a) {  // a += m (mod 2^256)
    u32 c = 0;
    P2E_UNROLL
    for (int i = 0; i < 8; i++) a[i] = addc32(a[i], MOD::m(i), c);
}

P2E_HD U256 u256_zero() {
    U256 r;
    P2E_UNROLL
    for (int i = 0; i < 8; i++) r.w[i] = 0;
    return r;
}
P2E_HD U256 u256_small(u32 v) {
    U256 r = u256_zero();
    r.w[0] = v;
    return r;
}
P2E_HD bool u256_eq(const U256& a, const U256& b) {
    u32 o = 0;
    P2E_UNROLL
    for (int i = 0; i < 8; i++) o |= a.w[i] ^ b.w[i];
    return o == 0;
}
P2E_HD bool u256_is_zero(const U256& a) { return is_zero_n<8>(a.w); }
// word-wise select.  (A struct-level `c ? t : f` makes hipcc select between two ADDRESSES and read the
// winner back from scratch memory.)
P2E_HD U256 u256_select(bool c, const U256& t, const U256& f) {
    U256 r;
    P2E_UNROLL
    for (int i = 0; i < 8; i++) r.w[i] = c ? t.w[i] : f.w[i];
    return r;
}

// one fold step: (hi[NH], lo[8]) -> t = hi*C + lo ; lo = t mod 2^256 ; hi_out[NHO] = t >> 256 ;
// q += (t >> 256)
template <class MOD, int NH, int NHO, bool WANT_Q>
P2E_HD void fold_step(const u32* hi, u32* lo, u32* hi_out, u32* q /*9 words*/) {
    constexpr int NT = NH + MOD::NC + 1;
    u32 t[NT];
    u32 cw[MOD::NC];
    P2E_UNROLL
    for (int i = 0; i < MOD::NC; i++) cw[i] = MOD::c(i);
    mul_wide<NH, MOD::NC>(hi, cw, t);
    t[NT - 1] = 0;
    u64 c = 0;
    P2E_UNROLL
    for (int i = 0; i < NT; i++) {
        c += (u64)t[i] + (i < 8 ? lo[i] : 0u);
        t[i] = (u32)c;
        c >>= 32;
    }
    if (NT < 8) {  // NH + NC + 1 < 8 : remaining low words keep lo + carry
        P2E_UNROLL
        for (int i = NT; i < 8; i++) {
            c += (u64)lo[i];
            lo[i] = (u32)c;
            c >>= 32;
        }
        P2E_UNROLL
        for (int i = 0; i < NT; i++) lo[i] = t[i];
        P2E_UNROLL
        for (int k = 0; k < NHO; k++) hi_out[k] = k == 0 ? (u32)c : 0u;
    } else {
        P2E_UNROLL
        for (int i = 0; i < 8; i++) lo[i] = t[i];
        P2E_UNROLL
        for (int k = 0; k < NHO; k++) hi_out[k] = (8 + k < NT) ? t[8 + k] : 0u;
    }
    if (WANT_Q) {
        u64 cq = 0;
        P2E_UNROLL
        for (int i = 0; i < 9; i++) {
            cq += (u64)q[i] + (i < NHO ? hi_out[i] : 0u);
            q[i] = (u32)cq;
            cq >>= 32;
        }
    }
}

// Specialised reduction for p = 2^256 - C, C = 2^32 + 977, of a 16-word product: hi*C is one
// v_mad_u64_u32 chain by 977 plus a one-word shift, everything else is add-with-carry chains, and the
// final canonicalisation is "add C, keep the sum if it carries out of 2^256" (r >= p <=> r + C >= 2^256).
// Bounds as in DESIGN.md section 3.
template <bool WANT_Q>
P2E_HD void reduce_p16(const u32* prod, u32* r, u32* q /*9 or null*/) {
    const u32* lo = prod;
    const u32* hi = prod + 8;
    // u[0..8] = hi * 977
    u32 u[9];
    u64 m = 0;
    P2E_UNROLL
    for (int i = 0; i < 8; i++) {
        m = (u64)hi[i] * 977u + (m >> 32);
        u[i] = (u32)m;
    }
    u[8] = (u32)(m >> 32);            // < 2^10
    // t = lo + u[0..7] + (hi << 32)
    u32 t[8];
    u32 ca = 0, cb = 0;
    P2E_UNROLL
    for (int i = 0; i < 8; i++) t[i] = addc32(lo[i], u[i], ca);
    P2E_UNROLL
    for (int i = 1; i < 8; i++) t[i] = addc32(t[i], hi[i - 1], cb);
    // h1 = u[8] + hi[7] + ca + cb   (< 2^33 + 2^10 + 2)
    u32 ch = 0;
    u32 h1lo = addc32(u[8], hi[7], ch);
    u32 h1hi = ch;
    ch = 0;
    h1lo = addc32(h1lo, ca + cb, ch);
    h1hi += ch;
    // fold 2: t += h1 * 977 + (h1 << 32)
    const u64 v = (u64)h1lo * 977u + ((u64)(h1hi * 977u) << 32);   // < 2^44
    u32 c2 = 0;
    t[0] = addc32(t[0], (u32)v, c2);
    t[1] = addc32(t[1], (u32)(v >> 32), c2);
    P2E_UNROLL
    for (int i = 2; i < 8; i++) t[i] = addc32(t[i], 0u, c2);
    u32 c2b = 0;
    t[1] = addc32(t[1], h1lo, c2b);
    t[2] = addc32(t[2], h1hi, c2b);
    P2E_UNROLL
    for (int i = 3; i < 8; i++) t[i] = addc32(t[i], 0u, c2b);
    const u32 h2 = c2 + c2b;          // 0 or 1; if 1 then t < 2^78 and adding C cannot carry again
    u32 c3 = 0;
    t[0] = addc32(t[0], h2 ? 977u : 0u, c3);
    t[1] = addc32(t[1], h2, c3);
    P2E_UNROLL
    for (int i = 2; i < 8; i++) t[i] = addc32(t[i], 0u, c3);
    // canonical: w = t + C ; carry out <=> t >= p
    u32 w[8];
    u32 c4 = 0;
    w[0] = addc32(t[0], 977u, c4);
    w[1] = addc32(t[1], 1u, c4);
    P2E_UNROLL
    for (int i = 2; i < 8; i++) w[i] = addc32(t[i], 0u, c4);
    const bool ge = c4 != 0;
    P2E_UNROLL
    for (int i = 0; i < 8; i++) r[i] = ge ? w[i] : t[i];
    if (WANT_Q) {   // q = hi + h1 + h2 + ge
        u32 cq = 0;
        q[0] = addc32(hi[0], h1lo, cq);
        q[1] = addc32(hi[1], h1hi, cq);
        P2E_UNROLL
        for (int i = 2; i < 8; i++) q[i] = addc32(hi[i], 0u, cq);
        q[8] = cq;
        u32 cq2 = 0;
        q[0] = addc32(q[0], h2 + (ge ? 1u : 0u), cq2);
        P2E_UNROLL
        for (int i = 1; i < 9; i++) q[i] = addc32(q[i], 0u, cq2);
    }
}

// prod = hi:lo with NH high words.  Returns canonical r = prod mod m and (optionally) the exact
// integer quotient q = floor(prod / m) in 9 words.  Fold schedule (hi word counts): p: NH,2,1 ;
// n: NH,5,1,1 -- bounds derived in DESIGN.md "Field arithmetic".
template <class MOD, int NH, bool WANT_Q>
P2E_HD void reduce_wide(const u32* prod /*8+NH*/, u32* r /*8*/, u32* q /*9 or null*/) {
    u32 lo[8];
    P2E_UNROLL
    for (int i = 0; i < 8; i++) lo[i] = prod[i];
    if (WANT_Q) {
        P2E_UNROLL
        for (int i = 0; i < 9; i++) q[i] = i < NH ? prod[8 + i] : 0u;
    }
    u32 h1[MOD::NC];
    fold_step<MOD, NH, MOD::NC, WANT_Q>(prod + 8, lo, h1, q);
    u32 h2[1];
    fold_step<MOD, MOD::NC, 1, WANT_Q>(h1, lo, h2, q);
    u32 h3[1];
    fold_step<MOD, 1, 1, WANT_Q>(h2, lo, h3, q);
    if (MOD::kFourFolds) {
        u32 h4[1];
        fold_step<MOD, 1, 1, WANT_Q>(h3, lo, h4, q);
    }
    if (geq_mod<MOD>(lo)) {
        sub_mod_raw<MOD>(lo);
        if (WANT_Q) {
            u64 cq = 1;
            P2E_UNROLL
            for (int i = 0; i < 9; i++) {
                cq += q[i];
                q[i] = (u32)cq;
                cq >>= 32;
            }
        }
    }
    P2E_UNROLL
    for (int i = 0; i < 8; i++) r[i] = lo[i];
}

// low NR words of a[NA] * b[NB] (product scanning, only the columns below NR)
template <int NA, int NB, int NR>
P2E_HD void mul_lo(const u32* a, const u32* b, u32* r) {
#if defined(__HIP_DEVICE_COMPILE__)
    u64 acc = 0;
    u32 acc2 = 0;
    P2E_UNROLL
    for (int k = 0; k < NR; k++) {
        P2E_UNROLL
        for (int i = 0; i < NA; i++) {
            const int j = k - i;
            if (j >= 0 && j < NB) {
                asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc"
                    : "+v"(acc), "+v"(acc2)
                    : "v"(a[i]), "v"(b[j])
                    : "vcc");
            }
        }
        r[k] = (u32)acc;
        acc = (acc >> 32) | ((u64)acc2 << 32);
        acc2 = 0;
    }
#else
    P2E_UNROLL
    for (int i = 0; i < NR; i++) r[i] = 0;
    P2E_UNROLL
    for (int i = 0; i < NA; i++) {
        u64 c = 0;
        P2E_UNROLL
        for (int j = 0; j < NB; j++) {
            if (i + j < NR) {
                u64 t = (u64)a[i] * b[j] + r[i + j] + c;
                r[i + j] = (u32)t;
                c = t >> 32;
            }
        }
        if (i + NB < NR) r[i + NB] = (u32)c;
    }
#endif
}

// Barrett reduction (HAC 14.42 with b = 2^32, k = 8) of prod = hi:lo with NH high words, for any modulus with its top
// word set: q1 = prod >> 224, q3 = (q1 * mu) >> 288 with mu = floor(2^512 / m), then q3 <= floor(prod / m) <= q3 + 2 and
// r = (prod - q3 * m) mod 2^288 needs at most two subtractions of m.  Also yields the exact integer quotient the
// MulNonnative / Inverse witness generators need.  (64 + (NH+1)*9 + ~44 multiplications for NH = 8, against 73 for
// the fold chain of secp256k1's p: the price of a modulus without the 2^256 - small shape.)
template <class MOD, int NH, bool WANT_Q>
P2E_HD void reduce_barrett(const u32* prod /*8+NH*/, u32* r /*8*/, u32* q /*9 or null*/) {
    constexpr int NQ = NH + 1;
    u32 muw[9], mw[8];
    P2E_UNROLL
    for (int i = 0; i < 9; i++) muw[i] = MOD::mu(i);
    P2E_UNROLL
    for (int i = 0; i < 8; i++) mw[i] = MOD::m(i);
    u32 q2[NQ + 9];
    mul_wide<NQ, 9>(prod + 7, muw, q2);
    const u32* q3 = q2 + 9;   // NQ words
    u32 r2[9];
    mul_lo<NQ, 8, 9>(q3, mw, r2);
    u32 t[9];
    u32 br = 0;
    P2E_UNROLL
    for (int i = 0; i < 9; i++) t[i] = subb32(i < 8 + NH ? prod[i] : 0u, r2[i], br);
    u32 extra = 0;
    P2E_UNROLL
    for (int pass = 0; pass < 2; pass++) {
        const bool ge = t[8] != 0 || geq_mod<MOD>(t);
        u32 b2 = 0;
        P2E_UNROLL
        for (int i = 0; i < 9; i++) {
            const u32 d = subb32(t[i], i < 8 ? mw[i] : 0u, b2);
            t[i] = ge ? d : t[i];
        }
        extra += ge ? 1u : 0u;
    }
    P2E_UNROLL
    for (int i = 0; i < 8; i++) r[i] = t[i];
    if (WANT_Q) {
        u32 c = 0;
        P2E_UNROLL
        for (int i = 0; i < 9; i++) q[i] = addc32(i < NQ ? q3[i] : 0u, i == 0 ? extra : 0u, c);
    }
}

// The same for products beyond 2^512 (the raw MulNonnative entry point takes 261-bit operands: NH = 10).  There mu's
// rounding error is no longer below one: the first estimate can be short by up to 2^(32 NH - 256) multiples of m.  So:
// one estimate WITHOUT the final corrections -- the partial remainder R = prod - q3 m is below (2^(32 NH - 256) + 3) m and
// fits 9 words -- then the plain reduction of R (a 9-word value: its own estimate is within two of the truth).
template <class MOD, int NH, bool WANT_Q>
P2E_HD void reduce_barrett_wide(const u32* prod /*8+NH*/, u32* r /*8*/, u32* q /*9 or null*/) {
    static_assert(NH > 8 && NH <= 10, "products of operands up to 9 words");
    constexpr int NQ = NH + 1;
    u32 muw[9], mw[8];
    P2E_UNROLL
    for (int i = 0; i < 9; i++) muw[i] = MOD::mu(i);
    P2E_UNROLL
    for (int i = 0; i < 8; i++) mw[i] = MOD::m(i);
    u32 q2[NQ + 9];
    mul_wide<NQ, 9>(prod + 7, muw, q2);
    const u32* q3 = q2 + 9;   // NQ words; only the low 9 can be non-zero when the true quotient fits 9 words
    u32 r2[9];
    mul_lo<NQ, 8, 9>(q3, mw, r2);
    u32 t[9];
    u32 br = 0;
    P2E_UNROLL
    for (int i = 0; i < 9; i++) t[i] = subb32(prod[i], r2[i], br);
    u32 qe[9];
    reduce_barrett<MOD, 1, WANT_Q>(t, r, WANT_Q ? qe : nullptr);
    if (WANT_Q) {
        u32 c = 0;
        P2E_UNROLL
        for (int i = 0; i < 9; i++) q[i] = addc32(q3[i], qe[i], c);
        // a quotient beyond 9 words (q3's upper words, or the carry) cannot be emitted: saturate so that the caller's
        // range check on q flags the element
        u32 hi = c;
        P2E_UNROLL
        for (int i = 9; i < NQ; i++) hi |= q3[i];
        if (hi) {
            P2E_UNROLL
            for (int i = 0; i < 9; i++) q[i] = 0xFFFFFFFFu;
        }
    }
}

// 16-word product -> (r, q): specialised path for p, generic fold chain for n
template <class MOD, bool WANT_Q>
P2E_HD void reduce16(const u32* prod, u32* r, u32* q);
template <>
P2E_HD void reduce16<ModP, false>(const u32* prod, u32* r, u32* q) { reduce_p16<false>(prod, r, q); }
template <>
P2E_HD void reduce16<ModP, true>(const u32* prod, u32* r, u32* q) { reduce_p16<true>(prod, r, q); }
template <>
P2E_HD void reduce16<ModN, false>(const u32* prod, u32* r, u32* q) { reduce_wide<ModN, 8, false>(prod, r, q); }
template <>
P2E_HD void reduce16<ModN, true>(const u32* prod, u32* r, u32* q) { reduce_wide<ModN, 8, true>(prod, r, q); }
// x mod p for P-256's p = 2^256 - 2^224 + 2^192 + 2^96 - 1 without a multiplication: the NIST fast reduction (FIPS 186-4
// D.2.3), r = s1 + 2 s2 + 2 s3 + s4 + s5 - s6 - s7 - s8 - s9 over the 32-bit words c0 .. c15 of x, written out per
// result word as signed word sums.  The sum lies in (-5p, 5p): its signed overflow `top` is folded back twice through
// 2^256 = p + (2^224 - 2^192 - 2^96 + 1) -- |top| <= 7 leaves at most +-1 after the first fold and nothing after the
// second -- and one conditional subtraction canonicalises.  No quotient comes out of it: the chains and inversions use
// this form (fe_mul / fe_sqr), the witness generators keep reduce_barrett, which yields floor(x / p) as well.
P2E_HD void reduce_p256_solinas(const u32* c /*16*/, u32* r /*8*/) {
    i64 s[8];
    s[0] = (i64)c[0] + c[8] + c[9] - c[11] - c[12] - c[13] - c[14];
    s[1] = (i64)c[1] + c[9] + c[10] - c[12] - c[13] - c[14] - c[15];
    s[2] = (i64)c[2] + c[10] + c[11] - c[13] - c[14] - c[15];
    s[3] = (i64)c[3] + 2 * ((i64)c[11] + c[12]) + c[13] - c[15] - c[8] - c[9];
    s[4] = (i64)c[4] + 2 * ((i64)c[12] + c[13]) + c[14] - c[9] - c[10];
    s[5] = (i64)c[5] + 2 * ((i64)c[13] + c[14]) + c[15] - c[10] - c[11];
    s[6] = (i64)c[6] + 3 * (i64)c[14] + 2 * (i64)c[15] + c[13] - c[8] - c[9];
    s[7] = (i64)c[7] + 3 * (i64)c[15] + c[8] - c[10] - c[11] - c[12] - c[13];
    u32 t[8];
    i64 k = 0;
    P2E_UNROLL
    for (int i = 0; i < 8; i++) {
        k += s[i];
        t[i] = (u32)k;
        k >>= 32;   // arithmetic: the running carry is signed
    }
    P2E_UNROLL
    for (int pass = 0; pass < 2; pass++) {   // value = t + k * 2^256 == t + k * (2^224 - 2^192 - 2^96 + 1)  (mod p)
        const i64 top = k;
        k = 0;
        P2E_UNROLL
        for (int i = 0; i < 8; i++) {
            k += (i64)t[i] + ((i == 0 || i == 7) ? top : (i == 3 || i == 6) ? -top : 0);
            t[i] = (u32)k;
            k >>= 32;
        }
    }
    // now 0 <= t < 2^256 < 2p
    u32 w[8];
    u32 br = 0;
    P2E_UNROLL
    for (int i = 0; i < 8; i++) w[i] = subb32(t[i], ModP256::m(i), br);
    P2E_UNROLL
    for (int i = 0; i < 8; i++) r[i] = br ? t[i] : w[i];
}
template <>
P2E_HD void reduce16<ModP256, false>(const u32* prod, u32* r, u32* q) {
    (void)q;
    reduce_p256_solinas(prod, r);
}
template <>
P2E_HD void reduce16<ModP256, true>(const u32* prod, u32* r, u32* q) { reduce_barrett<ModP256, 8, true>(prod, r, q); }
template <>
P2E_HD void reduce16<ModN256, false>(const u32* prod, u32* r, u32* q) { reduce_barrett<ModN256, 8, false>(prod, r, q); }
template <>
P2E_HD void reduce16<ModN256, true>(const u32* prod, u32* r, u32* q) { reduce_barrett<ModN256, 8, true>(prod, r, q); }

// ------------------------------------------------------------------------------------------------
// field ops on canonical values
// ------------------------------------------------------------------------------------------------
// The multiply is the one function that is NOT force-inlined on the device: the Jacobian chains and
// the Fermat ladders call it hundreds of times and an inlined copy is ~400 instructions, which would
// blow the instruction cache.  Arguments/results travel by value in VGPRs.
template <class MOD>
P2E_HD_NOINLINE U256 fe_mul_call(U256 a, U256 b) {
    u32 prod[16];
    mul_wide<8, 8>(a.w, b.w, prod);
    U256 r;
    reduce16<MOD, false>(prod, r.w, nullptr);
    return r;
}
template <class MOD>
P2E_HD U256 fe_mul(const U256& a, const U256& b) {
    return fe_mul_call<MOD>(a, b);
}
template <class MOD>
P2E_HD_NOINLINE U256 fe_sqr_call(U256 a) {
    u32 prod[16];
    sqr_wide8(a.w, prod);
    U256 r;
    reduce16<MOD, false>(prod, r.w, nullptr);
    return r;
}
template <class MOD>
P2E_HD U256 fe_sqr(const U256& a) {
    return fe_sqr_call<MOD>(a);
}
template <class MOD>
P2E_HD U256 fe_add(const U256& a, const U256& b) {
    U256 r;
    u32 c = add_n<8>(r.w, a.w, b.w);
    if (c || geq_mod<MOD>(r.w)) sub_mod_raw<MOD>(r.w);
    return r;
}
template <class MOD>
P2E_HD U256 fe_sub(const U256& a, const U256& b) {
    U256 r;
    if (sub_n<8>(r.w, a.w, b.w)) add_mod_raw<MOD>(r.w);
    return r;
}
template <class MOD>
P2E_HD U256 fe_neg(const U256& a) {
    return fe_sub<MOD>(u256_zero(), a);
}
// to_canonical_biguint of the reference's field types: ONE conditional subtraction
template <class MOD>
P2E_HD U256 fe_canon(const U256& a) {
    U256 r = a;
    if (geq_mod<MOD>(r.w)) sub_mod_raw<MOD>(r.w);
    return r;
}
template <class MOD>
P2E_HD U256 fe_sqr_n(U256 a, int n) {
    for (int i = 0; i < n; i++) a = fe_sqr<MOD>(a);
    return a;
}
// a^(p-2) mod p: 255 squarings + 15 multiplications (addition chain over the run structure of p-2)
P2E_HD U256 fe_inv_p(const U256& a) {
    U256 x2 = fe_mul<ModP>(fe_sqr<ModP>(a), a);
    U256 x3 = fe_mul<ModP>(fe_sqr<ModP>(x2), a);
    U256 x6 = fe_mul<ModP>(fe_sqr_n<ModP>(x3, 3), x3);
    U256 x9 = fe_mul<ModP>(fe_sqr_n<ModP>(x6, 3), x3);
    U256 x11 = fe_mul<ModP>(fe_sqr_n<ModP>(x9, 2), x2);
    U256 x22 = fe_mul<ModP>(fe_sqr_n<ModP>(x11, 11), x11);
    U256 x44 = fe_mul<ModP>(fe_sqr_n<ModP>(x22, 22), x22);
    U256 x88 = fe_mul<ModP>(fe_sqr_n<ModP>(x44, 44), x44);
    U256 x176 = fe_mul<ModP>(fe_sqr_n<ModP>(x88, 88), x88);
    U256 x220 = fe_mul<ModP>(fe_sqr_n<ModP>(x176, 44), x44);
    U256 x223 = fe_mul<ModP>(fe_sqr_n<ModP>(x220, 3), x3);
    U256 t = fe_mul<ModP>(fe_sqr_n<ModP>(x223, 23), x22);
    t = fe_mul<ModP>(fe_sqr_n<ModP>(t, 5), a);
    t = fe_mul<ModP>(fe_sqr_n<ModP>(t, 3), x2);
    t = fe_mul<ModP>(fe_sqr_n<ModP>(t, 2), a);
    return t;
}
// a^(n-2) mod n: plain square-and-multiply.  The exponent is a compile-time constant, so the branch
// is wave-uniform (no divergence, no table in scratch); it runs once per signature (s^-1).
P2E_HD U256 fe_inv_n(const U256& a) {
    U256 acc = a;  // top bit of n-2 is set
    for (int i = 254; i >= 0; i--) {
        acc = fe_sqr<ModN>(acc);
        u32 word = ModN::m(i >> 5);
        if ((i >> 5) == 0) word -= 2;  // low word of n is 0xD0364141: no borrow
        if ((word >> (i & 31)) & 1) acc = fe_mul<ModN>(acc, a);
    }
    return acc;
}
// ------------------------------------------------------------------------------------------------
// modular inversion by binary GCD with 64-bit approximations (T. Pornin, "Optimized Binary GCD for
// Modular Inversion", 2020; k = 32: 17 outer rounds of 31 branch-free inner steps on 64-bit
// approximations of (a, b), then one exact linear update of the full-width values).  ~14 k full-rate
// VALU instructions and ~600 mads instead of the 66 k instructions / 19.7 k mads of a Fermat ladder.
// Invariants: a * 2^(31 i) = y * u, b * 2^(31 i) = y * v (mod m); after 17 rounds a = 0, b = gcd = 1 and
// y^-1 = v * 2^(-527).  Every step is a select, so lanes do not diverge.  If the round bound ever failed
// (b != 1; never observed, ~10^8 inputs tested incl. the structured ones) the caller falls back to Fermat.
// ------------------------------------------------------------------------------------------------
template <class MOD>
P2E_HD U256 fe_mul_small(const U256& x, u32 f) {   // x * f mod m, x < 2^256
    u32 t[9];
    u64 c = 0;
    P2E_UNROLL
    for (int i = 0; i < 8; i++) {
        c += (u64)x.w[i] * f;
        t[i] = (u32)c;
        c >>= 32;
    }
    t[8] = (u32)c;
    U256 r;
    if constexpr (MOD::kBarrett)
        reduce_barrett<MOD, 1, false>(t, r.w, nullptr);
    else
        reduce_wide<MOD, 1, false>(t, r.w, nullptr);
    return r;
}
// |fa * a  +-  fb * b| >> 31 for magnitudes fa, fb <= 2^31 and signs na, nb (true = negative); the sum is an
// exact multiple of 2^31 and its magnitude fits 256 bits.  Returns the result's sign.
P2E_HD bool gcd_lincomb(const u32* a, u32 fa, bool na, const u32* b, u32 fb, bool nb, u32* out) {
    u32 p[10], q[10];
    u64 c = 0, d = 0;
    P2E_UNROLL
    for (int i = 0; i < 8; i++) {
        c += (u64)a[i] * fa;
        p[i] = (u32)c;
        c >>= 32;
        d += (u64)b[i] * fb;
        q[i] = (u32)d;
        d >>= 32;
    }
    p[8] = (u32)c;
    p[9] = 0;
    q[8] = (u32)d;
    q[9] = 0;
    // s = (+-p) + (+-q) in 10-word two's complement: negate by (x ^ mask) + carry-in
    const u32 mp = na ? 0xFFFFFFFFu : 0u, mq = nb ? 0xFFFFFFFFu : 0u;
    u32 cp = na ? 1u : 0u, cq = nb ? 1u : 0u, cs = 0;
    u32 s[10];
    P2E_UNROLL
    for (int i = 0; i < 10; i++) {
        u32 zero = 0;
        u32 pi = addc32(p[i] ^ mp, 0u, cp);
        u32 qi = addc32(q[i] ^ mq, 0u, cq);
        (void)zero;
        s[i] = addc32(pi, qi, cs);
    }
    const bool neg = (s[9] >> 31) != 0;
    const u32 ms = neg ? 0xFFFFFFFFu : 0u;
    u32 cn = neg ? 1u : 0u;
    P2E_UNROLL
    for (int i = 0; i < 10; i++) s[i] = addc32(s[i] ^ ms, 0u, cn);
    P2E_UNROLL
    for (int i = 0; i < 8; i++) out[i] = (s[i] >> 31) | (s[i + 1] << 1);
    return neg;
}
template <class MOD>
P2E_HD bool fe_inv_bingcd(const U256& y, U256& result) {
    u32 a[8], b[8];
    U256 u = u256_small(1), v = u256_zero();
    P2E_UNROLL
    for (int i = 0; i < 8; i++) {
        a[i] = y.w[i];
        b[i] = MOD::m(i);
    }
    for (int round = 0; round < 17; round++) {
        // 64-bit approximations: low 31 bits + the 33 bits below the top of max(len a, len b, 64)
        u32 a2 = a[7], a1 = a[6], a0 = a[5], b2 = b[7], b1 = b[6], b0 = b[5];
        u32 ctop = a[7] | b[7];
        P2E_UNROLL
        for (int j = 6; j >= 2; j--) {
            const bool found = ctop != 0;
            a2 = found ? a2 : a[j];
            a1 = found ? a1 : a[j - 1];
            a0 = found ? a0 : a[j - 2];
            b2 = found ? b2 : b[j];
            b1 = found ? b1 : b[j - 1];
            b0 = found ? b0 : b[j - 2];
            ctop = found ? ctop : (a[j] | b[j]);
        }
        u64 ah, bh;
        if (ctop != 0) {
#if defined(__HIP_DEVICE_COMPILE__)
            const int sh = __clz((int)ctop);
#else
            const int sh = __builtin_clz(ctop);
#endif
            ah = (((u64)a2 << 32) | a1) << sh;
            bh = (((u64)b2 << 32) | b1) << sh;
            if (sh) {
                ah |= (u64)(a0 >> (32 - sh));
                bh |= (u64)(b0 >> (32 - sh));
            }
        } else {
            ah = ((u64)a[1] << 32) | a[0];
            bh = ((u64)b[1] << 32) | b[0];
        }
        u64 xa = ((ah >> 31) << 31) | (a[0] & 0x7FFFFFFFu);
        u64 xb = ((bh >> 31) << 31) | (b[0] & 0x7FFFFFFFu);
        i64 f0 = 1, g0 = 0, f1 = 0, g1 = 1;
        for (int i = 0; i < 31; i++) {
            const bool odd = (xa & 1) != 0;
            const bool swap = odd && (xa < xb);
            const u64 ta = swap ? xb : xa, tb = swap ? xa : xb;
            const i64 tf0 = swap ? f1 : f0, tf1 = swap ? f0 : f1, tg0 = swap ? g1 : g0, tg1 = swap ? g0 : g1;
            xa = (odd ? ta - tb : ta) >> 1;
            xb = tb;
            f0 = odd ? tf0 - tf1 : tf0;
            g0 = odd ? tg0 - tg1 : tg0;
            f1 = tf1 * 2;   // (not << 1: shifting a negative value is undefined before C++20; |f|, |g| <= 2^31)
            g1 = tg1 * 2;
        }
        u32 na_[8], nb_[8];
        const bool sf0 = f0 < 0, sg0 = g0 < 0, sf1 = f1 < 0, sg1 = g1 < 0;
        const u32 mf0 = (u32)(sf0 ? -f0 : f0), mg0 = (u32)(sg0 ? -g0 : g0);
        const u32 mf1 = (u32)(sf1 ? -f1 : f1), mg1 = (u32)(sg1 ? -g1 : g1);
        const bool nega = gcd_lincomb(a, mf0, sf0, b, mg0, sg0, na_);
        const bool negb = gcd_lincomb(a, mf1, sf1, b, mg1, sg1, nb_);
        P2E_UNROLL
        for (int i = 0; i < 8; i++) {
            a[i] = na_[i];
            b[i] = nb_[i];
        }
        // (u, v) <- (f0 u + g0 v, f1 u + g1 v) mod m, with the signs of the rows whose a / b came out negative flipped
        U256 uf0 = fe_mul_small<MOD>(u, mf0), vg0 = fe_mul_small<MOD>(v, mg0);
        U256 uf1 = fe_mul_small<MOD>(u, mf1), vg1 = fe_mul_small<MOD>(v, mg1);
        uf0 = u256_select(sf0 != nega, fe_neg<MOD>(uf0), uf0);
        vg0 = u256_select(sg0 != nega, fe_neg<MOD>(vg0), vg0);
        uf1 = u256_select(sf1 != negb, fe_neg<MOD>(uf1), uf1);
        vg1 = u256_select(sg1 != negb, fe_neg<MOD>(vg1), vg1);
        u = fe_add<MOD>(uf0, vg0);
        v = fe_add<MOD>(uf1, vg1);
    }
    // 2^-527 mod m
    U256 k;   // 2^-527 mod m
    P2E_UNROLL
    for (int i = 0; i < 8; i++) k.w[i] = MOD::k527(i);
    result = fe_mul<MOD>(v, k);
    bool ok = b[0] == 1;
    P2E_UNROLL
    for (int i = 1; i < 8; i++) ok = ok && b[i] == 0;
    P2E_UNROLL
    for (int i = 0; i < 8; i++) ok = ok && a[i] == 0;
    return ok;
}

// ------------------------------------------------------------------------------------------------
// modular inversion by "safegcd" (D. J. Bernstein, B.-Y. Yang, "Fast constant-time gcd computation and modular
// inversion", 2019) in the 32-bit formulation published with libsecp256k1 (doc/safegcd_implementation.md: signed 30-bit
// limbs, 30 division steps at a time on the low limbs alone, then ONE exact update of the full-width f, g and d, e by the
// 2 x 2 transition matrix).  20 x 30 = 600 steps cover any 256-bit modulus (590 needed).  Every step is mask
// arithmetic on 32-bit values -- no 64-bit compares, no carry chains through VCC -- which is what makes it 2.2 x faster
// than fe_inv_bingcd above for a lone wave (35 against 77 us, tools/ubench/fe29_latency.hip): the inversion is on the
// critical path of every call twice over (s^-1 in the scalar phase, the window table's inversion batch).
// Returns false if gcd(y, m) != 1, i.e. for y = 0 (the caller's Fermat ladder then yields the reference's 0).
// ------------------------------------------------------------------------------------------------
template <class MOD>
P2E_HD bool fe_inv_safegcd(const U256& y, U256& result) {
    typedef int32_t i32;
    const i32 M30 = (i32)((1u << 30) - 1u);
    i32 m[9], d[9], e[9], f[9], g[9];
    {   // 8 x 32-bit words -> 9 x 30-bit limbs
        u32 mw[9], yw[9];
        P2E_UNROLL
        for (int i = 0; i < 8; i++) {
            mw[i] = MOD::m(i);
            yw[i] = y.w[i];
        }
        mw[8] = yw[8] = 0;
        P2E_UNROLL
        for (int i = 0; i < 9; i++) {
            const int bit = 30 * i, wi = bit >> 5, sh = bit & 31;
            u32 lm = mw[wi] >> sh, ly = yw[wi] >> sh;
            if (sh > 2 && wi + 1 < 9) {
                lm |= mw[wi + 1] << (32 - sh);
                ly |= yw[wi + 1] << (32 - sh);
            }
            m[i] = (i32)(lm & (u32)M30);
            g[i] = (i32)(ly & (u32)M30);
            f[i] = m[i];
            d[i] = 0;
            e[i] = i == 0 ? 1 : 0;
        }
    }
    u32 minv = (u32)m[0];   // m^-1 mod 2^30 (Newton: 5 doublings of the 3 bits every odd number is its own inverse to)
    P2E_UNROLL
    for (int k = 0; k < 5; k++) minv *= 2u - (u32)m[0] * minv;
    minv &= (u32)M30;
    i32 zeta = -1;   // -(delta + 1/2), delta = 1/2
    for (int it = 0; it < 20; it++) {
        // ---- 30 division steps on the low limbs: t = [u v; q r] with t [f; g] = 2^30 [f'; g']
        u32 u = 1, v = 0, q = 0, r = 1, ff = (u32)f[0], gg = (u32)g[0];
        for (int i = 0; i < 30; i++) {
            u32 c1 = (u32)(zeta >> 31);          // all ones if zeta < 0
            const u32 c2 = 0u - (gg & 1u);       // all ones if g is odd
            const u32 x = (ff ^ c1) - c1, yy = (u ^ c1) - c1, z = (v ^ c1) - c1;   // f, u, v negated if zeta < 0
            gg += x & c2;
            q += yy & c2;
            r += z & c2;
            c1 &= c2;
            zeta = (i32)(((u32)zeta ^ c1) - 1u);
            ff += gg & c1;
            u += q & c1;
            v += r & c1;
            gg >>= 1;
            u <<= 1;
            v <<= 1;
        }
        const i32 tu = (i32)u, tv = (i32)v, tq = (i32)q, tr = (i32)r;
        // ---- (d, e) <- t (d, e) / 2^30 mod m, kept in (-2 m, m)
        {
            const i32 sd = d[8] >> 31, se = e[8] >> 31;
            i32 md = (tu & sd) + (tv & se), me = (tq & sd) + (tr & se);
            i64 cd = (i64)tu * d[0] + (i64)tv * e[0], ce = (i64)tq * d[0] + (i64)tr * e[0];
            md -= (i32)((minv * (u32)cd + (u32)md) & (u32)M30);
            me -= (i32)((minv * (u32)ce + (u32)me) & (u32)M30);
            cd += (i64)m[0] * md;
            ce += (i64)m[0] * me;
            cd >>= 30;
            ce >>= 30;
            P2E_UNROLL
            for (int i = 1; i < 9; i++) {
                cd += (i64)tu * d[i] + (i64)tv * e[i];
                ce += (i64)tq * d[i] + (i64)tr * e[i];
                cd += (i64)m[i] * md;
                ce += (i64)m[i] * me;
                d[i - 1] = (i32)cd & M30;
                e[i - 1] = (i32)ce & M30;
                cd >>= 30;
                ce >>= 30;
            }
            d[8] = (i32)cd;
            e[8] = (i32)ce;
        }
        // ---- (f, g) <- t (f, g) / 2^30 (exact)
        {
            i64 cf = (i64)tu * f[0] + (i64)tv * g[0], cg = (i64)tq * f[0] + (i64)tr * g[0];
            cf >>= 30;
            cg >>= 30;
            P2E_UNROLL
            for (int i = 1; i < 9; i++) {
                cf += (i64)tu * f[i] + (i64)tv * g[i];
                cg += (i64)tq * f[i] + (i64)tr * g[i];
                f[i - 1] = (i32)cf & M30;
                g[i - 1] = (i32)cg & M30;
                cf >>= 30;
                cg >>= 30;
            }
            f[8] = (i32)cf;
            g[8] = (i32)cg;
        }
    }
    // g = 0 and f = +-1 unless gcd(y, m) != 1
    i32 gz = 0, f_one = f[0] ^ 1, f_minus = (f[0] ^ M30) | (f[8] ^ -1);
    P2E_UNROLL
    for (int i = 0; i < 9; i++) gz |= g[i];
    P2E_UNROLL
    for (int i = 1; i < 9; i++) f_one |= f[i];
    P2E_UNROLL
    for (int i = 1; i < 8; i++) f_minus |= f[i] ^ M30;
    const bool ok = gz == 0 && (f_one == 0 || f_minus == 0);
    // y^-1 = sign(f) d, brought into [0, m)
    {
        i32 add = d[8] >> 31;
        const i32 neg = f[8] >> 31;
        P2E_UNROLL
        for (int i = 0; i < 9; i++) d[i] = ((d[i] + (m[i] & add)) ^ neg) - neg;
        P2E_UNROLL
        for (int i = 0; i < 8; i++) {
            d[i + 1] += d[i] >> 30;
            d[i] &= M30;
        }
        add = d[8] >> 31;
        P2E_UNROLL
        for (int i = 0; i < 9; i++) d[i] += m[i] & add;
        P2E_UNROLL
        for (int i = 0; i < 8; i++) {
            d[i + 1] += d[i] >> 30;
            d[i] &= M30;
        }
    }
    P2E_UNROLL
    for (int j = 0; j < 8; j++) {   // 9 x 30-bit limbs -> 8 x 32-bit words
        const int bit = 32 * j, li = bit / 30, sh = bit - 30 * li;
        u32 w = (u32)d[li] >> sh;
        w |= (u32)d[li + 1] << (30 - sh);
        if (60 - sh < 32 && li + 2 < 9) w |= (u32)d[li + 2] << (60 - sh);
        result.w[j] = w;
    }
    return ok;
}

// a^(m-2) mod m by plain square-and-multiply (the low word of every modulus here is >= 2: no borrow).  Only the
// fallback of fe_inv for the moduli without a dedicated ladder, i.e. reached by zero inputs.
template <class MOD>
P2E_HD U256 fe_inv_fermat(const U256& a) {
    U256 acc = a;  // the top bit of m - 2 is set
    for (int i = 254; i >= 0; i--) {
        acc = fe_sqr<MOD>(acc);
        u32 word = MOD::m(i >> 5);
        if ((i >> 5) == 0) word -= 2;
        if ((word >> (i & 31)) & 1) acc = fe_mul<MOD>(acc, a);
    }
    return acc;
}
template <class MOD>
P2E_HD U256 fe_inv(const U256& a) {
    U256 r;
    if (fe_inv_safegcd<MOD>(a, r)) return r;
    return fe_inv_fermat<MOD>(a);
}
template <>
P2E_HD U256 fe_inv<ModP>(const U256& a) {
    U256 r;
    if (fe_inv_safegcd<ModP>(a, r)) return r;
    return fe_inv_p(a);   // a == 0: the reference's inverse of zero is a^(p-2) = 0
}
template <>
P2E_HD U256 fe_inv<ModN>(const U256& a) {
    U256 r;
    if (fe_inv_safegcd<ModN>(a, r)) return r;
    return fe_inv_n(a);
}

// ------------------------------------------------------------------------------------------------
// 29-bit limb split / pack (reference gadgets/biguint.rs:27-51 convert_base, :454-463 set_biguint_target)
// ------------------------------------------------------------------------------------------------
// funnel shift right: low 32 bits of (hi:lo) >> sh, 0 < sh < 32
P2E_HD u32 fsr(u32 lo, u32 hi, int sh) { return (lo >> sh) | (hi << (32 - sh)); }
// 29-bit limbs of an 8-word value.  Written with scalar operands only (no 64-bit pair built from two
// array elements): hipcc otherwise keeps the word array in scratch memory to read the pairs as unaligned
// 64-bit loads, and a scratch reload in the middle of a kernel waits (vmcnt is in-order) for every column
// store issued before it.
P2E_HD void split29(const U256& a, u32* l /*9*/) {
    l[0] = a.w[0] & MASK29;
    l[1] = fsr(a.w[0], a.w[1], 29) & MASK29;
    l[2] = fsr(a.w[1], a.w[2], 26) & MASK29;
    l[3] = fsr(a.w[2], a.w[3], 23) & MASK29;
    l[4] = fsr(a.w[3], a.w[4], 20) & MASK29;
    l[5] = fsr(a.w[4], a.w[5], 17) & MASK29;
    l[6] = fsr(a.w[5], a.w[6], 14) & MASK29;
    l[7] = fsr(a.w[6], a.w[7], 11) & MASK29;
    l[8] = a.w[7] >> 8;
}
// same for a 9-word value (the quotient of the witness multiplication, < 2^261)
P2E_HD void split29_9(const u32* w /*9*/, u32* l /*9*/) {
    l[0] = w[0] & MASK29;
    l[1] = fsr(w[0], w[1], 29) & MASK29;
    l[2] = fsr(w[1], w[2], 26) & MASK29;
    l[3] = fsr(w[2], w[3], 23) & MASK29;
    l[4] = fsr(w[3], w[4], 20) & MASK29;
    l[5] = fsr(w[4], w[5], 17) & MASK29;
    l[6] = fsr(w[5], w[6], 14) & MASK29;
    l[7] = fsr(w[6], w[7], 11) & MASK29;
    l[8] = fsr(w[7], w[8], 8) & MASK29;
}
// 9 limbs (each < 2^29) -> 9 words (261 bits)
P2E_HD void pack29_wide(const u32* l /*9*/, u32* w /*9*/) {
    P2E_UNROLL
    for (int i = 0; i < 9; i++) w[i] = 0;
    P2E_UNROLL
    for (int k = 0; k < NL; k++) {
        int bit = BITS * k;
        int wi = bit >> 5, sh = bit & 31;
        u64 v = (u64)l[k] << sh;
        w[wi] |= (u32)v;
        if (wi + 1 < 9) w[wi + 1] |= (u32)(v >> 32);
    }
}

// Goldilocks-canonical encoding of a signed integer |v| < 2^63
P2E_HD u64 gl_from_i64(i64 v) { return v >= 0 ? (u64)v : (u64)v + P_GL; }

// Goldilocks multiply (only used by the standalone CheckSum entry point, which must do a true field
// division for arbitrary inputs; the fused mul path never needs it)
P2E_HD u64 gl_reduce128(u64 lo, u64 hi) {
    u64 hh = hi >> 32, hl = hi & 0xFFFFFFFFull;
    // x = lo + hl*2^64 + hh*2^96 ; 2^64 = 2^32 - 1, 2^96 = -1 (mod p)
    u64 t0 = lo - hh;
    if (lo < hh) t0 -= 0xFFFFFFFFull;  // borrow: add p (== subtract 2^32-1 mod 2^64)
    u64 t1 = hl * 0xFFFFFFFFull;
    u64 r = t0 + t1;
    if (r < t1) r += 0xFFFFFFFFull;  // carry: subtract p (== add 2^32-1 mod 2^64)
    if (r >= P_GL) r -= P_GL;
    return r;
}
P2E_HD u64 gl_mul(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    u64 lo = a * b;
    u64 hi = __umul64hi(a, b);
#else
    unsigned __int128 p = (unsigned __int128)a * b;
    u64 lo = (u64)p, hi = (u64)(p >> 64);
#endif
    return gl_reduce128(lo, hi);
}
P2E_HD u64 gl_add(u64 a, u64 b) {
    u64 s = a + b;
    if (s < a || s >= P_GL) s -= P_GL;
    return s;
}

}  // namespace p2e
