// CPU emulation harness -- TEST INFRASTRUCTURE ONLY (never loaded by the product path).
// Compiles the very same per-lane bodies the HIP kernels run (pipeline.hpp / prims.hpp) with g++ and
// drives them with plain loops, phase by phase, so kernel logic can be debugged in the GPU-less dev
// container.  tests/ compare its output with the oracle; the GPU tests then compare the real kernels.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../plonky2-ecdsa_amd/csrc/consts.hpp"
#include "../../plonky2-ecdsa_amd/csrc/pipeline.hpp"
#include "../../plonky2-ecdsa_amd/csrc/prims.hpp"
#include "../../plonky2-ecdsa_amd/csrc/fe29.hpp"
#include "../../plonky2-ecdsa_amd/csrc/quad.hpp"
#include "../../plonky2-ecdsa_amd/csrc/schedule.hpp"
#include "../../plonky2-ecdsa_amd/csrc/curve_program.hpp"

using namespace p2e;

// E = Emit (u64 column matrix `cols`) or CompactEmit (narrow / wide matrices of the compact container)
template <class E>
static long run(int program, const uint8_t* msg, const uint8_t* r, const uint8_t* s, const uint8_t* pkx,
                const uint8_t* pky, uint64_t* cols, size_t n, size_t ld, uint8_t* err, uint8_t* valid, int chunk,
                int run_iters, uint32_t* narrow = nullptr, size_t ldn = 0, uint64_t* wide = nullptr, size_t ldw = 0) {
    host::ScheduleBuilder sb;
    if (program == 0)
        sb.verify_secp256k1_message_circuit();
    else
        sb.glv_mul_circuit();
    if (chunk < 0) run_iters = 0;   // the small-batch plan expands op by op (plain op table), as run_program() does
    sb.mark_runs(run_iters);
    const Program& G = sb.prog;
    const host::Consts& C = host::consts();
    std::vector<U256> PX((size_t)G.num_slots * n), PY((size_t)G.num_slots * n), PZ((size_t)G.num_slots * n),
        PW((size_t)G.num_ops * n), PREF((size_t)G.num_ops * n), AX((size_t)G.num_slots * n), AY((size_t)G.num_slots * n);
    std::vector<uint8_t> dig4((size_t)FB_WINDOWS * n), dig2((size_t)MSM_DIGITS * n), valid8(n);
    std::vector<uint16_t> dyn((size_t)G.num_cadd * n), src((size_t)G.num_ops * 2 * n), msrc((size_t)MSM_DIGITS * n);
    std::vector<u32> err32(n);
    Buffers B{};
    B.msg = msg; B.r = r; B.s = s; B.pkx = pkx; B.pky = pky;
    // wide_before[c] = check_sum / carry columns (33 per mul generator, after its r and q limbs) before column c
    std::vector<u32> wide_before((size_t)G.num_cols + 1, 0);
    for (const auto& g : sb.gens)
        for (u32 k = 0; k < g.ncols; k++) wide_before[g.col + k + 1] = (g.kind == host::GEN_MUL && k >= 2 * NL) ? 1u : 0u;
    for (size_t c2 = 1; c2 < wide_before.size(); c2++) wide_before[c2] += wide_before[c2 - 1];
    B.sink = Sink{cols, ld, narrow, ldn, wide, ldw, wide_before.data()};
    B.n = n;
    B.err = err32.data(); B.valid = valid8.data();
    B.PX = PX.data(); B.PY = PY.data(); B.PZ = PZ.data(); B.PW = PW.data(); B.PREF = PREF.data();
    B.AX = AX.data(); B.AY = AY.data();
    B.dig4 = dig4.data(); B.dig2 = dig2.data(); B.dyn = dyn.data(); B.src = src.data(); B.msrc = msrc.data();
    B.cpts = C.cpts; B.fbtab = C.fbtab.data(); B.ops = sb.ops.data();
#pragma omp parallel for
    for (long long i = 0; i < (long long)n; i++) body_scalar<E>(G, B, (size_t)i);
    // same dependency order as the stream plan of p2e_hip.hip, with the MSM chain cut into `pieces` pieces whose
    // phases B and C are interleaved with the following pieces of the chain
    // chunk >= piece length: one batch per piece using phase A's prefix products (what the GPU launches);
    // smaller chunks re-run the forward pass inside phase B (arbitrary batching must not change any output)
    // chunk < 0: the small-batch plan of run_program() (quad.hpp): four lanes per signature walk the chains, the
    // window table is walked as rows, every inversion batch is cut into -chunk sub-ranges
    const bool quad = chunk < 0;
    const int split = quad ? -chunk : 1;
    auto binv = [&](int lo, int hi, bool have_prefix = true) {
        if (quad) {
#pragma omp parallel for
            for (long long i = 0; i < (long long)n; i++)
                for (int q = 0; q < split; q++) body_batch_inv_split(G, B, (size_t)i, lo, hi, have_prefix, q, split);
            return;
        }
        if (chunk >= hi - lo) {
#pragma omp parallel for
            for (long long i = 0; i < (long long)n; i++) body_batch_inv(G, B, (size_t)i, lo, hi, true);
            return;
        }
        for (int t0 = lo; t0 < hi; t0 += chunk) {
            int t1 = t0 + chunk < hi ? t0 + chunk : hi;
#pragma omp parallel for
            for (long long i = 0; i < (long long)n; i++) body_batch_inv(G, B, (size_t)i, t0, t1, false);
        }
    };
    auto chain = [&](int lo, int hi, bool table_affine, bool cont = false) {
#pragma omp parallel for
        for (long long i = 0; i < (long long)n; i++) {
            if (quad)   // every role walks the whole range (the host form of a level computes all four products) and
                        // writes only its own share of the scratch outputs
                for (int role = 0; role < 4; role++) body_chain_range_quad(G, B, (size_t)i, role, lo, hi, table_affine, cont);
            else
                body_chain_range(G, B, (size_t)i, lo, hi, table_affine, cont);
        }
    };
    auto expand = [&](int lo, int hi) {
#pragma omp parallel for
        for (long long i = 0; i < (long long)n; i++)
            for (int t = lo; t < hi; t++) body_expand<E>(G, B, (size_t)i, t);
    };
    auto expand_runs = [&](int it0, int it1) {
        for (int a = it0; a < it1; a += run_iters) {
            int b = a + run_iters < it1 ? a + run_iters : it1;
#pragma omp parallel for
            for (long long i = 0; i < (long long)n; i++) body_expand_run<E>(G, B, (size_t)i, a, b);
        }
    };
    // the same plan as run_program() in csrc/p2e_hip.hip: fixed-base chain; window table piece (inverted
    // before the loop, which then reads it affine); the loop cut at run boundaries into `groups` pieces whose
    // phase C walks runs; the trailing unblinding add and the final add op by op
    if (G.num_chains == 3) {
        chain(G.chain_begin[1], G.chain_end[1], false);
        binv(G.chain_begin[1], G.chain_end[1]);
        if (run_iters > 0 && !quad) {   // the windows as one run per signature, then the unblinding add (k_expand_fb_run)
            const int t_after = G.fb_begin + G.fb_windows;
#pragma omp parallel for
            for (long long i = 0; i < (long long)n; i++) body_expand_fb_run<E>(G, B, (size_t)i, t_after);
            expand(t_after, G.chain_end[1]);
        } else {
            expand(G.chain_begin[1], G.chain_end[1]);
        }
    }
    {
        const int lo0 = G.chain_begin[0], hi0 = G.chain_end[0], lb = G.msm_loop_begin, iters = G.msm_loop_iters;
        const int le = lb + 3 * iters, R = run_iters > 0 ? run_iters : 1, groups = 3;
        if (G.num_chains == 1 || quad) {
            // glv_mul alone (and small batches): the window table as rows of independent sub-chains, as the GPU walks it there
            // (body_chain_rows), prefix products left to phase B
            auto rows = [&](int lo, int nrows, int count) {
#pragma omp parallel for
                for (long long i = 0; i < (long long)n; i++)
                    for (int r = 0; r < nrows; r++) body_chain_rows(G, B, (size_t)i, lo, nrows, r, count);
            };
            rows(lo0, 2, 4);
            rows(lo0 + 8, 6, 1);
            rows(lo0 + 14, 9, 1);
            if (quad) {
                binv(lo0, lb, false);
            } else {
#pragma omp parallel for
                for (long long i = 0; i < (long long)n; i++) body_batch_inv(G, B, (size_t)i, lo0, lb, false);
            }
        } else {
            chain(lo0, lb, false);
            binv(lo0, lb);
        }
        expand(lo0, lb);
        const int nruns = (iters + R - 1) / R;
        int rn = 0;
        for (int g = 0; g < groups; g++) {
            int take = (nruns - rn + (groups - g) - 1) / (groups - g);
            int it0 = rn * R, it1 = (rn + take) * R < iters ? (rn + take) * R : iters;
            int a = lb + 3 * it0, b = lb + 3 * it1;
            const bool last = g == groups - 1;
            if (last) b = hi0;
            chain(a, b, true);
            if (last && G.num_chains == 3) {
                chain(G.chain_begin[2], G.chain_end[2], false, true);
                b = G.chain_end[2];
            }
            binv(a, b);
            if (run_iters > 0) {
                expand_runs(it0, it1);
                if (last) expand(le, b);
            } else {
                expand(a, b);
            }
            rn += take;
        }
    }
    long bad = 0;
    for (size_t i = 0; i < n; i++) {
        err[i] = (uint8_t)err32[i];
        if (valid) valid[i] = err32[i] ? 0 : valid8[i];
        bad += err32[i] != 0;
    }
    return bad;
}

// ---- fe29.hpp (lazy 29-bit limbs) against fe.hpp (canonical words): every operation on patterned and random values,
// with operands pushed through the lazy forms the chains use (sums, differences, small multiples) before they are used
static U256 f29_test_value(host::SplitMix64& rng) {
    U256 x;
    const unsigned long long sel = rng.next();
    for (int k = 0; k < 4; k++) {
        unsigned long long w = rng.next();
        const unsigned pat = (unsigned)(sel >> (8 * k)) & 7;
        if (pat == 0) w = 0;
        if (pat == 1) w = ~0ull;
        if (pat == 2) w &= 0xFFFFFFFFull;
        x.w[2 * k] = (u32)w;
        x.w[2 * k + 1] = (u32)(w >> 32);
    }
    const unsigned kind = (unsigned)(sel >> 40) & 15;
    if (kind == 0) x = u256_zero();
    if (kind == 1) x = u256_small(1);
    if (kind == 2 || kind == 3) {   // p - 1, p - small
        x = u256_zero();
        x = fe_sub<ModP>(x, u256_small(kind == 2 ? 1 : (u32)(sel >> 48) & 0xFFF));
    }
    return fe_canon<ModP>(fe_canon<ModP>(x));   // (two conditional subtractions: any 256-bit pattern -> canonical)
}
extern "C" long emu_f29_selftest(unsigned long long seed, size_t n) {
    long fails = 0;
#pragma omp parallel for reduction(+ : fails)
    for (long long i = 0; i < (long long)n; i++) {
        host::SplitMix64 rng{seed ^ (0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1))};
        const U256 a = f29_test_value(rng), b = f29_test_value(rng), c = f29_test_value(rng);
        const F29 fa = f29_from_u256(a), fb = f29_from_u256(b), fc = f29_from_u256(c);
        auto same = [&](const F29& got, const U256& want) { fails += !u256_eq(f29_canon(f29_norm(got)), want); };
        auto same_direct = [&](const F29& got, const U256& want) { fails += !u256_eq(f29_canon(got), want); };   // limbs < 2^31
        typedef ModP F;
        same(fa, a);
        same_direct(fa, a);
        same_direct(f29_mul(fa, fb), fe_mul<F>(a, b));
        same_direct(f29_sub<1>(fa, fb), fe_sub<F>(a, b));
        same_direct(f29_add(f29_add(fa, fb), fc), fe_add<F>(fe_add<F>(a, b), c));
        same(f29_mul(fa, fb), fe_mul<F>(a, b));
        same(f29_sqr(fa), fe_sqr<F>(a));
        same(f29_add(fa, fb), fe_add<F>(a, b));
        same(f29_sub<1>(fa, fb), fe_sub<F>(a, b));
        same(f29_norm(f29_sub<1>(fa, fb)), fe_sub<F>(a, b));
        same(f29_times<3>(fa), fe_add<F>(fe_add<F>(a, a), a));
        // lazy operands: (a + b) (2 c), (a - b)(c - a), (a + b)^2, (a - b - c)^2 normalised first, 4 (a - 2 b)
        const U256 apb = fe_add<F>(a, b), amb = fe_sub<F>(a, b), cma = fe_sub<F>(c, a), c2 = fe_add<F>(c, c);
        same(f29_mul(f29_add(fa, fb), f29_times<2>(fc)), fe_mul<F>(apb, c2));
        same(f29_mul(f29_norm(f29_sub<1>(fa, fb)), f29_sub<1>(fc, fa)), fe_mul<F>(amb, cma));
        same(f29_sqr(f29_add(fa, fb)), fe_sqr<F>(apb));
        same(f29_sqr(f29_norm(f29_sub<1>(f29_sub<1>(fa, fb), fc))), fe_sqr<F>(fe_sub<F>(amb, c)));
        const U256 b2 = fe_add<F>(b, b), am2b = fe_sub<F>(a, b2), am2b2 = fe_add<F>(am2b, am2b);
        same(f29_times<4>(f29_norm(f29_sub<2>(fa, f29_times<2>(fb)))), fe_add<F>(am2b2, am2b2));
        same(f29_sub<3>(fa, f29_times<3>(fb)), fe_sub<F>(a, fe_add<F>(b2, b)));
        same(f29_sub<4>(fa, f29_add(f29_times<3>(fb), fc)), fe_sub<F>(fe_sub<F>(a, fe_add<F>(b2, b)), c));
        // a chain of dependent operations on lazy values
        F29 x = fa;
        U256 y = a;
        for (int k = 0; k < 6; k++) {
            x = f29_mul(f29_add(x, fb), f29_sub<1>(x, fc));
            y = fe_mul<F>(fe_add<F>(y, b), fe_sub<F>(y, c));
            x = f29_sqr(f29_norm(f29_sub<1>(f29_times<2>(x), fa)));
            y = fe_sqr<F>(fe_sub<F>(fe_add<F>(y, y), a));
        }
        same(x, y);
        fails += f29_is_zero(f29_sub<1>(fa, fa)) ? 0 : 1;
        fails += f29_is_zero(f29_mul(f29_sub<1>(fa, fb), f29_norm(f29_sub<1>(fb, fa)))) != u256_eq(a, b) ? 1 : 0;
    }
    return fails;
}

// the same stress loop for fe_inv_safegcd (csrc/fe.hpp), over all four moduli of the crate (field: 0 p, 1 n of secp256k1,
// 2 p, 3 n of P-256): x * inv(x) == 1, the result is canonical and equal to fe_inv_bingcd's; zero is reported as not invertible
template <class MOD>
static long safegcd_one(const U256& raw) {
    const U256 x = fe_canon<MOD>(fe_canon<MOD>(raw));
    U256 r, r2;
    const bool ok = fe_inv_safegcd<MOD>(x, r);
    if (u256_is_zero(x)) return ok ? 1 : 0;
    const bool ok2 = fe_inv_bingcd<MOD>(x, r2);
    return (ok && ok2 && u256_eq(r, r2) && !geq_mod<MOD>(r.w) && u256_eq(fe_mul<MOD>(x, r), u256_small(1))) ? 0 : 1;
}
// ---- ec29.hpp / quad29.hpp against ec.hpp / quad.hpp: every variant of the addition and the doubling, lane per signature and
// four-lane form, on arbitrary coordinates (the formulas never use the curve equation), under the limb-bound tracking
template <bool Z1ONE, bool Z2ONE>
static long ec29_add_case(const Jac& p1, const Jac& p2, const U256& acc, const U256& zz1) {
    long fails = 0;
    Jac a = p1, b = p2;
    if (Z1ONE) a.Z = u256_small(1);
    if (Z2ONE) b.Z = u256_small(1);
    const JacW want = jac_add<Z1ONE, Z2ONE>(a, b);
    const JacWL got = jac_add29<Z1ONE, Z2ONE>(jacl_from(a), jacl_from(b));
    fails += !u256_eq(f29_canon(got.p.X), want.p.X) + !u256_eq(f29_canon(got.p.Y), want.p.Y) + !u256_eq(f29_canon(got.p.Z), want.p.Z) +
             !u256_eq(f29_canon(got.W), want.W);
    for (int have = 0; have < 2; have++) {
        const U256 z1sq = fe_sqr<ModP>(a.Z);
        const QuadRes wq = jac_add_quad<Z1ONE, Z2ONE>(0, a, have != 0, have ? z1sq : zz1, b, acc);
        for (int role = 0; role < 4; role++) {
            const QuadRes29 q = jac_add_quad29<Z1ONE, Z2ONE>(role, false, jacl_from(a), have != 0, f29_from_u256(have ? z1sq : zz1), jacl_from(b),
                                                             f29_from_u256(acc));
            const U256 mine = role == 0 ? wq.res.p.X : role == 1 ? acc : role == 2 ? wq.res.p.Z : wq.res.W;
            fails += !u256_eq(f29_canon(q.p.X), wq.res.p.X) + !u256_eq(f29_canon(q.p.Y), wq.res.p.Y) + !u256_eq(f29_canon(q.p.Z), wq.res.p.Z) +
                     !u256_eq(f29_canon(q.acc), wq.acc) + !u256_eq(f29_canon(q.zz3), wq.zz3) + !u256_eq(q.mine, mine) + (q.z3_zero != wq.z3_zero);
            if (!Z1ONE) fails += !u256_eq(f29_canon(q.zz1), wq.zz1);
        }
    }
    return fails;
}
extern "C" long emu_ec29_selftest(unsigned long long seed, size_t n) {
    long fails = 0;
#pragma omp parallel for reduction(+ : fails)
    for (long long i = 0; i < (long long)n; i++) {
        host::SplitMix64 rng{seed ^ (0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1))};
        Jac p1, p2;
        p1.X = f29_test_value(rng); p1.Y = f29_test_value(rng); p1.Z = f29_test_value(rng);
        p2.X = f29_test_value(rng); p2.Y = f29_test_value(rng); p2.Z = f29_test_value(rng);
        const U256 acc = f29_test_value(rng), zz1 = f29_test_value(rng);
        if ((i & 15) == 0) p2 = p1;                       // equal operands: H = 0, Z3 = 0 (the reference's inverse-of-zero case)
        fails += ec29_add_case<false, false>(p1, p2, acc, zz1) + ec29_add_case<false, true>(p1, p2, acc, zz1) +
                 ec29_add_case<true, false>(p1, p2, acc, zz1) + ec29_add_case<true, true>(p1, p2, acc, zz1);
        const JacW wd = jac_dbl(p1);
        const JacWL gd = jac_dbl29(jacl_from(p1));
        fails += !u256_eq(f29_canon(gd.p.X), wd.p.X) + !u256_eq(f29_canon(gd.p.Y), wd.p.Y) + !u256_eq(f29_canon(gd.p.Z), wd.p.Z) +
                 !u256_eq(f29_canon(gd.W), wd.W);
        const QuadRes wq = jac_dbl_quad(0, p1, acc);
        for (int role = 0; role < 4; role++)
            for (int na = 0; na < 2; na++) {
                const QuadRes29 q = jac_dbl_quad29(role, na != 0, jacl_from(p1), f29_from_u256(acc));
                const U256 mine = role == 1 ? acc : role == 3 ? wq.res.W : (role == 0 && !na) ? wq.res.p.X : wq.res.p.Z;
                fails += !u256_eq(f29_canon(q.p.X), wq.res.p.X) + !u256_eq(f29_canon(q.p.Y), wq.res.p.Y) + !u256_eq(f29_canon(q.p.Z), wq.res.p.Z) +
                         !u256_eq(f29_canon(q.acc), wq.acc) + !u256_eq(f29_canon(q.zz3), wq.zz3) + !u256_eq(f29_canon(q.zz1), wq.zz1) +
                         !u256_eq(q.mine, mine) + (q.z3_zero != wq.z3_zero);
            }
    }
    return fails;
}

extern "C" {
// raw access to the binary-GCD inversion (csrc/fe.hpp) for the stress test: ok[i] = round bound held
long emu_bingcd(int field, const uint8_t* x32, uint8_t* inv32, uint8_t* ok, size_t n) {
    long fails = 0;
#pragma omp parallel for reduction(+ : fails)
    for (long long i = 0; i < (long long)n; i++) {
        U256 x, r;
        std::memcpy(x.w, x32 + 32 * i, 32);
        bool good = field ? fe_inv_bingcd<ModN>(x, r) : fe_inv_bingcd<ModP>(x, r);
        std::memcpy(inv32 + 32 * i, r.w, 32);
        ok[i] = good;
        fails += !good;
    }
    return fails;
}
// self-checking stress loop: x from splitmix64 with assorted bit lengths / word patterns, x * inv(x) == 1
long emu_bingcd_selfcheck(int field, unsigned long long seed, size_t n) {
    long fails = 0;
#pragma omp parallel for reduction(+ : fails)
    for (long long i = 0; i < (long long)n; i++) {
        host::SplitMix64 rng{seed ^ (0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1))};
        U256 x;
        unsigned long long sel = rng.next();
        for (int k = 0; k < 4; k++) {
            unsigned long long w = rng.next();
            unsigned pat = (unsigned)(sel >> (8 * k)) & 7;
            if (pat == 0) w = 0;
            if (pat == 1) w = ~0ull;
            if (pat == 2) w &= 0xFFFFFFFFull;
            x.w[2 * k] = (u32)w;
            x.w[2 * k + 1] = (u32)(w >> 32);
        }
        int bits = 1 + (int)((sel >> 40) % 256);
        for (int b = bits; b < 256; b++) x.w[b >> 5] &= ~(1u << (b & 31));
        x = field ? fe_canon<ModN>(x) : fe_canon<ModP>(x);
        if (u256_is_zero(x)) continue;
        U256 r;
        bool ok = field ? fe_inv_bingcd<ModN>(x, r) : fe_inv_bingcd<ModP>(x, r);
        U256 one = field ? fe_mul<ModN>(x, r) : fe_mul<ModP>(x, r);
        if (!ok || !u256_eq(one, u256_small(1))) fails++;
    }
    return fails;
}
long emu_safegcd_selfcheck(int field, unsigned long long seed, size_t n) {
    long fails = 0;
#pragma omp parallel for reduction(+ : fails)
    for (long long i = 0; i < (long long)n; i++) {
        host::SplitMix64 rng{seed ^ (0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1))};
        U256 x;
        unsigned long long sel = rng.next();
        for (int k = 0; k < 4; k++) {
            unsigned long long w = rng.next();
            unsigned pat = (unsigned)(sel >> (8 * k)) & 7;
            if (pat == 0) w = 0;
            if (pat == 1) w = ~0ull;
            if (pat == 2) w &= 0xFFFFFFFFull;
            x.w[2 * k] = (u32)w;
            x.w[2 * k + 1] = (u32)(w >> 32);
        }
        int bits = 1 + (int)((sel >> 40) % 256);
        if ((sel >> 50) & 1)
            for (int b = bits; b < 256; b++) x.w[b >> 5] &= ~(1u << (b & 31));
        if (((sel >> 52) & 63) == 0) {   // m - small
            x = u256_zero();
            x.w[0] = (u32)(sel >> 58) + 1u;
            x = field == 0 ? fe_neg<ModP>(x) : field == 1 ? fe_neg<ModN>(x) : field == 2 ? fe_neg<ModP256>(x) : fe_neg<ModN256>(x);
        }
        fails += field == 0 ? safegcd_one<ModP>(x) : field == 1 ? safegcd_one<ModN>(x) : field == 2 ? safegcd_one<ModP256>(x) : safegcd_one<ModN256>(x);
    }
    return fails;
}
long emu_verify(const uint8_t* msg, const uint8_t* r, const uint8_t* s, const uint8_t* pkx, const uint8_t* pky,
                uint64_t* cols, size_t n, size_t ld, uint8_t* err, uint8_t* valid, int chunk, int run_iters) {
    return run<Emit>(0, msg, r, s, pkx, pky, cols, n, ld, err, valid, chunk, run_iters);
}
// verdict only (p2e_ecdsa_verify_batch): scalar phase without emission, Jacobian chains, r == x on the Jacobian result
long emu_verify_only(const uint8_t* msg, const uint8_t* r, const uint8_t* s, const uint8_t* pkx, const uint8_t* pky, size_t n,
                     uint8_t* err, uint8_t* valid) {
    host::ScheduleBuilder sb;
    sb.verify_secp256k1_message_circuit();
    const Program& G = sb.prog;
    const host::Consts& C = host::consts();
    std::vector<U256> PX((size_t)G.num_slots * n), PY((size_t)G.num_slots * n), PZ((size_t)G.num_slots * n),
        PW((size_t)G.num_ops * n), PREF((size_t)G.num_ops * n), AX((size_t)G.num_slots * n), AY((size_t)G.num_slots * n);
    std::vector<uint8_t> dig4((size_t)FB_WINDOWS * n), dig2((size_t)MSM_DIGITS * n), valid8(n);
    std::vector<uint16_t> dyn((size_t)G.num_cadd * n), src((size_t)G.num_ops * 2 * n), msrc((size_t)MSM_DIGITS * n);
    std::vector<u32> err32(n);
    Buffers B{};
    B.msg = msg; B.r = r; B.s = s; B.pkx = pkx; B.pky = pky;
    B.n = n;
    B.err = err32.data(); B.valid = valid8.data();
    B.PX = PX.data(); B.PY = PY.data(); B.PZ = PZ.data(); B.PW = PW.data(); B.PREF = PREF.data();
    B.AX = AX.data(); B.AY = AY.data();
    B.dig4 = dig4.data(); B.dig2 = dig2.data(); B.dyn = dyn.data(); B.src = src.data(); B.msrc = msrc.data();
    B.cpts = C.cpts; B.fbtab = C.fbtab.data(); B.ops = sb.ops.data();
#pragma omp parallel for
    for (long long i = 0; i < (long long)n; i++) {
        body_scalar<NullEmit>(G, B, (size_t)i);
        body_chain_range(G, B, (size_t)i, G.chain_begin[1], G.chain_end[1], false, false);
        body_chain_range(G, B, (size_t)i, G.chain_begin[0], G.chain_end[0], false, false);
        body_chain_range(G, B, (size_t)i, G.chain_begin[2], G.chain_end[2], false, false);
        body_verify_check(G, B, (size_t)i);
    }
    long bad = 0;
    for (size_t i = 0; i < n; i++) {
        err[i] = (uint8_t)err32[i];
        valid[i] = err32[i] ? 0 : valid8[i];
        bad += err32[i] != 0;
    }
    return bad;
}
// the same walk writing the compact container directly (CompactEmit: what the compact kernels' ragged tail runs)
long emu_verify_compact(const uint8_t* msg, const uint8_t* r, const uint8_t* s, const uint8_t* pkx, const uint8_t* pky,
                        uint32_t* narrow, size_t ldn, uint64_t* wide, size_t ldw, size_t n, uint8_t* err, uint8_t* valid,
                        int run_iters) {
    return run<CompactEmit>(0, msg, r, s, pkx, pky, nullptr, n, 0, err, valid, 512, run_iters, narrow, ldn, wide, ldw);
}
long emu_glv_mul_compact(const uint8_t* px, const uint8_t* py, const uint8_t* k, uint32_t* narrow, size_t ldn, uint64_t* wide,
                         size_t ldw, size_t n, uint8_t* err, uint8_t* valid, int run_iters) {
    return run<CompactEmit>(1, k, k, k, px, py, nullptr, n, 0, err, valid, 512, run_iters, narrow, ldn, wide, ldw);
}
long emu_glv_mul(const uint8_t* px, const uint8_t* py, const uint8_t* k, uint64_t* cols, size_t n, size_t ld,
                 uint8_t* err, uint8_t* valid, int chunk, int run_iters) {
    return run<Emit>(1, k, k, k, px, py, cols, n, ld, err, valid, chunk, run_iters);
}
// built-in-generator columns from a finished witness matrix (aux.hpp: the body k_aux runs)
long emu_aux(int program, const uint8_t* pky, const uint64_t* cols, size_t ld, uint64_t* aux, size_t ald, size_t n,
             uint8_t* err) {
    host::ScheduleBuilder sb;
    if (program == 0)
        sb.verify_secp256k1_message_circuit();
    else
        sb.glv_mul_circuit();
    const host::Consts& C = host::consts();
    std::vector<u32> err32(n);
    AuxArgs A{cols, ld, aux, ald, n, pky, C.cpts, C.fbtab.data(), sb.aux_items.data(), &sb.aux_tab, err32.data(),
              nullptr, 0, nullptr};
    for (int item = 0; item < (int)sb.aux_items.size(); item++) {
#pragma omp parallel for
        for (long long i = 0; i < (long long)n; i++) body_aux<Emit>(A, item, (size_t)i);
    }
    long bad = 0;
    for (size_t i = 0; i < n; i++) {
        err[i] = (uint8_t)err32[i];
        bad += err32[i] != 0;
    }
    return bad;
}
// the same pass inside the compact container: narrow u32 matrix in, u32 aux matrix out
long emu_aux_compact(int program, const uint8_t* pky, const uint32_t* narrow, size_t ldn, uint32_t* aux32, size_t ald, size_t n,
                     uint8_t* err) {
    host::ScheduleBuilder sb;
    if (program == 0)
        sb.verify_secp256k1_message_circuit();
    else
        sb.glv_mul_circuit();
    const host::Consts& C = host::consts();
    std::vector<u32> wide_before((size_t)sb.prog.num_cols + 1, 0);
    for (const auto& g : sb.gens)
        for (u32 k = 0; k < g.ncols; k++) wide_before[g.col + k + 1] = (g.kind == host::GEN_MUL && k >= 2 * NL) ? 1u : 0u;
    for (size_t c2 = 1; c2 < wide_before.size(); c2++) wide_before[c2] += wide_before[c2 - 1];
    std::vector<u32> err32(n);
    AuxArgs A{nullptr, 0, aux32, ald, n, pky, C.cpts, C.fbtab.data(), sb.aux_items.data(), &sb.aux_tab, err32.data(),
              narrow, ldn, wide_before.data()};
    for (int item = 0; item < (int)sb.aux_items.size(); item++) {
#pragma omp parallel for
        for (long long i = 0; i < (long long)n; i++) body_aux<Emit32>(A, item, (size_t)i);
    }
    long bad = 0;
    for (size_t i = 0; i < n; i++) {
        err[i] = (uint8_t)err32[i];
        bad += err32[i] != 0;
    }
    return bad;
}
// constraint-block columns from finished matrices (ux.hpp: the body k_ux runs); inputs by INPUT_* slot
long emu_ux(int program, const uint8_t* msg, const uint8_t* r, const uint8_t* s, const uint8_t* pkx, const uint8_t* pky,
            const uint64_t* cols, size_t ld, const uint64_t* aux, size_t ald, uint64_t* ux, size_t uld, size_t n, uint8_t* err) {
    host::ScheduleBuilder sb;
    if (program == 0)
        sb.verify_secp256k1_message_circuit();
    else
        sb.glv_mul_circuit();
    U256 cv[NUM_CONSTV];
    for (u32 id = 0; id < NUM_CONSTV; id++) cv[id] = host::ScheduleBuilder::const_value(id);
    std::vector<u32> err32(n);
    UxArgs A{};
    A.cols = cols; A.ld = ld; A.aux = aux; A.ald = ald; A.ux = ux; A.uld = uld; A.n = n;
    A.in[INPUT_PY] = pky; A.in[INPUT_PX] = pkx; A.in[INPUT_MSG] = msg; A.in[INPUT_R] = r ? r : msg; A.in[INPUT_S] = s ? s : msg;
    A.consts = cv; A.items = sb.ux_items.data(); A.err = err32.data();
    for (int item = 0; item < (int)sb.ux_items.size(); item++) {
#pragma omp parallel for
        for (long long i = 0; i < (long long)n; i++) body_ux<Emit>(A, item, (size_t)i);
    }
    long bad = 0;
    for (size_t i = 0; i < n; i++) {
        err[i] = (uint8_t)err32[i];
        bad += err32[i] != 0;
    }
    return (long)sb.num_ux_cols + 0 * bad;
}
// gate-internal values from an aux matrix (aux.hpp body_gate)
long emu_gate(int program, const uint64_t* aux, size_t ald, uint64_t* gate, size_t gld, size_t n) {
    host::ScheduleBuilder sb;
    if (program == 0)
        sb.verify_secp256k1_message_circuit();
    else
        sb.glv_mul_circuit();
    GateArgs A{};
    A.aux = aux; A.ald = ald; A.gate = gate; A.gld = gld; A.n = n; A.items = sb.gate_items.data();
    A.inv16[0] = 0;
    for (u64 d = 1; d < 16; d++) {
        u64 r = 1, a = d, e = P_GL - 2;
        while (e) {
            if (e & 1) r = gl_mul(r, a);
            a = gl_mul(a, a);
            e >>= 1;
        }
        A.inv16[d] = r;
    }
    for (int item = 0; item < (int)sb.gate_items.size(); item++)
        for (size_t i = 0; i < n; i++) body_gate<Emit>(A, item, i);
    return (long)sb.num_gate_cols;
}
long emu_aux_num_cols(int program) {
    host::ScheduleBuilder sb;
    if (program == 0)
        sb.verify_secp256k1_message_circuit();
    else
        sb.glv_mul_circuit();
    return (long)sb.aux_tab.num_aux_cols;
}
#define LOOP(expr)                                  \
    long bad = 0;                                   \
    for (size_t i = 0; i < n; i++) {                \
        uint8_t e = (expr);                         \
        err[i] = e;                                 \
        bad += e != 0;                              \
    }                                               \
    return bad;
long emu_mul(int field, const u64* x, const u64* y, u64* r, u64* q, u64* cs, u64* b, size_t n, size_t ld, uint8_t* err) {
    LOOP(field == 3   ? prim_mul<ModN256>(x, y, r, q, cs, b, ld, i)
         : field == 2 ? prim_mul<ModP256>(x, y, r, q, cs, b, ld, i)
         : field      ? prim_mul<ModN>(x, y, r, q, cs, b, ld, i)
                      : prim_mul<ModP>(x, y, r, q, cs, b, ld, i))
}
long emu_checksum(const u64* a, u64* b, size_t n, size_t ld, uint8_t* err) { LOOP(prim_checksum(a, b, ld, i)) }
long emu_add(int field, const u64* a, const u64* b, u64* out, u64* ov, size_t n, size_t ld, uint8_t* err) {
    LOOP(field == 3   ? (prim_addsub<ModN256, false>(a, b, out, ov, ld, i))
         : field == 2 ? (prim_addsub<ModP256, false>(a, b, out, ov, ld, i))
         : field      ? (prim_addsub<ModN, false>(a, b, out, ov, ld, i))
                      : (prim_addsub<ModP, false>(a, b, out, ov, ld, i)))
}
long emu_sub(int field, const u64* a, const u64* b, u64* out, u64* ov, size_t n, size_t ld, uint8_t* err) {
    LOOP(field == 3   ? (prim_addsub<ModN256, true>(a, b, out, ov, ld, i))
         : field == 2 ? (prim_addsub<ModP256, true>(a, b, out, ov, ld, i))
         : field      ? (prim_addsub<ModN, true>(a, b, out, ov, ld, i))
                      : (prim_addsub<ModP, true>(a, b, out, ov, ld, i)))
}
long emu_add_many(int field, const u64* s, int k, u64* out, u64* ov, size_t n, size_t ld, uint8_t* err) {
    LOOP(field == 3   ? prim_add_many<ModN256>(s, k, out, ov, ld, i)
         : field == 2 ? prim_add_many<ModP256>(s, k, out, ov, ld, i)
         : field      ? prim_add_many<ModN>(s, k, out, ov, ld, i)
                      : prim_add_many<ModP>(s, k, out, ov, ld, i))
}
long emu_inv(int field, const u64* x, u64* inv, u64* div, size_t n, size_t ld, uint8_t* err) {
    LOOP(field == 3   ? prim_inv<ModN256>(x, inv, div, ld, i)
         : field == 2 ? prim_inv<ModP256>(x, inv, div, ld, i)
         : field      ? prim_inv<ModN>(x, inv, div, ld, i)
                      : prim_inv<ModP>(x, inv, div, ld, i))
}
long emu_div_rem(const u64* a, int na, const u64* b, int nb, u64* div, u64* rem, size_t n, size_t ld, uint8_t* err) {
    LOOP(prim_div_rem(a, na, b, nb, div, rem, ld, i))
}
long emu_glv(const u64* k, u64* k1, u64* k2, u64* n1, u64* n2, size_t n, size_t ld, uint8_t* err) {
    LOOP(prim_glv(k, k1, k2, n1, n2, ld, i))
}
long emu_split(const uint8_t* packed, u64* limbs, size_t n, size_t ld) {
    for (size_t i = 0; i < n; i++) prim_split(packed, limbs, ld, i);
    return 0;
}
long emu_pack(const u64* limbs, uint8_t* packed, size_t n, size_t ld, uint8_t* err) { LOOP(prim_pack(limbs, packed, ld, i)) }
}

// ---- curve programs (curves.hpp): the bodies kc_scalar / kc_chains / kc_batch_inv / kc_expand run, in the piece order
// of run_curve_program (pieces of `piece` ops, the per-signature table as its own piece and affine afterwards) --------
template <class CV>
static long run_curve(const host::CurveProgramHost& H, const uint8_t* msg, const uint8_t* r, const uint8_t* s, const uint8_t* pkx,
                      const uint8_t* pky, uint64_t* cols, size_t n, size_t ld, uint8_t* err, uint8_t* valid, int piece) {
    const host::ScheduleBuilder& sb = H.sb;
    const Program& G = sb.prog;
    const size_t rows = (size_t)(G.cp_rows > MSM_DIGITS ? G.cp_rows : MSM_DIGITS);
    std::vector<U256> PX((size_t)G.num_slots * n), PY((size_t)G.num_slots * n), PZ((size_t)G.num_slots * n),
        PW((size_t)G.num_ops * n), PREF((size_t)G.num_ops * n), AX((size_t)G.num_slots * n), AY((size_t)G.num_slots * n);
    std::vector<uint8_t> dig4((size_t)FB_WINDOWS * n), dig2(rows * n), valid8(n);
    std::vector<uint16_t> dyn((size_t)G.num_cadd * n + 1), src((size_t)G.num_ops * 2 * n), msrc(rows * n);
    std::vector<u32> err32(n);
    Buffers B{};
    B.msg = msg; B.r = r; B.s = s; B.pkx = pkx; B.pky = pky;
    B.sink = Sink{cols, ld, nullptr, 0, nullptr, 0, nullptr};
    B.n = n;
    B.err = err32.data(); B.valid = valid8.data();
    B.PX = PX.data(); B.PY = PY.data(); B.PZ = PZ.data(); B.PW = PW.data(); B.PREF = PREF.data();
    B.AX = AX.data(); B.AY = AY.data();
    B.dig4 = dig4.data(); B.dig2 = dig2.data(); B.dyn = dyn.data(); B.src = src.data(); B.msrc = msrc.data();
    B.cpts = sb.gpts.data(); B.fbtab = sb.gfbtab.data(); B.ops = sb.ops.data();
#pragma omp parallel for
    for (long long i = 0; i < (long long)n; i++) body_cscalar<CV, Emit>(G, B, (size_t)i);
    int tb = -1, te = -1;
    if (G.cp_table_ops > 0) {
        te = 0;
        for (int k = 1; k < 16; k++) te = std::max(te, (int)ref_id(G.msm_tab[k]) + 1);
        tb = te - G.cp_table_ops;
    }
    bool table_done = false;
    auto chain_binv = [&](int lo, int hi, bool table_affine) {
#pragma omp parallel for
        for (long long i = 0; i < (long long)n; i++) {
            body_chain_range<CV, true>(G, B, (size_t)i, lo, hi, table_affine, false);
            body_batch_inv<CV>(G, B, (size_t)i, lo, hi, true);
        }
    };
    auto expand = [&](int lo, int hi) {
#pragma omp parallel for
        for (long long i = 0; i < (long long)n; i++)
            for (int t = lo; t < hi; t++) body_expand<Emit, CV>(G, B, (size_t)i, t);
    };
    host::ScheduleBuilder marked;
    if (piece < 0) {
        // the large-batch plan of run_curve_program: runs of -piece windows (op table with the run marks), the verifier's
        // fixed-base windows as one run per signature
        const int R = -piece, lb = G.msm_loop_begin, iters = G.msm_loop_iters, le = lb + 5 * iters;
        if (iters <= 0) return -2;
        marked = sb;
        marked.mark_runs(R);
        B.ops = marked.ops.data();
        if (G.fb_begin >= 0) {
            const int fe = G.fb_begin + G.fb_windows + 1;
            chain_binv(G.fb_begin, fe, false);
#pragma omp parallel for
            for (long long i = 0; i < (long long)n; i++) body_expand_fb_run<Emit, CV>(G, B, (size_t)i, fe - 1);
            expand(fe - 1, fe);
        }
        chain_binv(tb, te, false);
        expand(tb, te);
        for (int it = 0; it < iters; it += R) {
            const int it1 = std::min(it + R, iters);
            const int hi = it1 == iters ? G.num_ops : lb + 5 * it1;
            chain_binv(lb + 5 * it, hi, true);
#pragma omp parallel for
            for (long long i = 0; i < (long long)n; i++) body_expand_run<Emit, CV, 4>(G, B, (size_t)i, it, it1);
            if (it1 == iters) expand(le, G.num_ops);
        }
    } else {
        auto run_piece = [&](int lo, int hi, bool table) {
            chain_binv(lo, hi, table_done);
            expand(lo, hi);
            if (table) table_done = true;
        };
        auto cut = [&](int lo, int hi, bool table) {
            if (table) {
                run_piece(lo, hi, true);
                return;
            }
            for (int a2 = lo; a2 < hi; a2 += piece) run_piece(a2, a2 + piece < hi ? a2 + piece : hi, false);
        };
        if (tb >= 0) {
            cut(0, tb, false);
            cut(tb, te, true);
            cut(te, G.num_ops, false);
        } else {
            cut(0, G.num_ops, false);
        }
    }
    long bad = 0;
    for (size_t i = 0; i < n; i++) {
        err[i] = (uint8_t)err32[i];
        valid[i] = err32[i] ? 0 : valid8[i];
        bad += err32[i] != 0;
    }
    return bad;
}
extern "C" {
// kind / curve: include/p2e.h P2E_CP_* / P2E_CURVE_*; blind: the gadget's rand() point.  Returns the flagged count, or
// -1 for an unknown program; *num_cols receives the program's column count (call with n = 0 to query it).
long emu_curve_program(int kind, int curve, const uint8_t* blind_x, const uint8_t* blind_y, const uint8_t* msg, const uint8_t* r,
                       const uint8_t* s, const uint8_t* pkx, const uint8_t* pky, uint64_t* cols, size_t n, size_t ld, uint8_t* err,
                       uint8_t* valid, int piece, long* num_cols, long* num_gens, long* num_aux) {
    Aff blind;
    memcpy(blind.x.w, blind_x, 32);
    memcpy(blind.y.w, blind_y, 32);
    host::CurveProgramHost H;
    if (!host::make_curve_program(H, kind, curve, blind)) return -1;
    if (num_cols) *num_cols = H.sb.prog.num_cols;
    if (num_gens) *num_gens = (long)H.sb.gens.size();
    if (num_aux) *num_aux = (long)H.sb.aux_tab.num_aux_cols;
    if (n == 0) return 0;
    if (piece == 0) piece = 32;   // piece < 0: the run plan with -piece windows per run
    if (curve == 1) return run_curve<P256>(H, msg, r, s, pkx, pky, cols, n, ld, err, valid, piece);
    return run_curve<Secp256k1>(H, msg, r, s, pkx, pky, cols, n, ld, err, valid, piece);
}
// generator table (kind, field, first column, column count) and operand wiring of a curve program, for the host tests
long emu_curve_program_gens(int kind, int curve, const uint8_t* blind_x, const uint8_t* blind_y, int32_t* kinds, int32_t* fields,
                            uint32_t* first, uint32_t* ncols, uint32_t* src /*[4 per gen]*/, uint8_t* nl /*[4 per gen]*/, size_t cap) {
    Aff blind;
    memcpy(blind.x.w, blind_x, 32);
    memcpy(blind.y.w, blind_y, 32);
    host::CurveProgramHost H;
    if (!host::make_curve_program(H, kind, curve, blind)) return -1;
    const auto& g = H.sb.gens;
    for (size_t k = 0; k < g.size() && k < cap; k++) {
        kinds[k] = g[k].kind;
        fields[k] = g[k].field;
        first[k] = g[k].col;
        ncols[k] = g[k].ncols;
        for (int j = 0; j < 4; j++) {
            src[4 * k + j] = g[k].src[j];
            nl[4 * k + j] = g[k].nl[j];
        }
    }
    return (long)g.size();
}
// value of constant `id` of a curve program (source code AUX_SRC_CONST | id); 0, or -1 for an unknown id
int emu_curve_program_const(int kind, int curve, const uint8_t* blind_x, const uint8_t* blind_y, uint32_t id, uint8_t* out32) {
    Aff blind;
    memcpy(blind.x.w, blind_x, 32);
    memcpy(blind.y.w, blind_y, 32);
    host::CurveProgramHost H;
    if (!host::make_curve_program(H, kind, curve, blind)) return -1;
    id &= 0xFFFFu;
    if (id >= AUX_GCONST_BASE ? id - AUX_GCONST_BASE >= H.sb.gvals.size() : (id >> 1) >= H.sb.gpts.size()) return -1;
    const U256 v = H.sb.cval(id);
    memcpy(out32, v.w, 32);
    return 0;
}
// built-in-generator / gate-internal / constraint-block values of a curve program from finished matrices (the bodies
// kc_aux / k_gate / kc_ux run); each returns the matrix' column count (call with n = 0 to query it)
long emu_curve_aux(int kind, int curve, const uint8_t* blind_x, const uint8_t* blind_y, const uint8_t* msg, const uint8_t* r,
                   const uint8_t* s, const uint8_t* pkx, const uint8_t* pky, const uint64_t* cols, size_t ld, uint64_t* aux, size_t ald,
                   size_t n, uint8_t* err) {
    Aff blind;
    memcpy(blind.x.w, blind_x, 32);
    memcpy(blind.y.w, blind_y, 32);
    host::CurveProgramHost H;
    if (!host::make_curve_program(H, kind, curve, blind)) return -1;
    const host::ScheduleBuilder& sb = H.sb;
    if (n == 0) return (long)sb.aux_tab.num_aux_cols;
    std::vector<u32> err32(n);
    AuxArgs A{cols, ld, aux, ald, n, pky, sb.gpts.data(), sb.gfbtab.data(), sb.aux_items.data(), &sb.aux_tab, err32.data(), nullptr, 0, nullptr, {}};
    A.in[INPUT_PY] = pky;
    A.in[INPUT_PX] = pkx;
    A.in[INPUT_MSG] = msg;
    A.in[INPUT_R] = r ? r : msg;
    A.in[INPUT_S] = s ? s : msg;
    for (int item = 0; item < (int)sb.aux_items.size(); item++) {
#pragma omp parallel for
        for (long long i = 0; i < (long long)n; i++) body_aux_cv<Emit>(A, item, (size_t)i);
    }
    for (size_t i = 0; i < n; i++) err[i] = (uint8_t)err32[i];
    return (long)sb.aux_tab.num_aux_cols;
}
long emu_curve_gate(int kind, int curve, const uint8_t* blind_x, const uint8_t* blind_y, const uint64_t* aux, size_t ald, uint64_t* gate,
                    size_t gld, size_t n) {
    Aff blind;
    memcpy(blind.x.w, blind_x, 32);
    memcpy(blind.y.w, blind_y, 32);
    host::CurveProgramHost H;
    if (!host::make_curve_program(H, kind, curve, blind)) return -1;
    const host::ScheduleBuilder& sb = H.sb;
    if (n == 0) return (long)sb.num_gate_cols;
    GateArgs A{};
    A.aux = aux;
    A.ald = ald;
    A.gate = gate;
    A.gld = gld;
    A.n = n;
    A.items = sb.gate_items.data();
    A.inv16[0] = 0;
    for (u64 d = 1; d < 16; d++) {   // d^(p-2) in Goldilocks
        u64 acc = 1, base = d, e = P_GL - 2;
        while (e) {
            if (e & 1) acc = gl_mul(acc, base);
            base = gl_mul(base, base);
            e >>= 1;
        }
        A.inv16[d] = acc;
    }
    for (int item = 0; item < (int)sb.gate_items.size(); item++)
        for (size_t i = 0; i < n; i++) body_gate<Emit>(A, item, i);
    return (long)sb.num_gate_cols;
}
long emu_curve_ux(int kind, int curve, const uint8_t* blind_x, const uint8_t* blind_y, const uint8_t* msg, const uint8_t* r,
                  const uint8_t* s, const uint8_t* pkx, const uint8_t* pky, const uint64_t* cols, size_t ld, const uint64_t* aux,
                  size_t ald, uint64_t* ux, size_t uld, size_t n, uint8_t* err) {
    Aff blind;
    memcpy(blind.x.w, blind_x, 32);
    memcpy(blind.y.w, blind_y, 32);
    host::CurveProgramHost H;
    if (!host::make_curve_program(H, kind, curve, blind)) return -1;
    const host::ScheduleBuilder& sb = H.sb;
    if (n == 0) return (long)sb.num_ux_cols;
    std::vector<U256> cv(AUX_GCONST_BASE + sb.gvals.size(), u256_zero());
    for (size_t k = 0; k < sb.gpts.size(); k++) {
        cv[2 * k] = sb.gpts[k].x;
        cv[2 * k + 1] = sb.gpts[k].y;
    }
    for (size_t j = 0; j < sb.gvals.size(); j++) cv[AUX_GCONST_BASE + j] = sb.gvals[j];
    std::vector<u32> err32(n);
    UxArgs A{};
    A.cols = cols;
    A.ld = ld;
    A.aux = aux;
    A.ald = ald;
    A.ux = ux;
    A.uld = uld;
    A.n = n;
    A.in[INPUT_PY] = pky;
    A.in[INPUT_PX] = pkx;
    A.in[INPUT_MSG] = msg;
    A.in[INPUT_R] = r ? r : msg;
    A.in[INPUT_S] = s ? s : msg;
    A.consts = cv.data();
    A.items = sb.ux_items.data();
    A.err = err32.data();
#pragma omp parallel for
    for (long long item = 0; item < (long long)sb.ux_items.size(); item++)
        for (size_t i = 0; i < n; i++) body_ux_cv<Emit>(A, (int)item, i);
    for (size_t i = 0; i < n; i++) err[i] = (uint8_t)err32[i];
    return (long)sb.num_ux_cols;
}
// P-256 base field: the multiplication-free reduction of the chains (reduce_p256_solinas) against the Barrett reduction
// of the witness generators on `count` products of random and structured operands and on raw 512-bit values; returns
// the number of mismatches
long emu_p256_reduce_selfcheck(long count, uint64_t seed) {
    host::SplitMix64 rng{seed};
    long bad = 0;
    const u32 edge[6] = {0u, 1u, 0xFFFFFFFFu, 0xFFFFFFFEu, 0x80000000u, 0x7FFFFFFFu};
    for (long it = 0; it < count; it++) {
        u32 prod[16];
        if (it % 3 == 0) {   // a raw 512-bit value built from edge words and random words
            for (int k = 0; k < 16; k++) prod[k] = (rng.next() & 1) ? edge[rng.next() % 6] : (u32)rng.next();
        } else {             // a product of two field elements (canonical, or structured near 0 / p)
            U256 a, b;
            for (int k = 0; k < 8; k++) {
                a.w[k] = (it % 3 == 1) ? (u32)rng.next() : edge[rng.next() % 6];
                b.w[k] = (u32)rng.next();
            }
            a = fe_canon<ModP256>(a);
            b = fe_canon<ModP256>(b);
            if (it % 7 == 2)
                for (int k = 0; k < 8; k++) b.w[k] = ModP256::m(k) - (k == 0 ? 1u + (u32)(it & 3) : 0u);   // p - 1 .. p - 4
            mul_wide<8, 8>(a.w, b.w, prod);
        }
        u32 r1[8], r2[8];
        reduce_p256_solinas(prod, r1);
        reduce_barrett<ModP256, 8, false>(prod, r2, nullptr);
        for (int k = 0; k < 8; k++)
            if (r1[k] != r2[k]) {
                bad++;
                break;
            }
    }
    return bad;
}
// verdict only of the P-256 verifier program (p2e_p256_verify_batch): scalar phase without emission, the chains in
// Jacobian coordinates (table left Jacobian), r == x on the Jacobian result of the final add
long emu_p256_verify_only(const uint8_t* blind_x, const uint8_t* blind_y, const uint8_t* msg, const uint8_t* r, const uint8_t* s,
                          const uint8_t* pkx, const uint8_t* pky, size_t n, uint8_t* err, uint8_t* valid) {
    Aff blind;
    memcpy(blind.x.w, blind_x, 32);
    memcpy(blind.y.w, blind_y, 32);
    host::CurveProgramHost H;
    if (!host::make_curve_program(H, CP_VERIFY, 1, blind)) return -1;
    const host::ScheduleBuilder& sb = H.sb;
    const Program& G = sb.prog;
    const size_t rows = (size_t)(G.cp_rows > MSM_DIGITS ? G.cp_rows : MSM_DIGITS);
    std::vector<U256> PX((size_t)G.num_slots * n), PY((size_t)G.num_slots * n), PZ((size_t)G.num_slots * n),
        PW((size_t)G.num_ops * n), PREF((size_t)G.num_ops * n), AX((size_t)G.num_slots * n), AY((size_t)G.num_slots * n);
    std::vector<uint8_t> dig4((size_t)FB_WINDOWS * n), dig2(rows * n), valid8(n);
    std::vector<uint16_t> dyn((size_t)G.num_cadd * n + 1), src((size_t)G.num_ops * 2 * n), msrc(rows * n);
    std::vector<u32> err32(n);
    Buffers B{};
    B.msg = msg; B.r = r; B.s = s; B.pkx = pkx; B.pky = pky;
    B.n = n;
    B.err = err32.data(); B.valid = valid8.data();
    B.PX = PX.data(); B.PY = PY.data(); B.PZ = PZ.data(); B.PW = PW.data(); B.PREF = PREF.data();
    B.AX = AX.data(); B.AY = AY.data();
    B.dig4 = dig4.data(); B.dig2 = dig2.data(); B.dyn = dyn.data(); B.src = src.data(); B.msrc = msrc.data();
    B.cpts = sb.gpts.data(); B.fbtab = sb.gfbtab.data(); B.ops = sb.ops.data();
    const int fb_end = G.fb_begin + G.fb_windows + 1;
#pragma omp parallel for
    for (long long i = 0; i < (long long)n; i++) {
        body_cscalar<P256, NullEmit>(G, B, (size_t)i);
        body_chain_range<P256, true>(G, B, (size_t)i, G.fb_begin, fb_end, false, false);
        body_chain_range<P256, true>(G, B, (size_t)i, fb_end, G.num_ops - 1, false, false);
        body_chain_range<P256, true>(G, B, (size_t)i, G.num_ops - 1, G.num_ops, false, false);
        body_verify_check<P256>(G, B, (size_t)i, G.num_ops - 1);
    }
    long bad = 0;
    for (size_t i = 0; i < n; i++) {
        err[i] = (uint8_t)err32[i];
        valid[i] = err32[i] ? 0 : valid8[i];
        bad += err32[i] != 0;
    }
    return bad;
}
int emu_synth_signatures_curve(int curve, uint64_t seed, size_t first, size_t n, uint8_t* msg32, uint8_t* r32, uint8_t* s32,
                               uint8_t* pkx32, uint8_t* pky32) {
#pragma omp parallel for
    for (long long i = 0; i < (long long)n; i++) {
        U256 msg, r, s;
        Aff pk;
        if (curve == 1)
            host::synth_signature_cv<P256>(seed, first + (u64)i, msg, r, s, pk);
        else
            host::synth_signature_cv<Secp256k1>(seed, first + (u64)i, msg, r, s, pk);
        memcpy(msg32 + 32 * i, msg.w, 32);
        memcpy(r32 + 32 * i, r.w, 32);
        memcpy(s32 + 32 * i, s.w, 32);
        memcpy(pkx32 + 32 * i, pk.x.w, 32);
        memcpy(pky32 + 32 * i, pk.y.w, 32);
    }
    return 0;
}
}
