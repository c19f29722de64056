// Witness values of the plonky2-ecdsa non-native generators, one signature (one lane) at a time.
// Each wit_* function computes exactly what the corresponding reference run_once body would
// set_target, and emits the values in generator-registration order through an Emit cursor that walks
// the Goldilocks column matrix  out[col * ld + sig]  (column-major over the batch: a wavefront's 64
// lanes are 64 consecutive signatures, so every emit is one coalesced 512-byte store).
//
// Reference (paths relative to /root/reference/src):
//   wit_add       gadgets/nonnative.rs:626-645   NonNativeAdditionGenerator::run_once
//   wit_sub       gadgets/nonnative.rs:792-810   NonNativeSubtractionGenerator::run_once
//   wit_add_many  gadgets/nonnative.rs:696-728   NonNativeMultipleAddsGenerator::run_once
//   wit_inv       gadgets/nonnative.rs:857-872   NonNativeInverseGenerator::run_once
//   wit_mul       gates/mul_nonnative.rs:249-324 MulNonnativeGenerator::run_once
//                 gates/mul_nonnative.rs:513-531 CheckSumGenerator::run_once
//   glv_decompose curve/glv.rs:39-77 + gadgets/glv.rs:128-142
#pragma once
#include "fe.hpp"

namespace p2e {

// Where a fused kernel's witness values go: the u64 column matrix out[col * ld + sig], or the compact container of
// include/p2e.h (p2e_columns_compact): u32 nar[narrow index][ldn] for the columns that only hold limbs, overflow
// words and flags, u64 wid[wide index][ldw] for the check_sum / carry columns of the mul generators.  Both indices
// advance in registration order, so a cursor opened at column `col` starts at wide index wide_before[col] and narrow
// index col - wide_before[col].
struct Sink {
    u64* out;
    size_t ld;
    u32* nar;
    size_t ldn;
    u64* wid;
    size_t ldw;
    const u32* wide_before;   // [num_cols + 1]
};

// cursor over one signature's column of the output matrix: plain 8-byte stores (one per lane per column).
// put_wide / put_wide_at / skip_wide mark the values that need all 64 bits (check_sum, carries): the same
// stream here, a separate matrix in the compact emitters below.
struct Emit {
    typedef u64 elem;
    u64* p;      // &out[col * ld + sig]
    size_t ld;   // column stride in elements
    P2E_HD static Emit at(u64* out, size_t ld_, size_t sig, u32 col) {
        Emit e;
        e.p = out + (size_t)col * ld_ + sig;
        e.ld = ld_;
        return e;
    }
    P2E_HD static Emit at(const Sink& S, size_t sig, u32 col) { return at(S.out, S.ld, sig, col); }
    P2E_HD void put(u64 v) {
        *p = v;
        p += ld;
    }
    // store k columns ahead of the cursor without moving it
    P2E_HD void put_at(int k, u64 v) { p[(size_t)k * ld] = v; }
    P2E_HD void skip(int k) { p += (size_t)k * ld; }
    P2E_HD void put_wide(u64 v) { put(v); }
    P2E_HD void put_wide_at(int k, u64 v) { put_at(k, v); }
    P2E_HD void skip_wide(int k) { skip(k); }
    P2E_HD void flush() {}
};
// discards everything: the value computations that only feed witness columns disappear at compile time
// (p2e_ecdsa_verify_batch: the native verification as a pre-filter, no witness)
struct NullEmit {
    P2E_HD static NullEmit at(const Sink&, size_t, u32) { return NullEmit(); }
    P2E_HD void put(u64) {}
    P2E_HD void put_at(int, u64) {}
    P2E_HD void skip(int) {}
    P2E_HD void put_wide(u64) {}
    P2E_HD void put_wide_at(int, u64) {}
    P2E_HD void skip_wide(int) {}
    P2E_HD void flush() {}
};
// a u32 column matrix (values known to fit: the built-in-generator columns of aux.hpp)
struct Emit32 {
    typedef u32 elem;
    u32* p;
    size_t ld;
    P2E_HD static Emit32 at(u32* out, size_t ld_, size_t sig, u32 col) {
        Emit32 e;
        e.p = out + (size_t)col * ld_ + sig;
        e.ld = ld_;
        return e;
    }
    P2E_HD void put(u64 v) {
        *p = (u32)v;
        p += ld;
    }
    P2E_HD void flush() {}
};
// the same for the compact container: 4-byte stores into the narrow matrix, 8-byte stores into the wide one
struct CompactEmit {
    u32* pn;
    u64* pw;
    size_t ldn, ldw;
    P2E_HD static CompactEmit at(const Sink& S, size_t sig, u32 col) {
        CompactEmit e;
        const u32 wb = S.wide_before[col];
        e.pn = S.nar + (size_t)(col - wb) * S.ldn + sig;
        e.pw = S.wid + (size_t)wb * S.ldw + sig;
        e.ldn = S.ldn;
        e.ldw = S.ldw;
        return e;
    }
    P2E_HD void put(u64 v) {
        *pn = (u32)v;
        pn += ldn;
    }
    P2E_HD void put_wide(u64 v) {
        *pw = v;
        pw += ldw;
    }
    P2E_HD void put_wide_at(int k, u64 v) { pw[(size_t)k * ldw] = v; }
    P2E_HD void skip_wide(int k) { pw += (size_t)k * ldw; }
    P2E_HD void flush() {}
};

#if defined(__HIP_DEVICE_COMPILE__)
// Paired column stores.  The column stores of the fused kernels are store-ISSUE bound with 8 B per lane
// (MI355X_MICROARCH.md "store tail": ~7 B/clk/CU), so two consecutive columns c, c+1 are written with ONE
// store per lane: the wave's lanes l and l+32 own ADJACENT signatures (sig = base + 2*(l & 31) + (l >> 5)), a
// v_permlane32_swap exchanges "my column-c+1 value" of the lower half with "my column-c value" of the upper
// half, after which lower lanes hold column c of signatures (2l, 2l+1) and upper lanes column c+1 of the same
// two signatures: 2 contiguous elements each (global_store_dwordx4 for u64 columns, dwordx2 for the u32 columns
// of the compact container).  Requires EXEC = all ones (full waves), an even stride, a base aligned to two
// elements; the launcher falls back to the plain emitters otherwise.  All bookkeeping below (column counters,
// "have a pending value") is compile-time after inlining: every put sequence is static.
template <class T>
struct PStream {
    // All lane dependence lives in ONE running pointer that is bumped by the (wave-uniform) stride per column.
    // Computing `out + (c + upper) * ld + sig` per store instead costs two quarter-rate v_mad_u64_u32 each and,
    // inside the run-walking loop, makes the optimiser keep one hoisted VGPR per column constant of the body
    // (349 registers, one wave per SIMD).
    T* two;           // this lane's 2-element slot of the column pair (col - 1, col): lags the cursor by one column
    ptrdiff_t delta;  // this lane's 1-element slot of column c is (its 2-element slot of pair (c, c+1)) + delta
    size_t ld;
    int col;          // cursor, relative to the opening column
    bool have, have2;
    T pend, pend2;
    int pcol, pcol2;
    P2E_HD static PStream at(T* col0 /* &m[opening column * ld] */, size_t ld_, size_t sig_) {
        PStream e;
        const size_t upper = sig_ & 1;   // the lane mapping makes the signature's parity the half-wave index
        // lower half-wave: column c of signatures (sig, sig+1); upper: column c+1 of (sig-1, sig)
        uintptr_t first = (uintptr_t)(col0 + (upper ? ld_ : 0) + (sig_ - upper));
        e.two = (T*)(first - sizeof(T) * ld_);
        e.delta = upper ? (ptrdiff_t)1 - (ptrdiff_t)ld_ : 0;
        e.ld = ld_;
        e.col = 0;
        e.have = e.have2 = false;
        e.pend = e.pend2 = 0;
        e.pcol = e.pcol2 = 0;
        return e;
    }
    P2E_HD T* slot2(int c) const {   // c - (col - 1) is a compile-time constant at every call site
        const int d = c - (col - 1);
        return d == 0 ? two : (T*)((uintptr_t)two + (intptr_t)d * (intptr_t)(sizeof(T) * ld));
    }
    P2E_HD void bump(int k) {
        two = (T*)((uintptr_t)two + (size_t)k * sizeof(T) * ld);
        asm("" : "+v"(two));   // keep it a running pointer: do not re-derive first + c * ld
        col += k;
    }
    // Output columns are write-once / never re-read by the pipeline: non-temporal stores keep them from
    // displacing the scratch arrays that phases B and C are about to read (-1.6 % on the whole step).
    // The running pointer went through an asm barrier, which hides from the compiler that it points into global
    // memory: without the explicit address-space casts below every column store is a FLAT store (counted in lgkmcnt
    // as well, so each scalar-load wait also waits for the stores).
    P2E_HD void store_single(int c, T v) {
        typedef T __attribute__((address_space(1))) * gptr;
        __builtin_nontemporal_store(v, (gptr)(slot2(c) + delta));
    }
    P2E_HD void store_pair(int c, T a, T b) {   // a: my value of column c, b: of column c + 1
        if (sizeof(T) == 8) {
            u32 ax = (u32)a, ay = (u32)((u64)a >> 32), bx = (u32)b, by = (u32)((u64)b >> 32);
            auto r0 = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
            auto r1 = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
            typedef unsigned int v4u __attribute__((ext_vector_type(4)));
            v4u ov = {r0[0], r1[0], r0[1], r1[1]};
            typedef v4u __attribute__((address_space(1))) * gptr;
            __builtin_nontemporal_store(ov, (gptr)slot2(c));
        } else {
            auto r0 = __builtin_amdgcn_permlane32_swap((u32)a, (u32)b, false, false);
            typedef unsigned int v2u __attribute__((ext_vector_type(2)));
            v2u ov = {r0[0], r0[1]};
            typedef v2u __attribute__((address_space(1))) * gptr;
            __builtin_nontemporal_store(ov, (gptr)slot2(c));
        }
    }
    P2E_HD void put(T v) {
        if (have && pcol + 1 == col) {
            store_pair(pcol, pend, v);
            have = false;
        } else {
            if (have) store_single(pcol, pend);
            pend = v;
            pcol = col;
            have = true;
        }
        bump(1);
    }
    P2E_HD void put_at(int k, T v) {
        const int c = col + k;
        if (have2 && pcol2 + 1 == c) {
            store_pair(pcol2, pend2, v);
            have2 = false;
        } else {
            if (have2) store_single(pcol2, pend2);
            pend2 = v;
            pcol2 = c;
            have2 = true;
        }
    }
    P2E_HD void skip(int k) { bump(k); }
    P2E_HD void flush() {
        if (have) store_single(pcol, pend);
        if (have2) store_single(pcol2, pend2);
        have = have2 = false;
    }
};
// u64 column matrix, 16-byte stores
struct PairEmit {
    typedef u64 elem;
    PStream<u64> s;
    P2E_HD static PairEmit at(u64* out, size_t ld_, size_t sig_, u32 col0) {
        PairEmit e;
        e.s = PStream<u64>::at(out + (size_t)col0 * ld_, ld_, sig_);
        return e;
    }
    P2E_HD static PairEmit at(const Sink& S, size_t sig, u32 col) { return at(S.out, S.ld, sig, col); }
    P2E_HD void put(u64 v) { s.put(v); }
    P2E_HD void put_at(int k, u64 v) { s.put_at(k, v); }
    P2E_HD void skip(int k) { s.skip(k); }
    P2E_HD void put_wide(u64 v) { s.put(v); }
    P2E_HD void put_wide_at(int k, u64 v) { s.put_at(k, v); }
    P2E_HD void skip_wide(int k) { s.skip(k); }
    P2E_HD void flush() { s.flush(); }
};
// u32 column matrix, 8-byte stores of two u32
struct PairEmit32 {
    typedef u32 elem;
    PStream<u32> s;
    P2E_HD static PairEmit32 at(u32* out, size_t ld_, size_t sig_, u32 col0) {
        PairEmit32 e;
        e.s = PStream<u32>::at(out + (size_t)col0 * ld_, ld_, sig_);
        return e;
    }
    P2E_HD void put(u64 v) { s.put((u32)v); }
    P2E_HD void flush() { s.flush(); }
};
// compact container: 8-byte stores of two u32 into the narrow matrix, 16-byte stores into the wide one
struct CompactPairEmit {
    PStream<u32> n;
    PStream<u64> w;
    P2E_HD static CompactPairEmit at(const Sink& S, size_t sig, u32 col) {
        CompactPairEmit e;
        const u32 wb = S.wide_before[col];
        e.n = PStream<u32>::at(S.nar + (size_t)(col - wb) * S.ldn, S.ldn, sig);
        e.w = PStream<u64>::at(S.wid + (size_t)wb * S.ldw, S.ldw, sig);
        return e;
    }
    P2E_HD void put(u64 v) { n.put((u32)v); }
    P2E_HD void put_wide(u64 v) { w.put(v); }
    P2E_HD void put_wide_at(int k, u64 v) { w.put_at(k, v); }
    P2E_HD void skip_wide(int k) { w.skip(k); }
    P2E_HD void flush() {
        n.flush();
        w.flush();
    }
};
#else
typedef Emit PairEmit;   // host passes only parse the kernels that name them
typedef CompactEmit CompactPairEmit;
typedef Emit32 PairEmit32;
#endif

template <class E>
P2E_HD void emit_limbs(E& e, const u32* l, int n) {
    P2E_UNROLL
    for (int i = 0; i < NL; i++)
        if (i < n) e.put((u64)l[i]);
}
template <class E>
P2E_HD void emit_u256(E& e, const U256& v) {
    u32 l[NL];
    split29(v, l);
    emit_limbs(e, l, NL);
}

// ---- add: returns the sum target value (== m possible, reference quirk Q1: strict '>') ----------
template <class MOD, class E>
P2E_HD U256 wit_add(E& e, const U256& a_raw, const U256& b_raw) {
    U256 a = fe_canon<MOD>(a_raw), b = fe_canon<MOD>(b_raw);
    U256 s;
    u32 c = add_n<8>(s.w, a.w, b.w);
    // s_total = c*2^256 + s ; overflow iff s_total > m  (strictly)
    bool gt = c != 0;
    if (!gt) {
        bool ge = geq_mod<MOD>(s.w);
        bool eq = true;
        P2E_UNROLL
        for (int i = 0; i < 8; i++) eq = eq && (s.w[i] == MOD::m(i));
        gt = ge && !eq;
    }
    if (gt) sub_mod_raw<MOD>(s.w);
    emit_u256(e, s);
    e.put(gt ? 1u : 0u);
    return s;
}

// ---- sub ------------------------------------------------------------------------------------------
template <class MOD, class E>
P2E_HD U256 wit_sub(E& e, const U256& a_raw, const U256& b_raw) {
    U256 a = fe_canon<MOD>(a_raw), b = fe_canon<MOD>(b_raw);
    U256 d;
    u32 br = sub_n<8>(d.w, a.w, b.w);
    if (br) add_mod_raw<MOD>(d.w);
    emit_u256(e, d);
    e.put(br ? 1u : 0u);
    return d;
}

// ---- add_many: sum of k canonicalised summands, true div_rem by m -----------------------------------
template <class MOD, int K, class E>
P2E_HD U256 wit_add_many(E& e, const U256* xs) {
    u32 acc[9];
    P2E_UNROLL
    for (int i = 0; i < 9; i++) acc[i] = 0;
    P2E_UNROLL
    for (int k = 0; k < K; k++) {
        U256 v = fe_canon<MOD>(xs[k]);
        u64 c = 0;
        P2E_UNROLL
        for (int i = 0; i < 9; i++) {
            c += (u64)acc[i] + (i < 8 ? v.w[i] : 0u);
            acc[i] = (u32)c;
            c >>= 32;
        }
    }
    u32 ov = 0;
    P2E_UNROLL
    for (int k = 0; k < K; k++) {  // quotient < K
        bool ge = acc[8] != 0 || geq_mod<MOD>(acc);
        if (ge) {
            u32 br = 0;
            P2E_UNROLL
            for (int i = 0; i < 9; i++) {
                u64 d = (u64)acc[i] - (i < 8 ? MOD::m(i) : 0u) - br;
                acc[i] = (u32)d;
                br = (u32)(d >> 63);
            }
            ov++;
        }
    }
    U256 s;
    P2E_UNROLL
    for (int i = 0; i < 8; i++) s.w[i] = acc[i];
    emit_u256(e, s);
    e.put(ov);
    return s;
}

// ---- inv: emits inv[9], div[9] where x*inv = div*m + 1.  `inv` is supplied by the caller (batched
// inversion) and must be the canonical inverse of canon(x_raw). ------------------------------------
template <class MOD, class E>
P2E_HD void wit_inv_given(E& e, const U256& x_raw, const U256& inv, uint8_t& err) {
    U256 x = fe_canon<MOD>(x_raw);
    if (u256_is_zero(x)) err |= ERR_INVERSE_OF_ZERO;
    u32 prod[16], q[9], r[8];
    mul_wide<8, 8>(x.w, inv.w, prod);
    reduce16<MOD, true>(prod, r, q);
    emit_u256(e, inv);
    u32 ql[NL];
    split29_9(q, ql);
    emit_limbs(e, ql, NL);
}
// stand-alone form: one Fermat ladder per call
template <class MOD, class E>
P2E_HD U256 wit_inv(E& e, const U256& x_raw, uint8_t& err) {
    U256 x = fe_canon<MOD>(x_raw);
    U256 inv = fe_inv<MOD>(x);
    wit_inv_given<MOD>(e, x, inv, err);
    return inv;
}

// ---- mul + checksum ---------------------------------------------------------------------------------
// un-carried q*m convolution column i, exploiting m29[j] = (2^29-1) - d[j] for p
template <class MOD>
P2E_HD u64 qm_column(const u32* q29, int i) {
    u64 acc = 0;
    P2E_UNROLL
    for (int j = 0; j < NL; j++) {
        int k = i - j;
        if (k >= 0 && k < NL) acc += (u64)q29[k] * MOD::m29(j);
    }
    return acc;
}
template <>
P2E_HD u64 qm_column<ModP>(const u32* q29, int i) {
    // sum_j q[i-j]*(M - d[j]) with M = 2^29-1, d = {976, 8, 0,0,0,0,0,0, 0x1F000000}
    u64 w = 0;
    P2E_UNROLL
    for (int j = 0; j < NL; j++) {
        int k = i - j;
        if (k >= 0 && k < NL) w += q29[k];
    }
    u64 acc = (w << BITS) - w;
    if (i < NL) acc -= (u64)q29[i] * 976u;
    if (i - 1 >= 0 && i - 1 < NL) acc -= (u64)q29[i - 1] * 8u;
    if (i - 8 >= 0 && i - 8 < NL) acc -= (u64)q29[i - 8] * 0x1F000000u;
    return acc;
}

// core: given 29-bit limbs of x, y (gate wires) and the quotient/remainder, emit r, q, cs, b
template <class MOD, class E>
P2E_HD void emit_mul_rows(E& e, const u32* x29, const u32* y29, const u32* q29, const u32* r29, uint8_t& err) {
    emit_limbs(e, r29, NL);
    emit_limbs(e, q29, NL);
    i64 last = 0;
    // check_sum[i] goes to the cursor, the carry b[i] 17 columns further (CheckSumGate row): stored as soon
    // as it is known, so nothing has to stay live across the 17 convolution columns
    P2E_UNROLL
    for (int i = 0; i < 2 * NL - 1; i++) {
        u64 xy = 0;
        P2E_UNROLL
        for (int j = 0; j < NL; j++) {
            int k = i - j;
            if (k >= 0 && k < NL) xy += (u64)x29[j] * y29[k];
        }
        i64 cs = (i64)(qm_column<MOD>(q29, i) - xy) + (i < NL ? (i64)r29[i] : 0);
        e.put_wide(gl_from_i64(cs));
        if (i < 2 * NL - 2) {
            i64 t = cs + last;
            i64 bi = t >> BITS;  // exact: the integer is a multiple of 2^29 whenever q, r are right
            u64 bo = (u64)(bi + ((i64)1 << 33));
            if (bo >> 34) err |= ERR_CARRY_RANGE;
            e.put_wide_at(2 * NL - 2, bo);   // cursor already advanced past check_sum[i]: b[i] is 16 columns ahead
            last = bi;
        }
    }
    e.skip_wide(2 * NL - 2);
}

// x, y: the values carried by the two operands' limbs (used RAW: reference quirk Q3), < 2^256
template <class MOD, class E>
P2E_HD U256 wit_mul(E& e, const U256& x, const U256& y, uint8_t& err) {
    u32 prod[16], q[9];
    U256 r;
    mul_wide<8, 8>(x.w, y.w, prod);
    reduce16<MOD, true>(prod, r.w, q);
    u32 x29[NL], y29[NL], q29[NL], r29[NL];
    split29(x, x29);
    split29(y, y29);
    split29(r, r29);
    split29_9(q, q29);
    emit_mul_rows<MOD>(e, x29, y29, q29, r29, err);
    return r;
}

// ---- nonnative_conditional_neg (gadgets/nonnative.rs:584-596): sub(0, x) then add(neg*b, x*!b) ----
template <class MOD, class E>
P2E_HD U256 wit_cond_neg(E& e, const U256& x, u32 b) {
    U256 neg = wit_sub<MOD>(e, u256_zero(), x);
    const U256 z = u256_zero();
    U256 t = u256_select(b != 0, neg, z);
    U256 f = u256_select(b != 0, z, x);
    return wit_add<MOD>(e, t, f);
}

// ---- GLV decomposition --------------------------------------------------------------------------------
struct GlvOut {
    U256 k1, k2;   // |k1|, |k2|
    u32 n1, n2;    // signs
};
P2E_HD U256 glv_round_div(const U256& k, const u32* c4 /*4 words*/) {
    // round(c*k / n) with num Ratio::round semantics for odd n: +1 iff 2*rem > n
    u32 prod[12], q[9], r[8];
    mul_wide<8, 4>(k.w, c4, prod);
    reduce_wide<ModN, 4, true>(prod, r, q);
    // 2*rem > n  <=>  rem > (n-1)/2  <=>  rem >= (n+1)/2
    u32 r2[9];
    u32 cw = 0;
    P2E_UNROLL
    for (int i = 0; i < 8; i++) {
        r2[i] = (r[i] << 1) | cw;
        cw = r[i] >> 31;
    }
    r2[8] = cw;
    bool gt = r2[8] != 0;
    if (!gt) {
        bool ge = geq_mod<ModN>(r2);
        bool eq = true;
        P2E_UNROLL
        for (int i = 0; i < 8; i++) eq = eq && (r2[i] == ModN::m(i));
        gt = ge && !eq;
    }
    U256 out;
    u64 c = gt ? 1 : 0;
    P2E_UNROLL
    for (int i = 0; i < 8; i++) {
        c += q[i];
        out.w[i] = (u32)c;
        c >>= 32;
    }
    return out;
}
P2E_HD GlvOut glv_decompose(const U256& k_raw) {
    // constants as u64 pairs from curve/glv.rs:25-32
    const u32 a1[8] = {(u32)16747920425669159701ull, (u32)(16747920425669159701ull >> 32),
                       (u32)3496713202691238861ull, (u32)(3496713202691238861ull >> 32), 0, 0, 0, 0};
    const u32 mb1[8] = {(u32)8022177200260244675ull, (u32)(8022177200260244675ull >> 32),
                        (u32)16448129721693014056ull, (u32)(16448129721693014056ull >> 32), 0, 0, 0, 0};
    const u32 a2[8] = {(u32)6323353552219852760ull, (u32)(6323353552219852760ull >> 32),
                       (u32)1498098850674701302ull, (u32)(1498098850674701302ull >> 32), 1, 0, 0, 0};
    U256 k = fe_canon<ModN>(k_raw);
    U256 c1 = glv_round_div(k, a1 /* B2 == A1 */);
    U256 c2 = glv_round_div(k, mb1);
    U256 A1v, MB1v, A2v;
    P2E_UNROLL
    for (int i = 0; i < 8; i++) {
        A1v.w[i] = a1[i];
        MB1v.w[i] = mb1[i];
        A2v.w[i] = a2[i];
    }
    U256 k1 = fe_sub<ModN>(fe_sub<ModN>(k, fe_mul<ModN>(c1, A1v)), fe_mul<ModN>(c2, A2v));
    U256 k2 = fe_sub<ModN>(fe_mul<ModN>(c1, MB1v), fe_mul<ModN>(c2, A1v));
    // half = n / 2 (integer division)
    u32 half[8];
    P2E_UNROLL
    for (int i = 0; i < 8; i++) half[i] = (ModN::m(i) >> 1) | (i < 7 ? (ModN::m(i + 1) << 31) : 0u);
    GlvOut o;
    // k > half  <=>  !(half >= k)
    o.n1 = geq_n<8>(half, k1.w) ? 0u : 1u;
    o.n2 = geq_n<8>(half, k2.w) ? 0u : 1u;
    o.k1 = o.n1 ? fe_sub<ModN>(u256_zero(), k1) : k1;
    o.k2 = o.n2 ? fe_sub<ModN>(u256_zero(), k2) : k2;
    return o;
}

}  // namespace p2e
