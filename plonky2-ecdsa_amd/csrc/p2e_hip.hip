// libp2e_hip.so: gfx950 kernels + the C ABI of include/p2e.h.
// One process drives one GPU; every entry point enqueues on the context's stream.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/p2e.h"
#include "aux.hpp"
#include "consts.hpp"
#include "pipeline.hpp"
#include "prims.hpp"
#include "quad.hpp"
#include "schedule.hpp"
#include "curve_program.hpp"

using namespace p2e;

// The library is ONE source compiled as four translation units in parallel (plonky2_ecdsa_amd.build: -DP2E_PART=0..3),
// because a single hipcc job over all kernels takes five minutes:  0 = context, single generators, layout helpers,
// streaming passes, descriptions;  1 = the fused pipeline of the two built-in programs (run_program);  2 / 3 = the
// curve-program pipeline instantiated for secp256k1 / P-256 (run_curve_program<CV>).  Undefined = everything in one
// unit.  Small static helpers (error text, staging, scratch) are compiled into every part.
#ifndef P2E_PART
#define P2E_PART (-1)
#endif
#define P2E_HAS(p) (P2E_PART < 0 || P2E_PART == (p))

// ====================================================================================================
// kernels
// ====================================================================================================
constexpr int BS = 256;          // 4 waves per workgroup

// Signature owned by this lane.  WIDE kernels (full workgroups only) give lanes l and l+32 of a wave
// adjacent signatures so that column pairs can be written with 16-byte stores (PairEmit); the narrow
// variants cover the ragged tail of the batch (first = first signature of the launch).
template <bool WIDE>
__device__ __forceinline__ size_t lane_sig(size_t first) {
    unsigned t = threadIdx.x;
    // (An XCD-contiguous remap of blockIdx -- every XCD owning one contiguous eighth of the batch -- was
    // measured 2.5 % SLOWER on k_expand than the plain round-robin order and is not used.)
    size_t base = first + (size_t)blockIdx.x * BS;
    if (WIDE) return base + (t & ~63u) + 2u * (t & 31u) + ((t >> 5) & 1u);
    return base + t;
}
// MODE of the emitting kernels: bit 0 = paired stores over full workgroups (else one signature per lane, ragged tail),
// bit 1 = compact container (u32 narrow + u64 wide matrices) instead of the u64 column matrix
template <int MODE> struct EmitOf;
template <> struct EmitOf<0> { typedef Emit type; };
template <> struct EmitOf<1> { typedef PairEmit type; };
template <> struct EmitOf<2> { typedef CompactEmit type; };
template <> struct EmitOf<3> { typedef CompactPairEmit type; };
template <> struct EmitOf<4> { typedef NullEmit type; };   // no witness: p2e_ecdsa_verify_batch
#if P2E_HAS(1)
template <int MODE>
__global__ __launch_bounds__(BS) void k_scalar(Program G, Buffers B, size_t first) {
    size_t i = lane_sig<(MODE & 1) != 0>(first);
    if ((MODE & 1) || i < B.n) body_scalar<typename EmitOf<MODE>::type>(G, B, i);
}
// ops [lo, hi) of a chain, sequential per lane
__global__ __launch_bounds__(BS) void k_chains(Program G, Buffers B, int lo, int hi, int table_affine, int continue_prefix) {
    // the chain is the critical path of the whole call and shares its SIMD with phase B/C waves of
    // earlier pieces: win every issue arbitration against them
    __builtin_amdgcn_s_setprio(3);
    size_t i = (size_t)blockIdx.x * BS + threadIdx.x;
#ifdef P2E_LDS_FBTAB
    // experiment: the fixed-base table rows of this piece's windows staged in LDS
    constexpr int MAXW = 36;
    __shared__ Aff s_fb[MAXW * 16];
    const OpDesc first = load_op(B.ops, lo);
    const bool fb = first.kind != OP_DBL && ref_kind(first.ref2) == R_FBTAB && hi - lo <= MAXW;
    const u32 w0 = ref_id(first.ref2);
    if (fb) {
        const int nwin = (hi - lo) < (int)(FB_WINDOWS - w0) ? (hi - lo) : (int)(FB_WINDOWS - w0);
        const u32* src = reinterpret_cast<const u32*>(B.fbtab + (size_t)w0 * 16);
        u32* dst = reinterpret_cast<u32*>(s_fb);
        for (int k = threadIdx.x; k < nwin * 16 * 16; k += BS) dst[k] = src[k];
    }
    __syncthreads();
    if (i < B.n) body_chain_range(G, B, i, lo, hi, table_affine != 0, continue_prefix != 0, fb ? s_fb : nullptr, w0);
#else
    if (i < B.n) body_chain_range(G, B, i, lo, hi, table_affine != 0, continue_prefix != 0);
#endif
}
// small batches: four lanes per signature (quad.hpp); the four lanes of a quad are consecutive threads
__global__ __launch_bounds__(BS) void k_chains_quad(Program G, Buffers B, int lo, int hi, int table_affine, int continue_prefix) {
    __builtin_amdgcn_s_setprio(3);
    const size_t g = (size_t)blockIdx.x * BS + threadIdx.x;
    const size_t i = g >> 2;
    if (i < B.n) body_chain_range_quad(G, B, i, (int)(g & 3), lo, hi, table_affine != 0, continue_prefix != 0);
}
// one inversion batch cut into 2^split_log2 sub-ranges, one lane each
__global__ __launch_bounds__(BS) void k_batch_inv_split(Program G, Buffers B, int lo, int hi, int have_prefix, int split_log2) {
    const size_t g = (size_t)blockIdx.x * BS + threadIdx.x;
    const size_t i = g >> split_log2;
    if (i < B.n) body_batch_inv_split(G, B, i, lo, hi, have_prefix != 0, (int)(g & ((1u << split_log2) - 1)), 1 << split_log2);
}
// independent interleaved sub-chains of a piece, one per blockIdx.y (the MSM window table)
__global__ __launch_bounds__(BS) void k_chain_rows(Program G, Buffers B, int lo, int count) {
    __builtin_amdgcn_s_setprio(3);
    size_t i = (size_t)blockIdx.x * BS + threadIdx.x;
    if (i < B.n) body_chain_rows(G, B, i, lo, (int)gridDim.y, (int)blockIdx.y, count);
}
// one inversion batch: ops [lo, hi); have_prefix: phase A has already left its prefix products in PREF
__global__ __launch_bounds__(BS) void k_batch_inv(Program G, Buffers B, int lo, int hi, int have_prefix) {
    size_t i = (size_t)blockIdx.x * BS + threadIdx.x;
    if (i < B.n) body_batch_inv(G, B, i, lo, hi, have_prefix != 0);
}
// op lo + blockIdx.y
template <int MODE>
__global__ __launch_bounds__(BS) void k_expand(Program G, Buffers B, int lo, size_t first) {
#ifdef P2E_EXPAND_PRIO
    __builtin_amdgcn_s_setprio(P2E_EXPAND_PRIO);   // experiment: issue priority of the expansion waves
#endif
    size_t i = lane_sig<(MODE & 1) != 0>(first);
#ifdef P2E_LDS_FBTAB
    // experiment: the 16 table entries of this op's window staged in LDS (one op per workgroup row)
    __shared__ Aff s_fb[16];
    const OpDesc op0 = load_op(B.ops, lo + (int)blockIdx.y);
    const bool fb = op0.kind != OP_DBL && ref_kind(op0.ref2) == R_FBTAB;
    if (fb) reinterpret_cast<u32*>(s_fb)[threadIdx.x] = reinterpret_cast<const u32*>(B.fbtab + (size_t)ref_id(op0.ref2) * 16)[threadIdx.x];
    __syncthreads();
    if ((MODE & 1) || i < B.n) body_expand<typename EmitOf<MODE>::type>(G, B, i, lo + (int)blockIdx.y, fb ? s_fb : nullptr);
#else
    if ((MODE & 1) || i < B.n) body_expand<typename EmitOf<MODE>::type>(G, B, i, lo + (int)blockIdx.y);
#endif
}
// runs of MSM-loop iterations: run blockIdx.y of the launch covers iterations [it_first + y*R, +R) capped at it_end
template <int MODE>
__global__ __launch_bounds__(BS) void k_expand_runs(Program G, Buffers B, int it_first, int run_iters, int it_end, size_t first) {
#ifdef P2E_EXPAND_PRIO
    __builtin_amdgcn_s_setprio(P2E_EXPAND_PRIO);
#endif
    size_t i = lane_sig<(MODE & 1) != 0>(first);
    int it0 = it_first + (int)blockIdx.y * run_iters;
    int it1 = it0 + run_iters < it_end ? it0 + run_iters : it_end;
    if ((MODE & 1) || i < B.n) body_expand_run<typename EmitOf<MODE>::type>(G, B, i, it0, it1);
}
// all fixed-base windows of a signature as one run (body_expand_fb_run); t_after: the unblinding add
template <int MODE>
__global__ __launch_bounds__(BS) void k_expand_fb_run(Program G, Buffers B, int t_after, size_t first) {
    size_t i = lane_sig<(MODE & 1) != 0>(first);
    if ((MODE & 1) || i < B.n) body_expand_fb_run<typename EmitOf<MODE>::type>(G, B, i, t_after);
}
// p2e_ecdsa_verify_batch: the connect r == x of gadgets/ecdsa.rs:48-52 on the final add's Jacobian result, without an
// inversion: x = X / Z^2 is canonical, so x == r  <=>  r < p and X == r * Z^2 (Z != 0, else phase A flagged the element)
__global__ __launch_bounds__(BS) void k_verify_check(Program G, Buffers B) {
    size_t i = (size_t)blockIdx.x * BS + threadIdx.x;
    if (i < B.n) body_verify_check(G, B, i);
}
#endif   // P2E_HAS(1)
// built-in-generator columns from the finished witness matrix: one (signature, item) per lane (aux.hpp)
// MODE bit 0 = paired stores over full workgroups, bit 1 = u32 output matrix
template <int MODE> struct AuxEmitOf;
template <> struct AuxEmitOf<0> { typedef Emit type; };
template <> struct AuxEmitOf<1> { typedef PairEmit type; };
template <> struct AuxEmitOf<2> { typedef Emit32 type; };
template <> struct AuxEmitOf<3> { typedef PairEmit32 type; };
#if P2E_HAS(0)
template <int MODE>
__global__ __launch_bounds__(BS) void k_aux(AuxArgs A, size_t first) {
    size_t i = lane_sig<(MODE & 1) != 0>(first);
    if ((MODE & 1) || i < A.n) body_aux<typename AuxEmitOf<MODE>::type>(A, (int)blockIdx.y, i);
}
// gate-internal values of the built-in gates, one (signature, window) per lane (aux.hpp body_gate); u64 only (the
// equality gadget's inverse is a full Goldilocks element)
template <int MODE>
__global__ __launch_bounds__(BS) void k_gate(GateArgs A, size_t first) {
    size_t i = lane_sig<(MODE & 1) != 0>(first);
    if ((MODE & 1) || i < A.n) body_gate<typename AuxEmitOf<MODE>::type>(A, (int)blockIdx.y, i);
}
// constraint-block columns from the finished matrices: one (signature, generator) per lane (ux.hpp)
template <int MODE>
__global__ __launch_bounds__(BS) void k_ux(UxArgs A, size_t first) {
    size_t i = lane_sig<(MODE & 1) != 0>(first);
    if ((MODE & 1) || i < A.n) body_ux<typename AuxEmitOf<MODE>::type>(A, (int)blockIdx.y, i);
}
#endif   // P2E_HAS(0)
// err words -> caller's err bytes, valid bytes, flagged count (every part launches it: internal linkage)
static __global__ __launch_bounds__(BS) void k_finalize(const u32* err32, const uint8_t* valid8, uint8_t* err_out,
                                                 uint8_t* valid_out, size_t n, unsigned long long* counter) {
    size_t i = (size_t)blockIdx.x * BS + threadIdx.x;
    bool bad = false;
    if (i < n) {
        u32 e = err32[i];
        bad = e != 0;
        if (err_out) err_out[i] = (uint8_t)e;
        if (valid_out) valid_out[i] = bad ? 0 : valid8[i];
    }
    unsigned long long m = __ballot(bad);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(counter, (unsigned long long)__popcll(m));
}

#if P2E_HAS(0)
// one empty wave: makes the runtime bind a hardware queue to a stream NOW (p2e_ctx_create), see there
__global__ void k_touch() {}
__device__ __forceinline__ void count_err(uint8_t e, unsigned long long* counter) {
    unsigned long long m = __ballot(e != 0);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(counter, (unsigned long long)__popcll(m));
}
template <class MOD>
__global__ __launch_bounds__(BS) void k_mul(const u64* x, const u64* y, u64* r, u64* q, u64* cs, u64* b, size_t n,
                                            size_t ld, uint8_t* err, unsigned long long* counter) {
    size_t i = (size_t)blockIdx.x * BS + threadIdx.x;
    uint8_t e = 0;
    if (i < n) {
        e = prim_mul<MOD>(x, y, r, q, cs, b, ld, i);
        err[i] = e;
    }
    count_err(e, counter);
}
__global__ __launch_bounds__(BS) void k_checksum(const u64* a, u64* b, size_t n, size_t ld, uint8_t* err,
                                                 unsigned long long* counter) {
    size_t i = (size_t)blockIdx.x * BS + threadIdx.x;
    uint8_t e = 0;
    if (i < n) {
        e = prim_checksum(a, b, ld, i);
        err[i] = e;
    }
    count_err(e, counter);
}
template <class MOD, bool IS_SUB>
__global__ __launch_bounds__(BS) void k_addsub(const u64* a, const u64* b, u64* out, u64* ov, size_t n, size_t ld,
                                               uint8_t* err, unsigned long long* counter) {
    size_t i = (size_t)blockIdx.x * BS + threadIdx.x;
    uint8_t e = 0;
    if (i < n) {
        e = prim_addsub<MOD, IS_SUB>(a, b, out, ov, ld, i);
        err[i] = e;
    }
    count_err(e, counter);
}
template <class MOD>
__global__ __launch_bounds__(BS) void k_add_many(const u64* s, int k, u64* out, u64* ov, size_t n, size_t ld,
                                                 uint8_t* err, unsigned long long* counter) {
    size_t i = (size_t)blockIdx.x * BS + threadIdx.x;
    uint8_t e = 0;
    if (i < n) {
        e = prim_add_many<MOD>(s, k, out, ov, ld, i);
        err[i] = e;
    }
    count_err(e, counter);
}
template <class MOD>
__global__ __launch_bounds__(BS) void k_inv(const u64* x, u64* inv, u64* div, size_t n, size_t ld, uint8_t* err,
                                            unsigned long long* counter) {
    size_t i = (size_t)blockIdx.x * BS + threadIdx.x;
    uint8_t e = 0;
    if (i < n) {
        e = prim_inv<MOD>(x, inv, div, ld, i);
        err[i] = e;
    }
    count_err(e, counter);
}
__global__ __launch_bounds__(BS) void k_div_rem(const u64* a, int na, const u64* b, int nb, u64* div, u64* rem, size_t n, size_t ld,
                                                uint8_t* err, unsigned long long* counter) {
    size_t i = (size_t)blockIdx.x * BS + threadIdx.x;
    uint8_t e = 0;
    if (i < n) {
        e = prim_div_rem(a, na, b, nb, div, rem, ld, i);
        err[i] = e;
    }
    count_err(e, counter);
}
__global__ __launch_bounds__(BS) void k_glv(const u64* k, u64* k1, u64* k2, u64* n1, u64* n2, size_t n, size_t ld,
                                            uint8_t* err, unsigned long long* counter) {
    size_t i = (size_t)blockIdx.x * BS + threadIdx.x;
    uint8_t e = 0;
    if (i < n) {
        e = prim_glv(k, k1, k2, n1, n2, ld, i);
        err[i] = e;
    }
    count_err(e, counter);
}
// 29-bit limb split: pure streaming, 32 B in + 72 B out per element.  Each lane converts two adjacent
// elements so that every global access is 16 B per lane (64 B contiguous loads, dwordx4 column stores).
__global__ __launch_bounds__(BS) void k_split(const uint8_t* packed, u64* limbs, size_t n, size_t ld) {
    size_t pairs = n >> 1;
    size_t stride = (size_t)gridDim.x * BS;
    bool aligned = ((ld & 1) == 0) && ((reinterpret_cast<uintptr_t>(limbs) & 15) == 0);
    for (size_t p = (size_t)blockIdx.x * BS + threadIdx.x; p < pairs; p += stride) {
        const uint4* src = reinterpret_cast<const uint4*>(packed + 64 * p);
        uint4 a0 = src[0], a1 = src[1], b0 = src[2], b1 = src[3];
        U256 va, vb;
        va.w[0] = a0.x; va.w[1] = a0.y; va.w[2] = a0.z; va.w[3] = a0.w;
        va.w[4] = a1.x; va.w[5] = a1.y; va.w[6] = a1.z; va.w[7] = a1.w;
        vb.w[0] = b0.x; vb.w[1] = b0.y; vb.w[2] = b0.z; vb.w[3] = b0.w;
        vb.w[4] = b1.x; vb.w[5] = b1.y; vb.w[6] = b1.z; vb.w[7] = b1.w;
        u32 la[NL], lb[NL];
        split29(va, la);
        split29(vb, lb);
        if (aligned) {
#pragma unroll
            for (int k = 0; k < NL; k++) {
                uint4 o = make_uint4(la[k], 0u, lb[k], 0u);
                *reinterpret_cast<uint4*>(limbs + (size_t)k * ld + 2 * p) = o;
            }
        } else {
#pragma unroll
            for (int k = 0; k < NL; k++) {
                limbs[(size_t)k * ld + 2 * p] = la[k];
                limbs[(size_t)k * ld + 2 * p + 1] = lb[k];
            }
        }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) prim_split(packed, limbs, ld, n - 1);
}
__global__ __launch_bounds__(BS) void k_pack(const u64* limbs, uint8_t* packed, size_t n, size_t ld, uint8_t* err,
                                             unsigned long long* counter) {
    size_t i = (size_t)blockIdx.x * BS + threadIdx.x;
    uint8_t e = 0;
    if (i < n) {
        e = prim_pack(limbs, packed, ld, i);
        err[i] = e;
    }
    count_err(e, counter);
}

// Column-major witness matrix -> one contiguous row per signature (what a per-signature PartialWitness
// fill wants to read): 64 x 64 u64 tiles through LDS, both global sides coalesced 512-byte runs.
// Rows of the tile are padded to 65 elements so the transposed read walks distinct banks.
template <class T>
__global__ __launch_bounds__(BS) void k_transpose(const T* __restrict__ cols, size_t ld, size_t n, size_t ncols,
                                                  T* __restrict__ rows, size_t row_ld) {
    __shared__ T tile[64][65];
    const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t s0 = (size_t)blockIdx.x * 64, c0 = (size_t)blockIdx.y * 64;
#pragma unroll 4
    for (unsigned r = w; r < 64; r += 4) {   // column c0 + r, signatures s0 .. s0 + 63
        size_t c = c0 + r, sg = s0 + lane;
        if (c < ncols && sg < n) tile[r][lane] = cols[c * ld + sg];
    }
    __syncthreads();
#pragma unroll 4
    for (unsigned r = w; r < 64; r += 4) {   // signature s0 + r, columns c0 .. c0 + 63
        size_t sg = s0 + r, c = c0 + lane;
        if (sg < n && c < ncols) rows[sg * row_ld + c] = tile[lane][r];
    }
}

// Wire-matrix assembly (p2e_assemble_wires): out[sig][dst(e)] = M_src(e)[col(e)][sig] for the entries e of a map
// sorted by dst.  A tile of 64 entries x 64 signatures goes through LDS like k_transpose: the gather side reads 64
// consecutive signatures of one column per row (512 B runs), the scatter side writes, per signature, 64 entries whose
// destinations are consecutive wherever the map is (runs of rows of one wire).
struct AssembleArgs {
    const u64* cols;
    size_t ld;
    const u64* aux;
    size_t ald;
    const void* ux;
    size_t uld;
    int ux_u32;
    const u64* gate;
    size_t gld;
    const u32* src;   // [count]
    const u32* dst;   // [count], ascending
    size_t count;
    u64* wires;
    size_t stride, n;
};
__global__ __launch_bounds__(BS) void k_assemble(AssembleArgs A) {
    __shared__ u64 tile[64][65];
    const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t s0 = (size_t)blockIdx.x * 64, e0 = (size_t)blockIdx.y * 64;
#pragma unroll 4
    for (unsigned r = w; r < 64; r += 4) {   // entry e0 + r, signatures s0 .. s0 + 63
        const size_t e = e0 + r, sg = s0 + lane;
        if (e < A.count && sg < A.n) {
            const u32 s = A.src[e], c = s & 0x3FFFFFFFu;
            u64 v;
            if ((s >> 30) == 0)
                v = A.cols[(size_t)c * A.ld + sg];
            else if ((s >> 30) == 1)
                v = A.aux[(size_t)c * A.ald + sg];
            else if ((s >> 30) == 2)
                v = A.ux_u32 ? (u64) static_cast<const u32*>(A.ux)[(size_t)c * A.uld + sg] : static_cast<const u64*>(A.ux)[(size_t)c * A.uld + sg];
            else
                v = A.gate[(size_t)c * A.gld + sg];
            tile[r][lane] = v;
        }
    }
    __syncthreads();
    const size_t e = e0 + lane;
    const u32 d = e < A.count ? A.dst[e] : 0u;
#pragma unroll 4
    for (unsigned r = w; r < 64; r += 4) {   // signature s0 + r, entries e0 .. e0 + 63
        const size_t sg = s0 + r;
        if (sg < A.n && e < A.count) A.wires[sg * A.stride + d] = tile[lane][r];
    }
}

// Compact container for transfers (p2e_columns_compact): every column whose values are < 2^32 by construction
// (29-bit limbs, overflow words, flags: 57 % of the columns) is repacked as u32, the check_sum / carry columns of
// the mul generators stay u64.  map[c] = index of column c in its matrix | P2E_COMPACT_WIDE.  One lane moves two
// adjacent signatures (16-byte loads, 8- or 16-byte stores); blockIdx.y walks groups of 32 columns.
constexpr u32 COMPACT_WIDE = 0x80000000u;
constexpr int COMPACT_GROUP = 32;
__global__ __launch_bounds__(BS) void k_compact(const u64* __restrict__ cols, size_t ld, size_t n, u32 ncols,
                                                const u32* __restrict__ map, u32* __restrict__ narrow, size_t ldn,
                                                u64* __restrict__ wide, size_t ldw, u32* err32, int pair_ok) {
    const size_t i = ((size_t)blockIdx.x * BS + threadIdx.x) * 2;
    if (i >= n) return;
    const bool two = pair_ok && i + 1 < n;
    const u32 c0 = blockIdx.y * COMPACT_GROUP;
    u32 bad_a = 0, bad_b = 0;   // high words seen in narrow columns, per signature of the lane
#pragma unroll 4
    for (u32 k = 0; k < COMPACT_GROUP; k++) {
        const u32 c = c0 + k;
        if (c >= ncols) break;
        const u32 m = map[c];
        const u64* src = cols + (size_t)c * ld + i;
        u64 a, b = 0;
        if (two) {
            const uint4 v = *reinterpret_cast<const uint4*>(src);
            a = (u64)v.x | ((u64)v.y << 32);
            b = (u64)v.z | ((u64)v.w << 32);
        } else {
            a = src[0];
            if (i + 1 < n) b = src[1];
        }
        if (m & COMPACT_WIDE) {
            u64* dst = wide + (size_t)(m & ~COMPACT_WIDE) * ldw + i;
            if (two) {
                *reinterpret_cast<uint4*>(dst) = make_uint4((u32)a, (u32)(a >> 32), (u32)b, (u32)(b >> 32));
            } else {
                dst[0] = a;
                if (i + 1 < n) dst[1] = b;
            }
        } else {
            bad_a |= (u32)(a >> 32);
            bad_b |= (u32)(b >> 32);
            u32* dst = narrow + (size_t)m * ldn + i;
            if (two) {
                *reinterpret_cast<uint2*>(dst) = make_uint2((u32)a, (u32)b);
            } else {
                dst[0] = (u32)a;
                if (i + 1 < n) dst[1] = (u32)b;
            }
        }
    }
    // a value that does not fit its 32-bit slot
    if (bad_a) atomicOr(&err32[i], (u32)ERR_LIMB_RANGE);
    if (bad_b) atomicOr(&err32[i + 1], (u32)ERR_LIMB_RANGE);
}
#endif   // P2E_HAS(0)

// ====================================================================================================
// context
// ====================================================================================================
#if P2E_HAS(0)
thread_local std::string g_last_error;
#else
extern thread_local std::string g_last_error;
#endif
static void set_error(const std::string& s) { g_last_error = s; }
#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                    \
            return P2E_E_HIP;                                                                \
        }                                                                                    \
    } while (0)

// Entry points run on the context's device and leave the caller's current device as they found it (a multi-GPU
// process, or torch, keeps its own notion of "current device").
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int device) {
        int cur = -1;
        if (hipGetDevice(&cur) == hipSuccess && cur != device) prev = cur;
        (void)hipSetDevice(device);
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

struct DeviceProgram {
    Program prog;
    OpDesc* d_ops = nullptr;        // with the run marks of ctx->run_iters (F_NO_AFFINE)
    OpDesc* d_ops_plain = nullptr;  // without: every op expanded on its own (small batches)
    OpDesc* d_ops_mid = nullptr;    // with the run marks of ctx->run_iters_mid (mid-size batches)
    OpDesc* d_ops_small = nullptr;  // with the run marks of ctx->run_iters_small (four-lane plan)
    std::vector<OpDesc> h_ops;
    std::vector<host::GenOp> gens;
    // built-in-generator columns (aux.hpp)
    std::vector<AuxItem> aux_items;
    std::vector<host::AuxGen> aux_gens;
    AuxTables aux_tab{};
    AuxItem* d_aux_items = nullptr;
    AuxTables* d_aux_tab = nullptr;
    // constraint-block columns (ux.hpp)
    std::vector<UxItem> ux_items;
    std::vector<u32> ux_first, ux_count;
    u32 num_ux_cols = 0;
    UxItem* d_ux_items = nullptr;
    std::vector<GateItem> gate_items;
    u32 num_gate_cols = 0;
    GateItem* d_gate_items = nullptr;
    // compact container (p2e_columns_compact): per-column slot, narrow / wide column counts
    std::vector<u32> compact_map;
    u32 num_narrow = 0, num_wide = 0;
    u32* d_compact_map = nullptr;
    std::vector<u32> wide_before;   // [num_cols + 1]: wide columns before column c (Sink::wide_before)
    u32* d_wide_before = nullptr;
};

struct p2e_ctx {
    int device = 0;
    unsigned flags = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // The Jacobian chains are latency-bound (one wave per SIMD, strictly sequential) and leave most of the
    // ALU and all of the HBM bandwidth idle; the witness expansion is HBM-bound.  So the chains run on
    // their own high-priority streams, cut into pieces, and phases B/C of every finished piece run on the
    // caller's stream underneath the following pieces.
    static constexpr int MAX_PIECES = 16;
    static constexpr int MAX_SEG = 2 * MAX_PIECES + 2;
    // st_binv, st_c2: second phase-B and second phase-C stream of the small-batch plan
    hipStream_t st_msm = nullptr, st_fixed = nullptr, st_binv = nullptr, st_c2 = nullptr;
    // st_c1: the library's OWN first expansion stream (stream_layout with a '1'): the caller's stream then only starts the
    // call (scalar phase) and ends it (finalisation); st_pad: spare streams that only exist to occupy hardware queues
    hipStream_t st_c1 = nullptr, st_pad[8] = {};
    // experiment (P2E_QUAD_EXPAND_CUS=k): expansion streams of the four-lane plan restricted to k compute units, so that
    // the latency-critical chain waves find SIMDs the HBM-bound expansions do not occupy
    hipStream_t st_c1q = nullptr, st_c2q = nullptr;
    hipEvent_t ev_c1join = nullptr;
    hipEvent_t ev_c2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_fixed = nullptr, ev_piece[MAX_SEG] = {}, ev_binv[MAX_SEG] = {};
    // one event pair around every expansion launch; kind 0 = k_expand (op by op), 1 = k_expand_runs, 2 = k_expand_fb_run
    static constexpr int MAX_EXPAND = 2 * MAX_SEG;
    hipEvent_t ev_c0[MAX_EXPAND] = {}, ev_c1[MAX_EXPAND] = {};
    int n_expand = 0, n_seg = 0;
    double expand_cols[MAX_EXPAND] = {};
    int expand_kind[MAX_EXPAND] = {};
    // column blocks of the last fused call in issue order (p2e_segments_describe): ev < 0 = the scalar phase (ev[1]),
    // otherwise the block is final when ev_c1[ev] has completed
    struct SegBlock {
        u32 col0, ncols;
        int ev;
    };
    std::vector<SegBlock> seg_blocks;
    int msm_pieces = 8, fixed_pieces = 2;   // one Montgomery inversion batch per piece
    int msm_pieces_small = 5, fixed_pieces_small = 1;   // ... of the small-batch plan (fewer launches and inversions)
    int run_iters = 9;                      // MSM-loop iterations per expansion run (0: expand op by op)
    // with runs: the fixed-base windows as one run per signature (k_expand_fb_run).  OFF by default in the built-in
    // verifier: measured 11.43 / 11.05 ms against 10.86 / 11.28 ms per 2^16 batch (alternating processes on one box) --
    // the single run has one wave per SIMD and lands in the window where the chains and inversions contend with it
    // (0.40-0.42 of peak against 0.43-0.47 for the same columns through k_expand); P2E_FB_RUN=1 turns it on.  The
    // P-256 verifier program uses it (curve_api.inc), where it sits beside a longer windowed chain.
    bool fb_run = false;
    // A run is walked by ONE lane, so a launch of r runs has only r * n/64 waves: below this batch size the
    // 1024 SIMDs are better filled by one workgroup row per op (2^10 glv_mul fills: 3.0 ms against 9.5 ms)
    size_t runs_min_n = 21505;              // (= every batch the four-lane plan does not take, see quad_max_n)
    size_t cp_runs_min_n = 49152;           // the same threshold for the curve programs (curve_api.inc), measured there only at 2^13 / 2^16
    // Below this batch size phases A and B are latency, not throughput: four lanes per signature walk the chains
    // (k_chains_quad) and every inversion batch is cut into 2^binv_split_log2 sub-ranges (k_batch_inv_split)
    // (24 576 until phase B of the lane-per-signature plan went onto two streams below binv_alt_max_n: since then that plan
    // wins from 2^14 up -- 3.94 against 4.60 ms at 16 384, 4.96 against 6.46 ms at 24 576, 3.81 against 3.62 ms at 12 288)
    // (round 3, with short expansion runs and front-loaded pieces: 3.56-3.76 against 4.10-4.15 ms at 16 384, 4.73 against
    // 4.5 ms at 20 480, profiles/r03_plan_threshold_resweep.txt -- the threshold moved up from 14 336 and now takes 2^14)
    // (with the chains on lazy limbs and the safegcd inversion: 3.44-3.52 against 3.96-4.03 ms at 16 384, 4.05-4.28 against
    // 4.25-4.35 at 20 480, 5.01-5.04 against 4.72-4.73 at 24 576, profiles/r03_plan_threshold_lazy_limbs.txt: 17 408 -> 21 504)
    size_t quad_max_n = 21504;
    size_t cp_quad_max_n = 17408;   // the curve programs' own threshold (P-256 keeps canonical words: measured with those)
    // Between the two plans (lane per signature, but fewer than one chain wave per SIMD) phase B is the serial resource:
    // its kernels are latency-bound (half a wave per SIMD at 2^15) and queue on one stream from the first piece to the
    // last, with every expansion waiting behind them.  Below this batch size the inversion batches of consecutive pieces
    // alternate between two streams -- and are cut into 2^binv_mid_split_log2 sub-ranges each -- so that they overlap.
    size_t binv_alt_max_n = 49152;
    // ... and the loop is cut into fewer pieces of longer runs there (5 pieces of 12-iteration runs instead of 8 of 9:
    // 5.96 against 6.45 ms at 2^15, 7.65 against 7.95 ms at 40 960, 9.15 against 9.53 ms at 48 896)
    int msm_pieces_mid = 5, run_iters_mid = 12;
    // four-lane plan: the loop expanded as SHORT runs (4 iterations = 12 ops, two of them keep their affine form): phase B
    // then does 3 instead of 8 multiplications for the other ten, and a piece of 5 runs still launches 5 * n/64 waves.
    // 2.35 / 2.27 against 2.42 ms at 2^13, 2.88 / 2.96 against 3.05 ms at 12 288 (profiles/r03_quad_plan_run_expansion_sweep.txt;
    // R = 2, 3, 6 and 7 pieces are slower).  0: every op expanded on its own, as in round 2.
    int run_iters_small = 4;
    int binv_mid_split_log2 = 1;   // 2^15 per call: 7.03-7.08 ms on one stream, 6.76-6.83 alternating, 6.68-6.72 alternating and split in two
    int binv_split_log2 = 2;
    // small-batch plan: dynamic LDS bytes requested by the expansion kernels (they do not use it): caps how many of
    // their workgroups share a CU, so that the register file keeps room for the chain waves queued behind them
    // small-batch plan: runs per loop piece (front-loaded: the LAST piece's inversion batch and expansion are the
    // exposed tail of the call, so it is the shortest), 0-terminated; empty = equal pieces.  And the split of the last
    // piece's inversion batch (latency matters there; the earlier ones only need throughput: fewer inversions).
    int small_takes[p2e_ctx::MAX_PIECES + 1] = {0};
    int binv_split_log2_last = 3;
    // the fixed-base chain's batch: 67 ops that all keep their affine form, the longest walk, and its expansion is the largest
    // single launch of the call -- eight sub-ranges: 1.89 against 1.95 ms at 2^13, level at 2^14 (profiles/r03_fixed_base_batch_split.txt)
    int binv_split_log2_fixed = 3;
    // small-batch plan: which of the two phase-B streams takes the FIRST batch after the window table's (the fixed-base
    // chain's).  1: the fixed-base chain's own stream -- the table's batch occupies the other one until ~0.7 ms, and the
    // fixed-base batch (67 ops that all keep their affine form: the longest) queued behind it used to hold up the second
    // loop piece's batch in turn.  0: the round-2 order.
    int quad_b_first_on_fixed = 1;
    bool quad_few_waits = true;   // P2E_QUAD_FEW_WAITS=0: one wait per earlier piece, as before
    unsigned expand_lds_small = 54000;   // (160 000 -- one expansion workgroup per CU -- while the chains were the bottleneck; with lazy-limb chains 54 000 is 3-4 % faster at 2^13, profiles/r03_quad_plan_lazy_limbs_sweeps.txt)
    unsigned expand_lds = 0;   // the same knob for the large-batch plan
    Aff* d_cpts = nullptr;
    Aff* d_fbtab = nullptr;
    U256* d_constv = nullptr;   // circuit constants by id (AUX_SRC_CONST | id), for the constraint-block pass
    DeviceProgram progs[2];
    void* scratch = nullptr;
    size_t scratch_bytes = 0;
    unsigned long long* d_counter = nullptr;
    unsigned long long* h_counter = nullptr;  // pinned
    hipEvent_t ev[6] = {};   // [0],[1] around k_scalar, [5] end of the call (2..4 unused)
    float phase_ms[12] = {0};   // [0] scalar, [4] whole call, per expansion kind k (launches, columns, summed ms): [1..3] k_expand,
                                // [5..7] k_expand_runs, [8..10] k_expand_fb_run
    bool have_phases = false;
};

#if P2E_HAS(0)
// wide = the 33 check_sum / carry columns of every mul generator (Goldilocks residues of signed sums, carries offset
// by 2^33: gates/mul_nonnative.rs:305-322,518-527); everything else on the path is < 2^32 by construction
static void build_compact_map(DeviceProgram& P) {
    P.compact_map.assign((size_t)P.prog.num_cols, 0);
    for (const auto& g : P.gens)
        for (u32 k = 0; k < g.ncols; k++) {
            const bool wide = g.kind == host::GEN_MUL && k >= 2 * NL;
            P.compact_map[g.col + k] = wide ? (COMPACT_WIDE | P.num_wide++) : P.num_narrow++;
        }
    P.wide_before.assign((size_t)P.prog.num_cols + 1, 0);
    for (size_t col = 0; col < P.compact_map.size(); col++)
        P.wide_before[col + 1] = P.wide_before[col] + ((P.compact_map[col] & COMPACT_WIDE) ? 1u : 0u);
}
static const DeviceProgram& host_program(int program) {
    static DeviceProgram P[2];
    static std::once_flag once;
    std::call_once(once, [] {
        host::ScheduleBuilder b0;
        b0.verify_secp256k1_message_circuit();
        P[0].prog = b0.prog;
        P[0].gens = b0.gens;
        P[0].aux_items = b0.aux_items;
        P[0].aux_gens = b0.aux_gens;
        P[0].aux_tab = b0.aux_tab;
        P[0].ux_items = b0.ux_items;
        P[0].ux_first = b0.ux_first;
        P[0].ux_count = b0.ux_count;
        P[0].num_ux_cols = b0.num_ux_cols;
        P[0].gate_items = b0.gate_items;
        P[0].num_gate_cols = b0.num_gate_cols;
        build_compact_map(P[0]);
        host::ScheduleBuilder b1;
        b1.glv_mul_circuit();
        P[1].prog = b1.prog;
        P[1].gens = b1.gens;
        P[1].aux_items = b1.aux_items;
        P[1].aux_gens = b1.aux_gens;
        P[1].aux_tab = b1.aux_tab;
        P[1].ux_items = b1.ux_items;
        P[1].ux_first = b1.ux_first;
        P[1].ux_count = b1.ux_count;
        P[1].num_ux_cols = b1.num_ux_cols;
        P[1].gate_items = b1.gate_items;
        P[1].num_gate_cols = b1.num_gate_cols;
        build_compact_map(P[1]);
    });
    return P[program];
}
static std::vector<OpDesc> host_ops(int program, int run_iters, bool fb_run = true) {
    host::ScheduleBuilder b;
    if (program == 0)
        b.verify_secp256k1_message_circuit();
    else
        b.glv_mul_circuit();
    b.mark_runs(run_iters, fb_run);
    return b.ops;
}
#endif   // P2E_HAS(0)

struct ScratchLayout {
    size_t px, py, pz, pw, pref, ax, ay, dig4, dig2, msrc, dyn, src, err32, valid8, total;
};
static ScratchLayout scratch_layout(const Program& G, size_t n) {
    ScratchLayout L{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    };
    L.px = take((size_t)G.num_slots * n * 32);
    L.py = take((size_t)G.num_slots * n * 32);
    L.pz = take((size_t)G.num_slots * n * 32);
    L.pw = take((size_t)G.num_ops * n * 32);
    L.pref = take((size_t)G.num_ops * n * 32);
    L.ax = take((size_t)G.num_slots * n * 32);
    L.ay = take((size_t)G.num_slots * n * 32);
    L.dig4 = take((size_t)FB_WINDOWS * n);
    const size_t rows = (size_t)(G.cp_rows > MSM_DIGITS ? G.cp_rows : MSM_DIGITS);   // curve programs: up to 261 bit rows
    L.dig2 = take(rows * n);
    L.msrc = take(rows * n * 2);
    L.dyn = take((size_t)G.num_cadd * n * 2);
    L.src = take((size_t)G.num_ops * 2 * n * 2);
    L.err32 = take(n * 4);
    L.valid8 = take(n);
    L.total = off;
    return L;
}

#if P2E_HAS(0)
extern "C" const char* p2e_last_error(void) { return g_last_error.c_str(); }

extern "C" size_t p2e_scratch_bytes(int program, size_t n) {
    if (program < 0 || program > 1) return 0;
    return scratch_layout(host_program(program).prog, n).total;
}

extern "C" int p2e_ctx_create(int device, unsigned flags, void* stream, p2e_ctx** out) {
    if (!out) return P2E_E_INVALID;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
        set_error("no HIP device visible: libp2e_hip needs a gfx950 GPU (there is no CPU fallback)");
        return P2E_E_NO_DEVICE;
    }
    if (device < 0 || device >= count) {
        set_error("device index out of range");
        return P2E_E_INVALID;
    }
    DeviceGuard guard(device);
    p2e_ctx* c = new p2e_ctx();
    c->device = device;
    c->flags = flags;
    // a failing HIP call below must not leak the context and what it already owns
    struct Cleanup {
        p2e_ctx* c;
        ~Cleanup() {
            if (c) p2e_ctx_destroy(c);
        }
    } cleanup{c};
    if (stream) {
        c->stream = (hipStream_t)stream;
    } else {
        // A BLOCKING stream: it keeps the implicit ordering with the legacy default stream, which is what a caller
        // that passes "its current stream" means when that stream is the default one (handle 0, indistinguishable
        // from NULL): e.g. torch.zeros(...) on torch's default stream followed by a call that writes the tensor.
        HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamDefault));
        c->own_stream = true;
    }
    const host::Consts& C = host::consts();
    HIP_TRY(hipMalloc(&c->d_cpts, sizeof(Aff) * NUM_CONST_PTS));
    HIP_TRY(hipMalloc(&c->d_fbtab, sizeof(Aff) * C.fbtab.size()));
    HIP_TRY(hipMemcpy(c->d_cpts, C.cpts, sizeof(Aff) * NUM_CONST_PTS, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_fbtab, C.fbtab.data(), sizeof(Aff) * C.fbtab.size(), hipMemcpyHostToDevice));
    {
        U256 cv[NUM_CONSTV];
        for (u32 id = 0; id < NUM_CONSTV; id++) cv[id] = host::ScheduleBuilder::const_value(id);
        HIP_TRY(hipMalloc(&c->d_constv, sizeof cv));
        HIP_TRY(hipMemcpy(c->d_constv, cv, sizeof cv, hipMemcpyHostToDevice));
    }
    if (const char* env = getenv("P2E_RUN_ITERS")) {
        int v = atoi(env);
        if (v >= 0 && v <= MSM_DIGITS) c->run_iters = v;
    }
    if (const char* env = getenv("P2E_FB_RUN")) c->fb_run = atoi(env) != 0;
    if (const char* env = getenv("P2E_RUNS_MIN_N")) c->runs_min_n = (size_t)strtoull(env, nullptr, 10);
    if (const char* env = getenv("P2E_QUAD_MAX_N")) c->quad_max_n = c->cp_quad_max_n = (size_t)strtoull(env, nullptr, 10);
    if (const char* env = getenv("P2E_BINV_ALT_MAX_N")) c->binv_alt_max_n = (size_t)strtoull(env, nullptr, 10);
    if (const char* env = getenv("P2E_CP_RUNS_MIN_N")) c->cp_runs_min_n = (size_t)strtoull(env, nullptr, 10);
    if (const char* env = getenv("P2E_MSM_PIECES_MID")) {
        int v = atoi(env);
        if (v >= 1 && v <= p2e_ctx::MAX_PIECES) c->msm_pieces_mid = v;
    }
    if (const char* env = getenv("P2E_RUN_ITERS_SMALL")) {
        int v = atoi(env);
        if (v >= 0 && v <= MSM_DIGITS) c->run_iters_small = v;
    }
    if (const char* env = getenv("P2E_RUN_ITERS_MID")) {
        int v = atoi(env);
        if (v >= 0 && v <= MSM_DIGITS) c->run_iters_mid = v;
    }
    if (const char* env = getenv("P2E_BINV_MID_SPLIT_LOG2")) {
        int v = atoi(env);
        if (v >= 0 && v <= 3) c->binv_mid_split_log2 = v;
    }
    if (const char* env = getenv("P2E_EXPAND_LDS_SMALL")) c->expand_lds_small = (unsigned)strtoul(env, nullptr, 10);
    if (const char* env = getenv("P2E_EXPAND_LDS")) c->expand_lds = (unsigned)strtoul(env, nullptr, 10);
    {   // never ask for more dynamic LDS than a workgroup may have on this device (160 KB on gfx950)
        int max_lds = 0;
        if (hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device) != hipSuccess || max_lds <= 0) max_lds = 65536;
        (void)hipGetLastError();
        if (c->expand_lds_small > (unsigned)max_lds) c->expand_lds_small = (unsigned)max_lds;
        if (c->expand_lds > (unsigned)max_lds) c->expand_lds = (unsigned)max_lds;
    }
    if (const char* env = getenv("P2E_QUAD_B_FIRST_ON_FIXED")) c->quad_b_first_on_fixed = atoi(env) != 0;
    if (const char* env = getenv("P2E_QUAD_FEW_WAITS")) c->quad_few_waits = atoi(env) != 0;
    if (const char* env = getenv("P2E_BINV_SPLIT_LOG2_FIXED")) {
        int v = atoi(env);
        if (v >= 0 && v <= 4) c->binv_split_log2_fixed = v;
    }
    if (const char* env = getenv("P2E_BINV_SPLIT_LOG2_LAST")) {
        int v = atoi(env);
        if (v >= 0 && v <= 4) c->binv_split_log2_last = v;
    }
    if (const char* env = getenv("P2E_SMALL_TAKES")) {   // e.g. "24,20,16,9,4": loop iterations per piece (op-by-op expansion: any cut)
        int k = 0;
        for (const char* p = env; *p && k < p2e_ctx::MAX_PIECES; k++) {
            c->small_takes[k] = atoi(p);
            while (*p && *p != ',') p++;
            if (*p == ',') p++;
        }
        c->small_takes[k] = 0;
    }
    if (const char* env = getenv("P2E_BINV_SPLIT_LOG2")) {
        int v = atoi(env);
        if (v >= 0 && v <= 4) c->binv_split_log2 = v;
    }
    for (int p = 0; p < 2; p++) {
        c->progs[p].prog = host_program(p).prog;
        std::vector<OpDesc> ops = host_ops(p, c->run_iters, c->fb_run);
        c->progs[p].h_ops = ops;
        HIP_TRY(hipMalloc(&c->progs[p].d_ops, sizeof(OpDesc) * ops.size()));
        HIP_TRY(hipMemcpy(c->progs[p].d_ops, ops.data(), sizeof(OpDesc) * ops.size(), hipMemcpyHostToDevice));
        const DeviceProgram& HP = host_program(p);
        c->progs[p].aux_tab = HP.aux_tab;
        c->progs[p].aux_items = HP.aux_items;
        HIP_TRY(hipMalloc(&c->progs[p].d_aux_items, sizeof(AuxItem) * HP.aux_items.size()));
        HIP_TRY(hipMemcpy(c->progs[p].d_aux_items, HP.aux_items.data(), sizeof(AuxItem) * HP.aux_items.size(), hipMemcpyHostToDevice));
        c->progs[p].ux_items = HP.ux_items;
        c->progs[p].num_ux_cols = HP.num_ux_cols;
        HIP_TRY(hipMalloc(&c->progs[p].d_ux_items, sizeof(UxItem) * HP.ux_items.size()));
        HIP_TRY(hipMemcpy(c->progs[p].d_ux_items, HP.ux_items.data(), sizeof(UxItem) * HP.ux_items.size(), hipMemcpyHostToDevice));
        c->progs[p].gate_items = HP.gate_items;
        c->progs[p].num_gate_cols = HP.num_gate_cols;
        HIP_TRY(hipMalloc(&c->progs[p].d_gate_items, sizeof(GateItem) * HP.gate_items.size()));
        HIP_TRY(hipMemcpy(c->progs[p].d_gate_items, HP.gate_items.data(), sizeof(GateItem) * HP.gate_items.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc(&c->progs[p].d_aux_tab, sizeof(AuxTables)));
        HIP_TRY(hipMemcpy(c->progs[p].d_aux_tab, &HP.aux_tab, sizeof(AuxTables), hipMemcpyHostToDevice));
        c->progs[p].num_narrow = HP.num_narrow;
        c->progs[p].num_wide = HP.num_wide;
        HIP_TRY(hipMalloc(&c->progs[p].d_compact_map, sizeof(u32) * HP.compact_map.size()));
        HIP_TRY(hipMemcpy(c->progs[p].d_compact_map, HP.compact_map.data(), sizeof(u32) * HP.compact_map.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc(&c->progs[p].d_wide_before, sizeof(u32) * HP.wide_before.size()));
        HIP_TRY(hipMemcpy(c->progs[p].d_wide_before, HP.wide_before.data(), sizeof(u32) * HP.wide_before.size(), hipMemcpyHostToDevice));
        std::vector<OpDesc> mid = host_ops(p, c->run_iters_mid, c->fb_run);
        HIP_TRY(hipMalloc(&c->progs[p].d_ops_mid, sizeof(OpDesc) * mid.size()));
        HIP_TRY(hipMemcpy(c->progs[p].d_ops_mid, mid.data(), sizeof(OpDesc) * mid.size(), hipMemcpyHostToDevice));
        std::vector<OpDesc> small = host_ops(p, c->run_iters_small, c->fb_run);
        HIP_TRY(hipMalloc(&c->progs[p].d_ops_small, sizeof(OpDesc) * small.size()));
        HIP_TRY(hipMemcpy(c->progs[p].d_ops_small, small.data(), sizeof(OpDesc) * small.size(), hipMemcpyHostToDevice));
        std::vector<OpDesc> plain = host_ops(p, 0);
        HIP_TRY(hipMalloc(&c->progs[p].d_ops_plain, sizeof(OpDesc) * plain.size()));
        HIP_TRY(hipMemcpy(c->progs[p].d_ops_plain, plain.data(), sizeof(OpDesc) * plain.size(), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMalloc(&c->d_counter, sizeof(unsigned long long)));
    HIP_TRY(hipHostMalloc(&c->h_counter, sizeof(unsigned long long)));
    // ev[1] closes the scalar phase's column block (p2e_segments_*): it exists either way, with a timestamp only on request
    for (auto& e : c->ev) HIP_TRY(hipEventCreateWithFlags(&e, (c->flags & P2E_CTX_PHASE_TIMING) ? hipEventDefault : hipEventDisableTiming));
    int prio_lo = 0, prio_hi = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    // Internal streams, created AND bound to their hardware queues (one empty launch each) in the order of the layout
    // string: M / F / B = the two chain streams and the second inversion stream (high priority), 2 = the second expansion
    // stream of the small-batch plan (low priority), 1 = an own first expansion stream (normal priority; without it the
    // caller's stream carries the expansions), P = a spare stream.  HIP binds a hardware queue to a stream at its first
    // launch and hands queues out in that order, and the step time of the small and mid-size plans depends on WHICH queues
    // the expansion, chain and inversion streams get relative to each other (profiles/r03_stream_order_*: the +-20 %
    // "creation order" effect of round 2 goes with the caller's queue sitting four positions from st_c2 in one order and
    // from st_binv in the other -- consistent with queues sharing a hardware pipe delaying each other's dispatch, though
    // the pipe assignment itself is not visible from here).  With every stream the pipeline dispatches on created here,
    // back to back, their relative placement no longer depends on what the caller did first.
    {
        const char* layout = getenv("P2E_STREAM_LAYOUT");
        // default: chain stream, OWN expansion stream, second chain stream, second inversion stream, second expansion
        // stream.  Of twelve layouts swept on two boxes (profiles/r03_stream_layout_sweep_box*.txt, one process per
        // setting) this is one of two whose step time is the same within 3 % whether the caller touched the GPU before
        // or after creating the context, at 2^13 ... 2^16 per call: 2.45 / 4.0-4.1 / 6.0-6.1 / 10.8 ms, against 2.29 or
        // 2.77 / 4.15 or 3.95 / 6.0 / 10.8 ms for the round-2 placement "MFB2" with the expansions on the caller's stream.
        if (!layout || !*layout) layout = "M1FB2";
        const char* tenv = getenv("P2E_TOUCH_STREAMS");
        const bool touch = !tenv || atoi(tenv) != 0;
        if (touch) {   // the legacy default stream first: a caller that has not touched the GPU yet gets its queue now
            hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, (hipStream_t)0);
            HIP_TRY(hipStreamSynchronize((hipStream_t)0));
            hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, c->stream);
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
        int pads = 0;
        for (const char* p = layout; *p; p++) {
            hipStream_t* slot = nullptr;
            int prio = prio_hi;
            switch (*p) {
            case 'M': slot = &c->st_msm; break;
            case 'F': slot = &c->st_fixed; break;
            case 'B': slot = &c->st_binv; break;
            case '2': slot = &c->st_c2; prio = prio_lo; break;
            case '1': slot = &c->st_c1; prio = 0; break;
            case 'P': if (pads < 8) slot = &c->st_pad[pads++]; prio = 0; break;
            default: break;
            }
            if (!slot || *slot) continue;
            HIP_TRY(hipStreamCreateWithPriority(slot, hipStreamNonBlocking, prio));
            if (touch) {
                hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, *slot);
                HIP_TRY(hipStreamSynchronize(*slot));
            }
        }
        if (!c->st_msm) HIP_TRY(hipStreamCreateWithPriority(&c->st_msm, hipStreamNonBlocking, prio_hi));
        if (!c->st_fixed) HIP_TRY(hipStreamCreateWithPriority(&c->st_fixed, hipStreamNonBlocking, prio_hi));
        if (!c->st_binv) HIP_TRY(hipStreamCreateWithPriority(&c->st_binv, hipStreamNonBlocking, prio_hi));
        if (!c->st_c2) HIP_TRY(hipStreamCreateWithPriority(&c->st_c2, hipStreamNonBlocking, prio_lo));
    }
    HIP_TRY(hipEventCreateWithFlags(&c->ev_c1join, hipEventDisableTiming));
    if (const char* env = getenv("P2E_QUAD_EXPAND_CUS")) {
        const int k = atoi(env);
        int ncu = 0;
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device);
        if (k > 0 && k < ncu) {
            const char* pat = getenv("P2E_QUAD_EXPAND_CU_PATTERN");   // "low" (default): CUs 0..k-1; "stride": every (ncu/k)-th
            std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
            for (int i = 0; i < k; i++) {
                const int cu = (pat && pat[0] == 's') ? (int)((long long)i * ncu / k) : i;
                mask[(size_t)cu >> 5] |= 1u << (cu & 31);
            }
            HIP_TRY(hipExtStreamCreateWithCUMask(&c->st_c1q, (uint32_t)mask.size(), mask.data()));
            HIP_TRY(hipExtStreamCreateWithCUMask(&c->st_c2q, (uint32_t)mask.size(), mask.data()));
        }
    }
    HIP_TRY(hipEventCreateWithFlags(&c->ev_c2, hipEventDisableTiming));
    for (auto& e : c->ev_binv) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_fixed, hipEventDisableTiming));
    for (auto& e : c->ev_piece) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : c->ev_c0) HIP_TRY(hipEventCreate(&e));
    for (auto& e : c->ev_c1) HIP_TRY(hipEventCreateWithFlags(&e, (c->flags & P2E_CTX_PHASE_TIMING) ? hipEventDefault : hipEventDisableTiming));
    if (const char* env = getenv("P2E_MSM_PIECES")) {
        int v = atoi(env);
        if (v >= 1 && v <= p2e_ctx::MAX_PIECES) c->msm_pieces = v;
    }
    if (const char* env = getenv("P2E_FIXED_PIECES")) {
        int v = atoi(env);
        if (v >= 1 && v <= p2e_ctx::MAX_PIECES) c->fixed_pieces = v;
    }
    if (const char* env = getenv("P2E_MSM_PIECES_SMALL")) {
        int v = atoi(env);
        if (v >= 1 && v <= p2e_ctx::MAX_PIECES) c->msm_pieces_small = v;
    }
    if (const char* env = getenv("P2E_FIXED_PIECES_SMALL")) {
        int v = atoi(env);
        if (v >= 1 && v <= p2e_ctx::MAX_PIECES) c->fixed_pieces_small = v;
    }
    cleanup.c = nullptr;
    *out = c;
    return 0;
}

extern "C" void p2e_ctx_destroy(p2e_ctx* c) {
    if (!c) return;
    DeviceGuard guard(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    (void)hipFree(c->d_cpts);
    (void)hipFree(c->d_fbtab);
    (void)hipFree(c->d_constv);
    for (auto& p : c->progs) {
        (void)hipFree(p.d_ux_items);
        (void)hipFree(p.d_gate_items);
        (void)hipFree(p.d_ops);
        (void)hipFree(p.d_ops_plain);
        (void)hipFree(p.d_ops_mid);
        (void)hipFree(p.d_ops_small);
        (void)hipFree(p.d_aux_items);
        (void)hipFree(p.d_aux_tab);
        (void)hipFree(p.d_compact_map);
        (void)hipFree(p.d_wide_before);
    }
    (void)hipFree(c->scratch);
    (void)hipFree(c->d_counter);
    (void)hipHostFree(c->h_counter);
    for (auto& e : c->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : c->ev_binv)
        if (e) (void)hipEventDestroy(e);
    for (hipStream_t st : {c->st_msm, c->st_fixed, c->st_binv, c->st_c2, c->st_c1, c->st_c1q, c->st_c2q, c->st_pad[0], c->st_pad[1], c->st_pad[2], c->st_pad[3],
                           c->st_pad[4], c->st_pad[5], c->st_pad[6], c->st_pad[7]})
        if (st) {
            (void)hipStreamSynchronize(st);
            (void)hipStreamDestroy(st);
        }
    for (hipEvent_t e : {c->ev_fork, c->ev_fixed, c->ev_c2, c->ev_c1join})
        if (e) (void)hipEventDestroy(e);
    for (auto& e : c->ev_piece)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : c->ev_c0)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : c->ev_c1)
        if (e) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// Which part of the fused pipeline an asynchronous failure (a kernel fault surfaces only when a stream is
// synchronised) had reached: the first recorded event of the call that did not complete.
static std::string pipeline_progress(p2e_ctx* c) {
    if (!c->have_phases) return "single-kernel call";
    if (hipEventQuery(c->ev[1]) != hipSuccess) return "scalar phase (k_scalar)";
    for (int k = 0; k < c->n_seg; k++) {
        if (hipEventQuery(c->ev_piece[k]) != hipSuccess) return "phase A (k_chains), chain piece " + std::to_string(k);
        if (hipEventQuery(c->ev_binv[k]) != hipSuccess) return "phase B (k_batch_inv), chain piece " + std::to_string(k);
    }
    for (int k = 0; k < c->n_expand; k++)
        if (hipEventQuery(c->ev_c1[k]) != hipSuccess)
            return std::string("phase C (") + (c->expand_kind[k] == 1 ? "k_expand_runs" : c->expand_kind[k] == 2 ? "k_expand_fb_run" : "k_expand") +
                   "), launch " + std::to_string(k);
    return "finalisation";
}

extern "C" int p2e_sync(p2e_ctx* c) {
    if (!c) return P2E_E_INVALID;
    DeviceGuard guard(c->device);
    hipError_t se = hipStreamSynchronize(c->stream);
    if (se != hipSuccess) {
        const std::string where = pipeline_progress(c);
        set_error(std::string("hipStreamSynchronize: ") + hipGetErrorString(se) + " [asynchronous failure; reached: " + where + "]");
        (void)hipGetLastError();
        return P2E_E_HIP;
    }
    if (c->have_phases) {   // launches and columns per kernel kind always; durations for P2E_CTX_PHASE_TIMING contexts
        const bool timed = (c->flags & P2E_CTX_PHASE_TIMING) != 0;
        c->phase_ms[0] = c->phase_ms[4] = 0.f;
        if (timed) {
            HIP_TRY(hipEventElapsedTime(&c->phase_ms[0], c->ev[0], c->ev[1]));   // scalar kernel
            HIP_TRY(hipEventElapsedTime(&c->phase_ms[4], c->ev[0], c->ev[5]));   // whole call
        }
        double cnt[3] = {0, 0, 0}, cols[3] = {0, 0, 0}, sum_ms[3] = {0, 0, 0};
        for (int k = 0; k < c->n_expand; k++) {
            float ms = 0.f;
            if (timed) HIP_TRY(hipEventElapsedTime(&ms, c->ev_c0[k], c->ev_c1[k]));
            const int kind = c->expand_kind[k];
            cnt[kind] += 1;
            cols[kind] += c->expand_cols[k];
            sum_ms[kind] += ms;
        }
        for (int kind = 0; kind < 3; kind++) {       // [1..3] k_expand, [5..7] k_expand_runs, [8..10] k_expand_fb_run
            const int o = kind == 0 ? 1 : kind == 1 ? 5 : 8;
            c->phase_ms[o] = (float)cnt[kind];       // launches
            c->phase_ms[o + 1] = (float)cols[kind];  // columns they wrote (per signature)
            c->phase_ms[o + 2] = (float)sum_ms[kind];  // their summed durations
        }
    }
    return (int)*c->h_counter;
}

extern "C" int p2e_last_phase_ms(p2e_ctx* c, float* out, int cap) {
    if (!c || !out) return P2E_E_INVALID;
    int k = cap < 12 ? cap : 12;
    for (int i = 0; i < k; i++) out[i] = c->phase_ms[i];
    return k;
}

extern "C" long p2e_segments_describe(p2e_ctx* c, p2e_segment_desc* out, size_t cap) {
    if (!c) return P2E_E_INVALID;
    for (size_t k = 0; out && k < c->seg_blocks.size() && k < cap; k++) {
        out[k].first_col = c->seg_blocks[k].col0;
        out[k].num_cols = c->seg_blocks[k].ncols;
    }
    return (long)c->seg_blocks.size();
}
static hipEvent_t segment_event(p2e_ctx* c, int k) {
    if (!c || k < 0 || (size_t)k >= c->seg_blocks.size()) return nullptr;
    const int e = c->seg_blocks[(size_t)k].ev;
    return e < 0 ? c->ev[1] : c->ev_c1[e];
}
extern "C" int p2e_segment_stream_wait(p2e_ctx* c, int k, void* stream) {
    hipEvent_t ev = segment_event(c, k);
    if (!ev) {
        set_error("no such segment (p2e_segments_describe of the last fused call)");
        return P2E_E_INVALID;
    }
    DeviceGuard guard(c->device);
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, ev, 0));
    return 0;
}
extern "C" int p2e_segment_sync(p2e_ctx* c, int k) {
    hipEvent_t ev = segment_event(c, k);
    if (!ev) {
        set_error("no such segment (p2e_segments_describe of the last fused call)");
        return P2E_E_INVALID;
    }
    DeviceGuard guard(c->device);
    HIP_TRY(hipEventSynchronize(ev));
    return 0;
}
#endif   // P2E_HAS(0)

// ---- column blocks of a fused call (p2e_segments_describe) -------------------------------------------
static inline u32 op_cols(const OpDesc& op) { return op.kind == OP_DBL ? COLS_DBL : op.kind == OP_CADD ? COLS_CADD : COLS_ADD; }
// the scalar phase writes every column that no curve op owns: the complement of the ops' column ranges
static void seg_begin_call(p2e_ctx* c, const std::vector<OpDesc>& h_ops, u32 num_cols) {
    c->seg_blocks.clear();
    std::vector<std::pair<u32, u32>> r;
    r.reserve(h_ops.size());
    for (const OpDesc& op : h_ops) r.push_back({op.col, op.col + op_cols(op)});
    std::sort(r.begin(), r.end());
    u32 at = 0;
    for (const auto& x : r) {
        if (x.first > at) c->seg_blocks.push_back({at, x.first - at, -1});
        at = std::max(at, x.second);
    }
    if (at < num_cols) c->seg_blocks.push_back({at, num_cols - at, -1});
}
// expansion launch e (event pair ev_c0[e] / ev_c1[e]) completes the columns of ops [lo, hi) (consecutive ops of one chain
// are consecutive generators: one contiguous block)
static void seg_note_expand(p2e_ctx* c, int e, const std::vector<OpDesc>& h_ops, int lo, int hi) {
    u32 c0 = 0xFFFFFFFFu, c1 = 0;
    for (int t = lo; t < hi; t++) {
        c0 = std::min(c0, h_ops[t].col);
        c1 = std::max(c1, h_ops[t].col + op_cols(h_ops[t]));
    }
    if (hi > lo) c->seg_blocks.push_back({c0, c1 - c0, e});
}

static int ensure_scratch(p2e_ctx* c, size_t bytes) {
    if (bytes <= c->scratch_bytes) return 0;
    if (c->scratch) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipFree(c->scratch));
        c->scratch = nullptr;
        c->scratch_bytes = 0;
    }
    hipError_t e = hipMalloc(&c->scratch, bytes);
    if (e != hipSuccess) {
        c->scratch = nullptr;
        (void)hipGetLastError();   // handled here: must not resurface as the next call's "kernel launch" error
        set_error("scratch allocation of " + std::to_string(bytes) + " bytes failed: " + hipGetErrorString(e));
        return P2E_E_NOMEM;
    }
    c->scratch_bytes = bytes;
    return 0;
}

// finish a call: fetch the flagged count (synchronously unless the context is async)
static long finish_call(p2e_ctx* c) {
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) {
        set_error(std::string("kernel launch: ") + hipGetErrorString(le));
        return P2E_E_HIP;
    }
    HIP_TRY(hipMemcpyAsync(c->h_counter, c->d_counter, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    if (c->flags & P2E_CTX_ASYNC) return 0;
    int r = p2e_sync(c);
    return r;
}

// ---- host-pointer staging (slow path, used by ctypes-only callers) ----------------------------------
struct Staged {
    p2e_ctx* c;
    DeviceGuard guard;
    std::vector<void*> dev;
    struct Out {
        void* host;
        void* dev;
        size_t bytes;
    };
    std::vector<Out> outs;
    bool host;
    bool finished = false;
    int rc = 0;
    explicit Staged(p2e_ctx* ctx) : c(ctx), guard(ctx->device), host((ctx->flags & P2E_CTX_HOST_POINTERS) != 0) {}
    Staged(const Staged&) = delete;
    Staged& operator=(const Staged&) = delete;
    void* stage(size_t bytes) {
        void* d = nullptr;
        hipError_t e = hipMalloc(&d, bytes ? bytes : 1);
        if (e != hipSuccess) {
            (void)hipGetLastError();   // handled: must not resurface as the next call's launch error
            if (!rc) set_error("staging allocation of " + std::to_string(bytes) + " bytes failed: " + hipGetErrorString(e));
            rc = P2E_E_NOMEM;
            return nullptr;
        }
        dev.push_back(d);
        return d;
    }
    template <class T>
    const T* in(const T* p, size_t bytes) {
        if (!host || !p) return p;
        if (rc) return nullptr;        // an earlier buffer failed: do not touch the host side again
        void* d = stage(bytes);
        if (!d) return nullptr;
        hipError_t e = hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            set_error(std::string("staging copy to the device failed: ") + hipGetErrorString(e));
            rc = P2E_E_HIP;
        }
        return (const T*)d;
    }
    template <class T>
    T* out(T* p, size_t bytes) {
        if (!host || !p) return p;
        if (rc) return nullptr;
        void* d = stage(bytes);
        if (!d) return nullptr;
        outs.push_back({p, d, bytes});
        return (T*)d;
    }
    // every exit of a staged entry point ends here: done(r) on the normal paths, the destructor on early error
    // returns (HIP_TRY / ZERO_COUNTER).  A failed call may have queued work on the internal streams: nothing of
    // it may still be running when the staged buffers are freed or the scratch is reused by the next call.
    void release(bool failed) {
        if (failed) {
            for (hipStream_t st : {c->st_msm, c->st_fixed, c->st_binv, c->st_c2, c->st_c1, c->st_c1q, c->st_c2q})
                if (st) (void)hipStreamSynchronize(st);
            (void)hipStreamSynchronize(c->stream);
            (void)hipGetLastError();
        }
        if (!dev.empty()) {
            if (!failed) (void)hipStreamSynchronize(c->stream);
            for (void* d : dev) (void)hipFree(d);
            dev.clear();
        }
        finished = true;
    }
    long done(long r) {
        if (host && r >= 0)
            for (auto& o : outs)
                if (hipMemcpyAsync(o.host, o.dev, o.bytes, hipMemcpyDeviceToHost, c->stream) != hipSuccess) {
                    set_error("staging copy to the host failed");
                    r = P2E_E_HIP;
                }
        release(r < 0);
        return r;
    }
    ~Staged() {
        if (!finished) release(true);
    }
};

static bool bad_common(p2e_ctx* c, size_t n, size_t ld) {
    if (!c) {
        set_error("null context");
        return true;
    }
    if (ld < n) {
        set_error("ld < n");
        return true;
    }
    return false;
}
static inline dim3 grid1(size_t n) { return dim3((unsigned)((n + BS - 1) / BS)); }
#define ZERO_COUNTER(c) HIP_TRY(hipMemsetAsync((c)->d_counter, 0, sizeof(unsigned long long), (c)->stream))

// ====================================================================================================
// single generators
// ====================================================================================================
#if P2E_HAS(0)
extern "C" long p2e_mul_witness_batch(p2e_ctx* c, int field, const uint64_t* x, const uint64_t* y, uint64_t* r,
                                      uint64_t* q, uint64_t* cs, uint64_t* b, size_t n, size_t ld, uint8_t* err) {
    if (bad_common(c, n, ld) || !x || !y || !r || !q || !cs || !b || !err || (field < 0 || field > 3)) return P2E_E_INVALID;
    if (n == 0) return 0;
    Staged S(c);
    size_t cb = ld * 8;
    x = S.in(x, 9 * cb);
    y = S.in(y, 9 * cb);
    r = S.out(r, 9 * cb);
    q = S.out(q, 9 * cb);
    cs = S.out(cs, 17 * cb);
    b = S.out(b, 16 * cb);
    err = S.out(err, n);
    if (S.rc) return S.done(S.rc);
    ZERO_COUNTER(c);
    if (field == P2E_FIELD_BASE)
        hipLaunchKernelGGL(k_mul<ModP>, grid1(n), dim3(BS), 0, c->stream, x, y, r, q, cs, b, n, ld, err, c->d_counter);
    else if (field == P2E_FIELD_SCALAR)
        hipLaunchKernelGGL(k_mul<ModN>, grid1(n), dim3(BS), 0, c->stream, x, y, r, q, cs, b, n, ld, err, c->d_counter);
    else if (field == P2E_FIELD_P256_BASE)
        hipLaunchKernelGGL(k_mul<ModP256>, grid1(n), dim3(BS), 0, c->stream, x, y, r, q, cs, b, n, ld, err, c->d_counter);
    else
        hipLaunchKernelGGL(k_mul<ModN256>, grid1(n), dim3(BS), 0, c->stream, x, y, r, q, cs, b, n, ld, err, c->d_counter);
    return S.done(finish_call(c));
}
extern "C" long p2e_checksum_witness_batch(p2e_ctx* c, const uint64_t* a, uint64_t* b, size_t n, size_t ld, uint8_t* err) {
    if (bad_common(c, n, ld) || !a || !b || !err) return P2E_E_INVALID;
    if (n == 0) return 0;
    Staged S(c);
    a = S.in(a, 17 * ld * 8);
    b = S.out(b, 16 * ld * 8);
    err = S.out(err, n);
    if (S.rc) return S.done(S.rc);
    ZERO_COUNTER(c);
    hipLaunchKernelGGL(k_checksum, grid1(n), dim3(BS), 0, c->stream, a, b, n, ld, err, c->d_counter);
    return S.done(finish_call(c));
}
static long addsub(p2e_ctx* c, int field, bool is_sub, const uint64_t* a, const uint64_t* b, uint64_t* out,
                   uint64_t* ov, size_t n, size_t ld, uint8_t* err) {
    if (bad_common(c, n, ld) || !a || !b || !out || !ov || !err || (field < 0 || field > 3)) return P2E_E_INVALID;
    if (n == 0) return 0;
    Staged S(c);
    a = S.in(a, 9 * ld * 8);
    b = S.in(b, 9 * ld * 8);
    out = S.out(out, 9 * ld * 8);
    ov = S.out(ov, n * 8);
    err = S.out(err, n);
    if (S.rc) return S.done(S.rc);
    ZERO_COUNTER(c);
    dim3 g = grid1(n), bs(BS);
    if (field == 0 && !is_sub) hipLaunchKernelGGL((k_addsub<ModP, false>), g, bs, 0, c->stream, a, b, out, ov, n, ld, err, c->d_counter);
    if (field == 0 && is_sub) hipLaunchKernelGGL((k_addsub<ModP, true>), g, bs, 0, c->stream, a, b, out, ov, n, ld, err, c->d_counter);
    if (field == 1 && !is_sub) hipLaunchKernelGGL((k_addsub<ModN, false>), g, bs, 0, c->stream, a, b, out, ov, n, ld, err, c->d_counter);
    if (field == 1 && is_sub) hipLaunchKernelGGL((k_addsub<ModN, true>), g, bs, 0, c->stream, a, b, out, ov, n, ld, err, c->d_counter);
    if (field == 2 && !is_sub) hipLaunchKernelGGL((k_addsub<ModP256, false>), g, bs, 0, c->stream, a, b, out, ov, n, ld, err, c->d_counter);
    if (field == 2 && is_sub) hipLaunchKernelGGL((k_addsub<ModP256, true>), g, bs, 0, c->stream, a, b, out, ov, n, ld, err, c->d_counter);
    if (field == 3 && !is_sub) hipLaunchKernelGGL((k_addsub<ModN256, false>), g, bs, 0, c->stream, a, b, out, ov, n, ld, err, c->d_counter);
    if (field == 3 && is_sub) hipLaunchKernelGGL((k_addsub<ModN256, true>), g, bs, 0, c->stream, a, b, out, ov, n, ld, err, c->d_counter);
    return S.done(finish_call(c));
}
extern "C" long p2e_add_witness_batch(p2e_ctx* c, int field, const uint64_t* a, const uint64_t* b, uint64_t* sum,
                                      uint64_t* ov, size_t n, size_t ld, uint8_t* err) {
    return addsub(c, field, false, a, b, sum, ov, n, ld, err);
}
extern "C" long p2e_sub_witness_batch(p2e_ctx* c, int field, const uint64_t* a, const uint64_t* b, uint64_t* diff,
                                      uint64_t* ov, size_t n, size_t ld, uint8_t* err) {
    return addsub(c, field, true, a, b, diff, ov, n, ld, err);
}
extern "C" long p2e_add_many_witness_batch(p2e_ctx* c, int field, const uint64_t* s, int k, uint64_t* sum,
                                           uint64_t* ov, size_t n, size_t ld, uint8_t* err) {
    if (bad_common(c, n, ld) || !s || !sum || !ov || !err || (field < 0 || field > 3) || k < 1 || k > 8) return P2E_E_INVALID;
    if (n == 0) return 0;
    Staged S(c);
    s = S.in(s, (size_t)k * 9 * ld * 8);
    sum = S.out(sum, 9 * ld * 8);
    ov = S.out(ov, n * 8);
    err = S.out(err, n);
    if (S.rc) return S.done(S.rc);
    ZERO_COUNTER(c);
    if (field == 0)
        hipLaunchKernelGGL(k_add_many<ModP>, grid1(n), dim3(BS), 0, c->stream, s, k, sum, ov, n, ld, err, c->d_counter);
    else if (field == 1)
        hipLaunchKernelGGL(k_add_many<ModN>, grid1(n), dim3(BS), 0, c->stream, s, k, sum, ov, n, ld, err, c->d_counter);
    else if (field == 2)
        hipLaunchKernelGGL(k_add_many<ModP256>, grid1(n), dim3(BS), 0, c->stream, s, k, sum, ov, n, ld, err, c->d_counter);
    else
        hipLaunchKernelGGL(k_add_many<ModN256>, grid1(n), dim3(BS), 0, c->stream, s, k, sum, ov, n, ld, err, c->d_counter);
    return S.done(finish_call(c));
}
extern "C" long p2e_inv_witness_batch(p2e_ctx* c, int field, const uint64_t* x, uint64_t* inv, uint64_t* div,
                                      size_t n, size_t ld, uint8_t* err) {
    if (bad_common(c, n, ld) || !x || !inv || !div || !err || (field < 0 || field > 3)) return P2E_E_INVALID;
    if (n == 0) return 0;
    Staged S(c);
    x = S.in(x, 9 * ld * 8);
    inv = S.out(inv, 9 * ld * 8);
    div = S.out(div, 9 * ld * 8);
    err = S.out(err, n);
    if (S.rc) return S.done(S.rc);
    ZERO_COUNTER(c);
    if (field == 0)
        hipLaunchKernelGGL(k_inv<ModP>, grid1(n), dim3(BS), 0, c->stream, x, inv, div, n, ld, err, c->d_counter);
    else if (field == 1)
        hipLaunchKernelGGL(k_inv<ModN>, grid1(n), dim3(BS), 0, c->stream, x, inv, div, n, ld, err, c->d_counter);
    else if (field == 2)
        hipLaunchKernelGGL(k_inv<ModP256>, grid1(n), dim3(BS), 0, c->stream, x, inv, div, n, ld, err, c->d_counter);
    else
        hipLaunchKernelGGL(k_inv<ModN256>, grid1(n), dim3(BS), 0, c->stream, x, inv, div, n, ld, err, c->d_counter);
    return S.done(finish_call(c));
}
extern "C" long p2e_biguint_div_rem_batch(p2e_ctx* c, const uint64_t* a, int na, const uint64_t* b, int nb, uint64_t* div,
                                          uint64_t* rem, size_t n, size_t ld, uint8_t* err) {
    const int nd = nb > na + 1 ? 0 : na - nb + 1;
    if (bad_common(c, n, ld) || na < 1 || na > 18 || nb < 1 || nb > 9 || !a || !b || (nd > 0 && !div) || !rem || !err)
        return P2E_E_INVALID;
    if (n == 0) return 0;
    Staged S(c);
    a = S.in(a, (size_t)na * ld * 8);
    b = S.in(b, (size_t)nb * ld * 8);
    if (nd > 0) div = S.out(div, (size_t)nd * ld * 8);
    rem = S.out(rem, (size_t)nb * ld * 8);
    err = S.out(err, n);
    if (S.rc) return S.done(S.rc);
    ZERO_COUNTER(c);
    hipLaunchKernelGGL(k_div_rem, grid1(n), dim3(BS), 0, c->stream, a, na, b, nb, div, rem, n, ld, err, c->d_counter);
    return S.done(finish_call(c));
}
extern "C" long p2e_glv_decompose_batch(p2e_ctx* c, const uint64_t* k, uint64_t* k1, uint64_t* k2, uint64_t* n1,
                                        uint64_t* n2, size_t n, size_t ld, uint8_t* err) {
    if (bad_common(c, n, ld) || !k || !k1 || !k2 || !n1 || !n2 || !err) return P2E_E_INVALID;
    if (n == 0) return 0;
    Staged S(c);
    k = S.in(k, 9 * ld * 8);
    k1 = S.out(k1, 5 * ld * 8);
    k2 = S.out(k2, 5 * ld * 8);
    n1 = S.out(n1, n * 8);
    n2 = S.out(n2, n * 8);
    err = S.out(err, n);
    if (S.rc) return S.done(S.rc);
    ZERO_COUNTER(c);
    hipLaunchKernelGGL(k_glv, grid1(n), dim3(BS), 0, c->stream, k, k1, k2, n1, n2, n, ld, err, c->d_counter);
    return S.done(finish_call(c));
}
extern "C" long p2e_limb_split(p2e_ctx* c, const uint8_t* packed, uint64_t* limbs, size_t n, size_t ld) {
    if (bad_common(c, n, ld) || !packed || !limbs) return P2E_E_INVALID;
    if (n == 0) return 0;
    Staged S(c);
    packed = S.in(packed, n * 32);
    limbs = S.out(limbs, 9 * ld * 8);
    if (S.rc) return S.done(S.rc);
    ZERO_COUNTER(c);
    size_t pairs = (n + 1) / 2;
    size_t blocks = (pairs + BS - 1) / BS;
    if (blocks > 256 * 16) blocks = 256 * 16;  // grid-stride above 16 workgroups per CU
    hipLaunchKernelGGL(k_split, dim3((unsigned)blocks), dim3(BS), 0, c->stream, packed, limbs, n, ld);
    return S.done(finish_call(c));
}
extern "C" long p2e_limb_pack(p2e_ctx* c, const uint64_t* limbs, uint8_t* packed, size_t n, size_t ld, uint8_t* err) {
    if (bad_common(c, n, ld) || !packed || !limbs || !err) return P2E_E_INVALID;
    if (n == 0) return 0;
    Staged S(c);
    limbs = S.in(limbs, 9 * ld * 8);
    packed = S.out(packed, n * 32);
    err = S.out(err, n);
    if (S.rc) return S.done(S.rc);
    ZERO_COUNTER(c);
    hipLaunchKernelGGL(k_pack, grid1(n), dim3(BS), 0, c->stream, limbs, packed, n, ld, err, c->d_counter);
    return S.done(finish_call(c));
}
#endif   // P2E_HAS(0)

// ====================================================================================================
// fused schedules
// ====================================================================================================
#if P2E_HAS(1)
// cols != nullptr: the u64 column matrix; otherwise the compact container (narrow, wide)
static long run_program(p2e_ctx* c, int program, const uint8_t* msg, const uint8_t* r, const uint8_t* s,
                        const uint8_t* pkx, const uint8_t* pky, uint64_t* cols, size_t n, size_t ld, uint8_t* err,
                        uint8_t* valid, uint32_t* narrow = nullptr, size_t ldn = 0, uint64_t* wide = nullptr, size_t ldw = 0,
                        bool verify_only = false) {
    const bool compact = cols == nullptr && !verify_only;
    const DeviceProgram& DP = c->progs[program];
    const Program& G = DP.prog;
    Staged S(c);
    msg = S.in(msg, n * 32);
    r = S.in(r, n * 32);
    s = S.in(s, n * 32);
    pkx = S.in(pkx, n * 32);
    pky = S.in(pky, n * 32);
    if (compact) {
        narrow = S.out(narrow, (size_t)DP.num_narrow * ldn * 4);
        wide = S.out(wide, (size_t)DP.num_wide * ldw * 8);
    } else if (!verify_only) {
        cols = S.out(cols, (size_t)G.num_cols * ld * 8);
    }
    err = S.out(err, n);
    valid = S.out(valid, n);
    if (S.rc) return S.done(S.rc);
    ScratchLayout L = scratch_layout(G, n);
    int rc = ensure_scratch(c, L.total);
    if (rc) return S.done(rc);
    char* base = (char*)c->scratch;
    Buffers B{};
    B.msg = msg;
    B.r = r;
    B.s = s;
    B.pkx = pkx;
    B.pky = pky;
    B.sink = Sink{cols, ld, narrow, ldn, wide, ldw, DP.d_wide_before};
    B.n = n;
    B.err = (u32*)(base + L.err32);
    B.valid = (uint8_t*)(base + L.valid8);
    B.PX = (U256*)(base + L.px);
    B.PY = (U256*)(base + L.py);
    B.PZ = (U256*)(base + L.pz);
    B.PW = (U256*)(base + L.pw);
    B.PREF = (U256*)(base + L.pref);
    B.AX = (U256*)(base + L.ax);
    B.AY = (U256*)(base + L.ay);
    B.dig4 = (uint8_t*)(base + L.dig4);
    B.dig2 = (uint8_t*)(base + L.dig2);
    B.msrc = (uint16_t*)(base + L.msrc);
    B.dyn = (uint16_t*)(base + L.dyn);
    B.src = (uint16_t*)(base + L.src);
    B.cpts = c->d_cpts;
    B.fbtab = c->d_fbtab;
    // three regimes: four lanes per signature (n <= quad_max_n), lane per signature with phase B on two streams and longer
    // runs in fewer pieces (below binv_alt_max_n), and the large-batch plan
    const bool mid_plan = n > c->quad_max_n && n < c->binv_alt_max_n;
    const bool quad_plan = n <= c->quad_max_n;
    int run_iters = 0;
    B.ops = DP.d_ops_plain;
    if (n >= c->runs_min_n) {
        run_iters = mid_plan ? c->run_iters_mid : c->run_iters;
        if (run_iters > 0) B.ops = mid_plan ? DP.d_ops_mid : DP.d_ops;
    } else if (quad_plan && c->run_iters_small > 0) {
        run_iters = c->run_iters_small;
        B.ops = DP.d_ops_small;
    }
    ZERO_COUNTER(c);
    unsigned gx = (unsigned)((n + BS - 1) / BS);
    c->n_expand = 0;
    c->seg_blocks.clear();
    if (verify_only) {
        // the native verification alone: scalar phase without emission, the two chains side by side in Jacobian
        // coordinates (no batch inversion, no expansion), the final add, r == x on its Jacobian result
        B.ops = DP.d_ops_plain;
        hipLaunchKernelGGL(k_scalar<4>, dim3(gx), dim3(BS), 0, c->stream, G, B, (size_t)0);
        HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
        HIP_TRY(hipStreamWaitEvent(c->st_fixed, c->ev_fork, 0));
        hipLaunchKernelGGL(k_chains, dim3(gx), dim3(BS), 0, c->st_fixed, G, B, G.chain_begin[1], G.chain_end[1], 0, 0);
        HIP_TRY(hipEventRecord(c->ev_fixed, c->st_fixed));
        hipLaunchKernelGGL(k_chains, dim3(gx), dim3(BS), 0, c->stream, G, B, G.chain_begin[0], G.chain_end[0], 0, 0);
        HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_fixed, 0));
        hipLaunchKernelGGL(k_chains, dim3(gx), dim3(BS), 0, c->stream, G, B, G.chain_begin[2], G.chain_end[2], 0, 0);
        hipLaunchKernelGGL(k_verify_check, dim3(gx), dim3(BS), 0, c->stream, G, B);
        hipLaunchKernelGGL(k_finalize, dim3(gx), dim3(BS), 0, c->stream, B.err, B.valid, err, valid, n, c->d_counter);
        c->have_phases = false;
        return S.done(finish_call(c));
    }
    // 16-byte column stores need full workgroups, an even column stride and a 16-byte aligned matrix
    // paired stores need full workgroups, even column strides and matrices aligned to two elements
    const bool wide_ok = !getenv("P2E_NARROW_STORES") &&
                         (compact ? (ldn % 2 == 0 && ldw % 2 == 0 && (reinterpret_cast<uintptr_t>(narrow) & 7) == 0 &&
                                     (reinterpret_cast<uintptr_t>(wide) & 15) == 0)
                                  : (ld % 2 == 0 && (reinterpret_cast<uintptr_t>(cols) & 15) == 0));
    // kernel variant per launch: EmitOf<MODE>
#define LAUNCH_EMIT(KERNEL, PAIRED, GRID, STREAM, ...)                                                    \
    do {                                                                                                  \
        if (compact) {                                                                                    \
            if (PAIRED) hipLaunchKernelGGL(KERNEL<3>, GRID, dim3(BS), emit_lds, STREAM, __VA_ARGS__);     \
            else hipLaunchKernelGGL(KERNEL<2>, GRID, dim3(BS), emit_lds, STREAM, __VA_ARGS__);            \
        } else {                                                                                          \
            if (PAIRED) hipLaunchKernelGGL(KERNEL<1>, GRID, dim3(BS), emit_lds, STREAM, __VA_ARGS__);     \
            else hipLaunchKernelGGL(KERNEL<0>, GRID, dim3(BS), emit_lds, STREAM, __VA_ARGS__);            \
        }                                                                                                 \
    } while (0)
    unsigned emit_lds = 0;   // dynamic LDS of the emitting kernels (only ever non-zero for the expansions of the small-batch plan)
    const size_t n_wide = wide_ok ? (n / BS) * BS : 0;
    const unsigned gx_wide = (unsigned)(n_wide / BS);
    const unsigned gx_tail = (unsigned)((n - n_wide + BS - 1) / BS);
    // ---- launch plan ------------------------------------------------------------------------------------
    // chains (latency-bound, sequential) : their own high-priority streams, cut into pieces
    // phase B of a piece                 : the fixed-base chain's stream, once the piece's chain kernel is done
    // phase C of a piece                 : the caller's stream, after its phase B (HBM-bound)
    // so that the HBM-bound expansion of finished pieces hides the chains and inversions of later ones.
    struct Seg {
        int lo, hi, order;   // ops for phases A and B; readiness estimate (ops walked on its chain before it completes)
        hipStream_t chain_stream;
        bool final_after;    // append the final add (needs the fixed-base chain) to this piece
        int s_lo, s_hi;      // ops expanded one by one (k_expand)
        int it0, it1;        // MSM-loop iterations expanded as runs (k_expand_runs)
        bool fb_run;         // the fixed-base windows of this piece expanded as one run per signature (k_expand_fb_run)
    };
    Seg segs[p2e_ctx::MAX_SEG];
    int ns = 0;
    const bool verify = G.num_chains == 3;
    // small batches: four lanes per signature in phase A, split inversion batches in phase B (quad.hpp)
    const bool quad = n <= c->quad_max_n;
    const int msm_pieces = quad ? c->msm_pieces_small : mid_plan ? c->msm_pieces_mid : c->msm_pieces;
    // with run expansion the fixed-base chain is ONE piece: its windows keep no X, Y / affine form in memory
    // (F_NO_AFFINE), so nothing could resume the chain from scratch in the middle
    const bool fb_run = run_iters > 0 && verify && c->fb_run && G.fb_begin == G.chain_begin[1];
    const int fixed_pieces = fb_run ? 1 : quad ? c->fixed_pieces_small : c->fixed_pieces;
    auto cut = [&](int lo, int hi, int pieces, hipStream_t st) {
        if (pieces > hi - lo) pieces = hi - lo;
        int a = lo;
        for (int k = 0; k < pieces; k++) {
            int rem = pieces - k;
            int len = (hi - a + rem - 1) / rem;
            segs[ns++] = Seg{a, a + len, a + len - lo, st, false, a, a + len, 0, 0, false};
            a += len;
        }
    };
    if (verify) cut(G.chain_begin[1], G.chain_end[1], fixed_pieces, c->st_fixed);
    if (fb_run) {   // windows as one run, then the unblinding add op by op
        segs[0].fb_run = true;
        segs[0].s_lo = G.fb_begin + G.fb_windows;
    }
    const int first_msm = ns;
    {
        // MSM chain = window table (its own piece: inverted first, read in affine form by everything after) +
        // the loop, cut at run boundaries into msm_pieces - 1 groups + the trailing unblinding add
        const int lo0 = G.chain_begin[0], hi0 = G.chain_end[0];
        const int lb = G.msm_loop_begin, iters = G.msm_loop_iters, le = lb + 3 * iters;
        const int R = run_iters > 0 ? run_iters : 1;
        const int nruns = (iters + R - 1) / R;
        int groups = msm_pieces > 1 ? msm_pieces - 1 : 1;
        if (groups > nruns) groups = nruns;
        // small-batch plan: an explicit list of runs per piece (the list is used while it fits; the last piece takes the rest)
        int listed = 0;
        // default list for the runs of 4 iterations (19 runs): 6, 5, 5, 2 and the last run alone -- the last piece's inversion
        // batch and expansion are the exposed tail of the call (2.27 / 2.92 against 2.32 / 2.98 ms at 2^13 / 12 288 with
        // equal pieces, profiles/r03_quad_plan_piece_sizes_sweep.txt)
        static const int takes_r4[p2e_ctx::MAX_PIECES + 1] = {6, 5, 5, 2, 0};
        const int* takes = c->small_takes[0] > 0 ? c->small_takes : (quad && run_iters == 4 && iters == MSM_DIGITS) ? takes_r4 : c->small_takes;
        if (quad && takes[0] > 0) {
            int sum = 0;
            while (listed < p2e_ctx::MAX_PIECES - 1 && takes[listed] > 0 && sum + takes[listed] < nruns) sum += takes[listed++];
            groups = listed + 1;
        }
        segs[ns++] = Seg{lo0, lb, lb - lo0, c->st_msm, false, lo0, lb, 0, 0, false};
        int run = 0;
        for (int g = 0; g < groups; g++) {
            int rem = groups - g;
            int take = (nruns - run + rem - 1) / rem;
            if (listed) take = g < listed ? takes[g] : nruns - run;
            int it0 = run * R, it1 = (run + take) * R < iters ? (run + take) * R : iters;
            Seg sg{lb + 3 * it0, lb + 3 * it1, lb + 3 * it1 - lo0, c->st_msm, false, 0, 0, it0, it1, false};
            if (run_iters == 0) {   // no run expansion: op by op
                sg.s_lo = sg.lo;
                sg.s_hi = sg.hi;
                sg.it0 = sg.it1 = 0;
            }
            if (g == groups - 1) {     // trailing ops of the chain (the unblinding add)
                sg.hi = hi0;
                if (run_iters == 0) sg.s_hi = hi0; else { sg.s_lo = le; sg.s_hi = hi0; }
                sg.order = hi0 - lo0;
            }
            segs[ns++] = sg;
            run += take;
        }
    }
    if (verify) segs[ns - 1].final_after = true;
    c->n_seg = ns;

    const unsigned gx4 = (unsigned)((4 * n + BS - 1) / BS);
    const bool alt_b = quad || n < c->binv_alt_max_n;   // phase B of consecutive pieces on two streams
    auto launch_binv = [&](hipStream_t st, int lo, int hi, int have_prefix, bool last_piece = false, bool fixed_piece = false) {
        const int sl = !quad ? (alt_b ? c->binv_mid_split_log2 : 0)
                             : last_piece ? c->binv_split_log2_last : fixed_piece ? c->binv_split_log2_fixed : c->binv_split_log2;
        if ((quad || alt_b) && sl > 0)
            hipLaunchKernelGGL(k_batch_inv_split, dim3((unsigned)(((n << sl) + BS - 1) / BS)), dim3(BS), 0, st, G, B, lo, hi,
                               have_prefix, sl);
        else
            hipLaunchKernelGGL(k_batch_inv, dim3(gx), dim3(BS), 0, st, G, B, lo, hi, have_prefix);
    };
    seg_begin_call(c, DP.h_ops, (u32)G.num_cols);
    if (c->flags & P2E_CTX_PHASE_TIMING) HIP_TRY(hipEventRecord(c->ev[0], c->stream));
    if (gx_wide) LAUNCH_EMIT(k_scalar, true, dim3(gx_wide), c->stream, G, B, (size_t)0);
    if (gx_tail) LAUNCH_EMIT(k_scalar, false, dim3(gx_tail), c->stream, G, B, n_wide);
    HIP_TRY(hipEventRecord(c->ev[1], c->stream));
    HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
    HIP_TRY(hipStreamWaitEvent(c->st_msm, c->ev_fork, 0));
    HIP_TRY(hipStreamWaitEvent(c->st_fixed, c->ev_fork, 0));
    if (alt_b) HIP_TRY(hipStreamWaitEvent(c->st_binv, c->ev_fork, 0));
    // first expansion stream: the library's own if the layout has one (the caller's stream then waits for it at the end)
    hipStream_t st_c1 = c->st_c1 ? c->st_c1 : c->stream;
    hipStream_t st_c2 = c->st_c2;
    const bool own_c1 = c->st_c1 != nullptr || (quad && c->st_c1q);
    if (quad && c->st_c1q) {
        st_c1 = c->st_c1q;
        st_c2 = c->st_c2q;
    }
    if (own_c1) HIP_TRY(hipStreamWaitEvent(st_c1, c->ev_fork, 0));
    // chains
    for (int k = 0; k < ns; k++) {
        Seg& sg = segs[k];
        // once the table piece has been inverted on this stream (below), later pieces read the table affine
        // (small-batch plan: the FIRST loop piece does not wait for the table's inversion -- it reads the table in
        // Jacobian form, one level more per addition, and starts ~0.15 ms earlier; the table's phase B runs beside it)
        const int table_affine = (sg.chain_stream == c->st_msm && k > first_msm + (quad ? 1 : 0)) ? 1 : 0;
        if (quad && k == first_msm + 2) HIP_TRY(hipStreamWaitEvent(c->st_msm, c->ev_binv[first_msm], 0));
        // glv_mul alone: the window table as 2 chains of 4 adds, then 6 and 9 independent adds (body_chain_rows):
        // -6 % at 2^10, -4 % at 2^16.  Not in the verify program: there the fixed-base chain runs beside it, three
        // 186-VGPR waves do not fit a SIMD and the rows only queue (+0.6...1 % measured).
        const bool table_rows = (!verify || quad) && k == first_msm && sg.hi - sg.lo == MSM_TABLE_OPS;
        if (table_rows) {
            hipLaunchKernelGGL(k_chain_rows, dim3(gx, 2), dim3(BS), 0, sg.chain_stream, G, B, sg.lo, 4);
            hipLaunchKernelGGL(k_chain_rows, dim3(gx, 6), dim3(BS), 0, sg.chain_stream, G, B, sg.lo + 8, 1);
            hipLaunchKernelGGL(k_chain_rows, dim3(gx, 9), dim3(BS), 0, sg.chain_stream, G, B, sg.lo + 14, 1);
        } else if (quad) {
            hipLaunchKernelGGL(k_chains_quad, dim3(gx4), dim3(BS), 0, sg.chain_stream, G, B, sg.lo, sg.hi, table_affine, 0);
        } else {
            hipLaunchKernelGGL(k_chains, dim3(gx), dim3(BS), 0, sg.chain_stream, G, B, sg.lo, sg.hi, table_affine, 0);
        }
        if (verify && k == first_msm - 1) HIP_TRY(hipEventRecord(c->ev_fixed, c->st_fixed));
        if (sg.final_after) {
            HIP_TRY(hipStreamWaitEvent(c->st_msm, c->ev_fixed, 0));
            // the final add joins the inversion batch of the last MSM piece
            if (quad)
                hipLaunchKernelGGL(k_chains_quad, dim3(gx4), dim3(BS), 0, c->st_msm, G, B, G.chain_begin[2], G.chain_end[2], 0, 1);
            else
                hipLaunchKernelGGL(k_chains, dim3(gx), dim3(BS), 0, c->st_msm, G, B, G.chain_begin[2], G.chain_end[2], 0, 1);
            sg.hi = G.chain_end[2];
            sg.s_hi = G.chain_end[2];
        }
        HIP_TRY(hipEventRecord(c->ev_piece[k], sg.chain_stream));
        if (k == first_msm) {
            // The first MSM piece is the 23-op table build.  Its phase B runs right here on the chain's own
            // stream (the rest of the chain is not on the critical path: it ends long before the expansion
            // does), so that the first k_expand can start ~0.8 ms earlier than if it queued behind the
            // fixed-base chain on the other stream.
            hipStream_t st_tab = quad ? c->st_binv : c->st_msm;
            if (quad) HIP_TRY(hipStreamWaitEvent(st_tab, c->ev_piece[k], 0));
            launch_binv(st_tab, sg.lo, sg.hi, table_rows ? 0 : 1);
            HIP_TRY(hipEventRecord(c->ev_binv[k], st_tab));
        }
    }
    // order of phases B / C: by readiness
    int order[p2e_ctx::MAX_SEG];
    for (int k = 0; k < ns; k++) order[k] = k;
    for (int a = 1; a < ns; a++)
        for (int b = a; b > 0 && segs[order[b]].order < segs[order[b - 1]].order; b--) {
            int t = order[b];
            order[b] = order[b - 1];
            order[b - 1] = t;
        }
    c->n_expand = 0;
    bool used_c2 = false;
    int last_b[2] = {-1, quad ? first_msm : -1};   // most recent inversion batch on st_fixed / st_binv (the table's: above)
    emit_lds = quad ? c->expand_lds_small : c->expand_lds;
    for (int q = 0; q < ns; q++) {
        const int k = order[q];
        const Seg& sg = segs[k];
        // HIP multiplexes streams onto a few hardware queues (4 by default) and kernels of one queue run
        // in order: a dedicated inversion stream ended up sharing the caller's queue and serialised B
        // with C.  The fixed-base chain's stream is idle after its first ~1 ms, so phase B lives there.
        // (small-batch plan: phase B of consecutive pieces alternates between two streams, so that an inversion batch
        // does not queue behind the previous one -- there the chains are no slower than phase B)
        const bool b_odd = (q & 1) != (quad && c->quad_b_first_on_fixed ? 1 : 0);
        hipStream_t st_b = (alt_b && b_odd) ? c->st_binv : c->st_fixed;
        if (k != first_msm) {
            HIP_TRY(hipStreamWaitEvent(st_b, c->ev_piece[k], 0));
            launch_binv(st_b, sg.lo, sg.hi, 1, k == ns - 1, k < first_msm);
            HIP_TRY(hipEventRecord(c->ev_binv[k], st_b));
            last_b[st_b == c->st_binv ? 1 : 0] = k;
        }
        // (small-batch plan: phase C alternates between the caller's stream and a second one, so that an expansion
        // waiting for its inversion batch does not hold up the expansions queued behind it)
        hipStream_t st_c = (quad && (q & 1)) ? st_c2 : st_c1;
        used_c2 = used_c2 || st_c == st_c2;
        HIP_TRY(hipStreamWaitEvent(st_c, c->ev_binv[k], 0));
        // an expansion also reads affine results of EARLIER pieces (the first operand of its first op, the window
        // table, the fixed-base result under the final add): with one expansion stream the queue order implied
        // their inversion batches, with two it has to be said
        // (the inversion batches sit on two in-order streams: the most recent one of each implies every earlier one, so
        // two waits say what q of them used to -- 28 barrier packets less on the expansion streams of a verify call)
        if (quad) {
            if (c->quad_few_waits) {
                for (int sb = 0; sb < 2; sb++)
                    if (last_b[sb] >= 0 && last_b[sb] != k) HIP_TRY(hipStreamWaitEvent(st_c, c->ev_binv[last_b[sb]], 0));
            } else {
                for (int q2 = 0; q2 < q; q2++) HIP_TRY(hipStreamWaitEvent(st_c, c->ev_binv[order[q2]], 0));
            }
        }
        auto cols_of = [&](int lo, int hi) {
            double cw = 0;
            for (int t = lo; t < hi; t++)
                cw += DP.h_ops[t].kind == OP_DBL ? COLS_DBL : DP.h_ops[t].kind == OP_CADD ? COLS_CADD : COLS_ADD;
            return cw;
        };
        if (sg.it1 > sg.it0) {
            const int R = run_iters;
            const unsigned nr = (unsigned)((sg.it1 - sg.it0 + R - 1) / R);
            const int e = c->n_expand++;
            if (c->flags & P2E_CTX_PHASE_TIMING) HIP_TRY(hipEventRecord(c->ev_c0[e], st_c));
            if (gx_wide) LAUNCH_EMIT(k_expand_runs, true, dim3(gx_wide, nr), st_c, G, B, sg.it0, R, sg.it1, (size_t)0);
            if (gx_tail) LAUNCH_EMIT(k_expand_runs, false, dim3(gx_tail, nr), st_c, G, B, sg.it0, R, sg.it1, n_wide);
            HIP_TRY(hipEventRecord(c->ev_c1[e], st_c));
            c->expand_kind[e] = 1;
            c->expand_cols[e] = cols_of(G.msm_loop_begin + 3 * sg.it0, G.msm_loop_begin + 3 * sg.it1);
            seg_note_expand(c, e, DP.h_ops, G.msm_loop_begin + 3 * sg.it0, G.msm_loop_begin + 3 * sg.it1);
        }
        if (sg.fb_run) {
            const int e = c->n_expand++;
            if (c->flags & P2E_CTX_PHASE_TIMING) HIP_TRY(hipEventRecord(c->ev_c0[e], st_c));
            if (gx_wide) LAUNCH_EMIT(k_expand_fb_run, true, dim3(gx_wide), st_c, G, B, G.fb_begin + G.fb_windows, (size_t)0);
            if (gx_tail) LAUNCH_EMIT(k_expand_fb_run, false, dim3(gx_tail), st_c, G, B, G.fb_begin + G.fb_windows, n_wide);
            HIP_TRY(hipEventRecord(c->ev_c1[e], st_c));
            c->expand_kind[e] = 2;
            c->expand_cols[e] = cols_of(G.fb_begin, G.fb_begin + G.fb_windows);
            seg_note_expand(c, e, DP.h_ops, G.fb_begin, G.fb_begin + G.fb_windows);
        }
        if (sg.s_hi > sg.s_lo) {
            const int e = c->n_expand++;
            if (c->flags & P2E_CTX_PHASE_TIMING) HIP_TRY(hipEventRecord(c->ev_c0[e], st_c));
            if (gx_wide) LAUNCH_EMIT(k_expand, true, dim3(gx_wide, (unsigned)(sg.s_hi - sg.s_lo)), st_c, G, B, sg.s_lo, (size_t)0);
            if (gx_tail) LAUNCH_EMIT(k_expand, false, dim3(gx_tail, (unsigned)(sg.s_hi - sg.s_lo)), st_c, G, B, sg.s_lo, n_wide);
            HIP_TRY(hipEventRecord(c->ev_c1[e], st_c));
            c->expand_kind[e] = 0;
            c->expand_cols[e] = cols_of(sg.s_lo, sg.s_hi);
            seg_note_expand(c, e, DP.h_ops, sg.s_lo, sg.s_hi);
        }
    }
    if (used_c2) {   // join the second expansion stream
        HIP_TRY(hipEventRecord(c->ev_c2, st_c2));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_c2, 0));
    }
    if (own_c1) {
        HIP_TRY(hipEventRecord(c->ev_c1join, st_c1));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_c1join, 0));
    }
    if (c->flags & P2E_CTX_PHASE_TIMING) HIP_TRY(hipEventRecord(c->ev[5], c->stream));
    hipLaunchKernelGGL(k_finalize, dim3(gx), dim3(BS), 0, c->stream, B.err, B.valid, err, valid, n, c->d_counter);
    c->have_phases = true;
    return S.done(finish_call(c));
#undef LAUNCH_EMIT
}

extern "C" long p2e_ecdsa_verify_witness_batch(p2e_ctx* c, const uint8_t* msg32, const uint8_t* r32,
                                               const uint8_t* s32, const uint8_t* pkx32, const uint8_t* pky32,
                                               uint64_t* cols, size_t n, size_t ld, uint8_t* err, uint8_t* valid) {
    if (bad_common(c, n, ld) || !msg32 || !r32 || !s32 || !pkx32 || !pky32 || !cols || !err) return P2E_E_INVALID;
    if (n == 0) return 0;
    return run_program(c, 0, msg32, r32, s32, pkx32, pky32, cols, n, ld, err, valid);
}
extern "C" long p2e_ecdsa_verify_batch(p2e_ctx* c, const uint8_t* msg32, const uint8_t* r32, const uint8_t* s32,
                                       const uint8_t* pkx32, const uint8_t* pky32, size_t n, uint8_t* err, uint8_t* valid) {
    if (bad_common(c, n, n) || !msg32 || !r32 || !s32 || !pkx32 || !pky32 || !err || !valid) return P2E_E_INVALID;
    if (n == 0) return 0;
    return run_program(c, 0, msg32, r32, s32, pkx32, pky32, nullptr, n, n, err, valid, nullptr, 0, nullptr, 0, true);
}
extern "C" long p2e_glv_mul_witness_batch(p2e_ctx* c, const uint8_t* px32, const uint8_t* py32, const uint8_t* k32,
                                          uint64_t* cols, size_t n, size_t ld, uint8_t* err, uint8_t* valid) {
    if (bad_common(c, n, ld) || !px32 || !py32 || !k32 || !cols || !err) return P2E_E_INVALID;
    if (n == 0) return 0;
    return run_program(c, 1, k32, k32, k32, px32, py32, cols, n, ld, err, valid);
}

extern "C" long p2e_ecdsa_verify_witness_compact_batch(p2e_ctx* c, const uint8_t* msg32, const uint8_t* r32, const uint8_t* s32,
                                                       const uint8_t* pkx32, const uint8_t* pky32, uint32_t* narrow, size_t ld_narrow,
                                                       uint64_t* wide, size_t ld_wide, size_t n, uint8_t* err, uint8_t* valid) {
    if (bad_common(c, n, ld_narrow) || ld_wide < n || !msg32 || !r32 || !s32 || !pkx32 || !pky32 || !narrow || !wide || !err)
        return P2E_E_INVALID;
    if (n == 0) return 0;
    return run_program(c, 0, msg32, r32, s32, pkx32, pky32, nullptr, n, 0, err, valid, narrow, ld_narrow, wide, ld_wide);
}
extern "C" long p2e_glv_mul_witness_compact_batch(p2e_ctx* c, const uint8_t* px32, const uint8_t* py32, const uint8_t* k32,
                                                  uint32_t* narrow, size_t ld_narrow, uint64_t* wide, size_t ld_wide, size_t n,
                                                  uint8_t* err, uint8_t* valid) {
    if (bad_common(c, n, ld_narrow) || ld_wide < n || !px32 || !py32 || !k32 || !narrow || !wide || !err) return P2E_E_INVALID;
    if (n == 0) return 0;
    return run_program(c, 1, k32, k32, k32, px32, py32, nullptr, n, 0, err, valid, narrow, ld_narrow, wide, ld_wide);
}
#endif   // P2E_HAS(1)

#if P2E_HAS(0)
extern "C" long p2e_columns_to_rows(p2e_ctx* c, const uint64_t* cols, size_t ld, size_t n, size_t ncols,
                                    uint64_t* rows, size_t row_ld) {
    if (bad_common(c, n, ld) || !cols || !rows || row_ld < ncols) return P2E_E_INVALID;
    if (n == 0 || ncols == 0) return 0;
    Staged S(c);
    cols = S.in(cols, ncols * ld * 8);
    rows = S.out(rows, n * row_ld * 8);
    if (S.rc) return S.done(S.rc);
    ZERO_COUNTER(c);
    dim3 grid((unsigned)((n + 63) / 64), (unsigned)((ncols + 63) / 64));
    hipLaunchKernelGGL(k_transpose<u64>, grid, dim3(BS), 0, c->stream, cols, ld, n, ncols, rows, row_ld);
    return S.done(finish_call(c));
}
extern "C" long p2e_compact_to_rows(p2e_ctx* c, int program, const uint32_t* narrow, size_t ld_narrow, const uint64_t* wide,
                                    size_t ld_wide, size_t n, uint32_t* rows_narrow, size_t row_ld_narrow, uint64_t* rows_wide,
                                    size_t row_ld_wide) {
    if (bad_common(c, n, ld_narrow) || program < 0 || program > 1 || ld_wide < n || !narrow || !wide || !rows_narrow || !rows_wide)
        return P2E_E_INVALID;
    const DeviceProgram& DP = c->progs[program];
    if (row_ld_narrow < DP.num_narrow || row_ld_wide < DP.num_wide) {
        set_error("row stride smaller than the matrix");
        return P2E_E_INVALID;
    }
    if (n == 0) return 0;
    Staged S(c);
    narrow = S.in(narrow, (size_t)DP.num_narrow * ld_narrow * 4);
    wide = S.in(wide, (size_t)DP.num_wide * ld_wide * 8);
    rows_narrow = S.out(rows_narrow, n * row_ld_narrow * 4);
    rows_wide = S.out(rows_wide, n * row_ld_wide * 8);
    if (S.rc) return S.done(S.rc);
    ZERO_COUNTER(c);
    const unsigned gx = (unsigned)((n + 63) / 64);
    hipLaunchKernelGGL(k_transpose<u32>, dim3(gx, (DP.num_narrow + 63) / 64), dim3(BS), 0, c->stream, narrow, ld_narrow, n,
                       (size_t)DP.num_narrow, rows_narrow, row_ld_narrow);
    hipLaunchKernelGGL(k_transpose<u64>, dim3(gx, (DP.num_wide + 63) / 64), dim3(BS), 0, c->stream, wide, ld_wide, n,
                       (size_t)DP.num_wide, rows_wide, row_ld_wide);
    return S.done(finish_call(c));
}

extern "C" long p2e_columns_compact(p2e_ctx* c, int program, const uint64_t* cols, size_t ld, size_t n, uint32_t* narrow,
                                    size_t ld_narrow, uint64_t* wide, size_t ld_wide, uint8_t* err) {
    if (bad_common(c, n, ld) || program < 0 || program > 1 || !cols || !narrow || !wide || !err || ld_narrow < n || ld_wide < n) {
        if (c && (ld_narrow < n || ld_wide < n)) set_error("ld_narrow / ld_wide < n");
        return P2E_E_INVALID;
    }
    if (n == 0) return 0;
    const DeviceProgram& DP = c->progs[program];
    Staged S(c);
    cols = S.in(cols, (size_t)DP.prog.num_cols * ld * 8);
    narrow = S.out(narrow, (size_t)DP.num_narrow * ld_narrow * 4);
    wide = S.out(wide, (size_t)DP.num_wide * ld_wide * 8);
    err = S.out(err, n);
    if (S.rc) return S.done(S.rc);
    if (int rc = ensure_scratch(c, n * sizeof(u32))) return S.done(rc);
    ZERO_COUNTER(c);
    u32* err32 = (u32*)c->scratch;
    HIP_TRY(hipMemsetAsync(err32, 0, n * sizeof(u32), c->stream));
    const int pair_ok = ld % 2 == 0 && ld_narrow % 2 == 0 && ld_wide % 2 == 0 && (reinterpret_cast<uintptr_t>(cols) & 15) == 0 &&
                        (reinterpret_cast<uintptr_t>(narrow) & 7) == 0 && (reinterpret_cast<uintptr_t>(wide) & 15) == 0;
    const u32 ncols = (u32)DP.prog.num_cols;
    dim3 grid((unsigned)((n + 2 * BS - 1) / (2 * BS)), (ncols + COMPACT_GROUP - 1) / COMPACT_GROUP);
    hipLaunchKernelGGL(k_compact, grid, dim3(BS), 0, c->stream, cols, ld, n, ncols, DP.d_compact_map, narrow, ld_narrow, wide,
                       ld_wide, err32, pair_ok);
    hipLaunchKernelGGL(k_finalize, dim3((unsigned)((n + BS - 1) / BS)), dim3(BS), 0, c->stream, err32, (const uint8_t*)nullptr,
                       err, (uint8_t*)nullptr, n, c->d_counter);
    c->have_phases = false;
    return S.done(finish_call(c));
}

// aux_u32: the output matrix is u32 (all values are limbs, bits and flags); cols == nullptr: read the compact
// container's narrow matrix instead of the u64 witness matrix
static long run_aux(p2e_ctx* c, int program, const uint8_t* pky32, const uint64_t* cols, size_t ld, const uint32_t* narrow,
                    size_t ldn, void* aux, bool aux_u32, size_t ld_aux, size_t n, uint8_t* err) {
    const DeviceProgram& DP = c->progs[program];
    Staged S(c);
    pky32 = S.in(pky32, 32 * n);
    if (cols) cols = S.in(cols, (size_t)DP.prog.num_cols * ld * 8);
    if (narrow) narrow = S.in(narrow, (size_t)DP.num_narrow * ldn * 4);
    aux = S.out((char*)aux, (size_t)DP.aux_tab.num_aux_cols * ld_aux * (aux_u32 ? 4 : 8));
    err = S.out(err, n);
    if (S.rc) return S.done(S.rc);
    if (int rc = ensure_scratch(c, n * sizeof(u32))) return S.done(rc);
    ZERO_COUNTER(c);
    u32* err32 = (u32*)c->scratch;
    HIP_TRY(hipMemsetAsync(err32, 0, n * sizeof(u32), c->stream));
    AuxArgs A{cols, ld, aux, ld_aux, n, pky32, c->d_cpts, c->d_fbtab, DP.d_aux_items, DP.d_aux_tab, err32,
              narrow, ldn, DP.d_wide_before};
    const unsigned gx = (unsigned)((n + BS - 1) / BS), items = (unsigned)DP.aux_items.size();
    // paired column stores for the full workgroups, one element per lane for the ragged tail
    const bool wide_ok = (ld_aux % 2 == 0) && ((reinterpret_cast<uintptr_t>(aux) & (aux_u32 ? 7 : 15)) == 0) &&
                         !getenv("P2E_NARROW_STORES");
    const size_t n_wide = wide_ok ? (n / BS) * BS : 0;
    const dim3 gw((unsigned)(n_wide / BS), items), gt((unsigned)((n - n_wide + BS - 1) / BS), items);
    if (aux_u32) {
        if (n_wide) hipLaunchKernelGGL(k_aux<3>, gw, dim3(BS), 0, c->stream, A, (size_t)0);
        if (n > n_wide) hipLaunchKernelGGL(k_aux<2>, gt, dim3(BS), 0, c->stream, A, n_wide);
    } else {
        if (n_wide) hipLaunchKernelGGL(k_aux<1>, gw, dim3(BS), 0, c->stream, A, (size_t)0);
        if (n > n_wide) hipLaunchKernelGGL(k_aux<0>, gt, dim3(BS), 0, c->stream, A, n_wide);
    }
    hipLaunchKernelGGL(k_finalize, dim3(gx), dim3(BS), 0, c->stream, err32, (const uint8_t*)nullptr, err,
                       (uint8_t*)nullptr, n, c->d_counter);
    c->have_phases = false;
    return S.done(finish_call(c));
}
extern "C" long p2e_aux_witness_batch(p2e_ctx* c, int program, const uint8_t* pky32, const uint64_t* cols, size_t ld,
                                      uint64_t* aux, size_t ld_aux, size_t n, uint8_t* err) {
    if (bad_common(c, n, ld) || program < 0 || program > 1 || !pky32 || !cols || !aux || !err || ld_aux < n) {
        if (c && ld_aux < n) set_error("ld_aux < n");
        return P2E_E_INVALID;
    }
    if (n == 0) return 0;
    return run_aux(c, program, pky32, cols, ld, nullptr, 0, aux, false, ld_aux, n, err);
}
extern "C" long p2e_aux_witness_compact_batch(p2e_ctx* c, int program, const uint8_t* pky32, const uint32_t* narrow,
                                              size_t ld_narrow, uint32_t* aux32, size_t ld_aux, size_t n, uint8_t* err) {
    if (bad_common(c, n, ld_narrow) || program < 0 || program > 1 || !pky32 || !narrow || !aux32 || !err || ld_aux < n) {
        if (c && ld_aux < n) set_error("ld_aux < n");
        return P2E_E_INVALID;
    }
    if (n == 0) return 0;
    return run_aux(c, program, pky32, nullptr, 0, narrow, ld_narrow, aux32, true, ld_aux, n, err);
}

struct p2e_wire_map {
    u32 limit[4] = {0, 0, 0, 0};   // column counts of the four source matrices (witness, aux, ux, gate) of the map's program
    u32* d_src = nullptr;
    u32* d_dst = nullptr;
    size_t count = 0;
    u32 num_wires = 0, degree = 0;
    bool uses[4] = {false, false, false, false};
};
static int make_wire_map(p2e_ctx* c, const u32 limit[4], const p2e_wire_map_entry* entries, size_t count, uint32_t num_wires,
                         uint32_t degree, p2e_wire_map** out) {
    if (!c || !out || (!entries && count) || !num_wires || !degree) return P2E_E_INVALID;
    const u64 cells = (u64)num_wires * degree;
    std::vector<p2e_wire_map_entry> v(entries, entries + count);
    std::stable_sort(v.begin(), v.end(), [](const p2e_wire_map_entry& a, const p2e_wire_map_entry& b) { return a.dst < b.dst; });
    auto m = new p2e_wire_map();
    for (size_t k = 0; k < count; k++) {
        const u32 kind = v[k].src >> 30, col = v[k].src & 0x3FFFFFFFu;
        if (col >= limit[kind] || v[k].dst >= cells || (k && v[k].dst == v[k - 1].dst)) {
            set_error(col >= limit[kind] ? "wire map: source column out of range"
                      : v[k].dst >= cells                          ? "wire map: destination outside num_wires * degree"
                                                                   : "wire map: two entries share a destination");
            delete m;
            return P2E_E_INVALID;
        }
        m->uses[kind] = true;
    }
    std::vector<u32> src(count), dst(count);
    for (size_t k = 0; k < count; k++) {
        src[k] = v[k].src;
        dst[k] = v[k].dst;
    }
    DeviceGuard guard(c->device);
    for (int k = 0; k < 4; k++) m->limit[k] = limit[k];
    m->count = count;
    m->num_wires = num_wires;
    m->degree = degree;
    if (hipMalloc(&m->d_src, sizeof(u32) * (count ? count : 1)) != hipSuccess || hipMalloc(&m->d_dst, sizeof(u32) * (count ? count : 1)) != hipSuccess ||
        hipMemcpy(m->d_src, src.data(), sizeof(u32) * count, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(m->d_dst, dst.data(), sizeof(u32) * count, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(m->d_src);
        (void)hipFree(m->d_dst);
        delete m;
        set_error("wire map: device allocation failed");
        return P2E_E_NOMEM;
    }
    *out = m;
    return 0;
}
extern "C" int p2e_wire_map_create(p2e_ctx* c, int program, const p2e_wire_map_entry* entries, size_t count, uint32_t num_wires,
                                   uint32_t degree, p2e_wire_map** out) {
    if (program < 0 || program > 1) return P2E_E_INVALID;
    const DeviceProgram& HP = host_program(program);
    const u32 limit[4] = {(u32)HP.prog.num_cols, HP.aux_tab.num_aux_cols, HP.num_ux_cols, HP.num_gate_cols};
    return make_wire_map(c, limit, entries, count, num_wires, degree, out);
}
extern "C" void p2e_wire_map_destroy(p2e_ctx* c, p2e_wire_map* m) {
    if (!m) return;
    if (c) {
        DeviceGuard guard(c->device);
        (void)hipStreamSynchronize(c->stream);
        (void)hipFree(m->d_src);
        (void)hipFree(m->d_dst);
    }
    delete m;
}
extern "C" long p2e_assemble_wires(p2e_ctx* c, const p2e_wire_map* m, const uint64_t* cols, size_t ld, const uint64_t* aux,
                                   size_t ld_aux, const void* ux, int ux_u32, size_t ld_ux, const uint64_t* gate, size_t ld_gate,
                                   uint64_t* wires, size_t wire_stride, size_t n) {
    if (!c || !m || !wires || wire_stride < (size_t)m->num_wires * m->degree || (m->uses[0] && (!cols || ld < n)) ||
        (m->uses[1] && (!aux || ld_aux < n)) || (m->uses[2] && (!ux || ld_ux < n)) || (m->uses[3] && (!gate || ld_gate < n))) {
        set_error("p2e_assemble_wires: null matrix the map reads, ld < n, or wire_stride < num_wires * degree");
        return P2E_E_INVALID;
    }
    if (n == 0 || m->count == 0) return 0;
    Staged S(c);
    if (cols) cols = S.in(cols, (size_t)m->limit[0] * ld * 8);
    if (aux) aux = S.in(aux, (size_t)m->limit[1] * ld_aux * 8);
    if (ux) ux = S.in((const char*)ux, (size_t)m->limit[2] * ld_ux * (ux_u32 ? 4 : 8));
    if (gate) gate = S.in(gate, (size_t)m->limit[3] * ld_gate * 8);
    uint64_t* const host_wires = wires;
    void* d_stage = nullptr;
    if (S.host) {   // in-out buffer: positions no entry names keep the caller's values
        d_stage = S.stage(n * wire_stride * 8);
        if (d_stage && hipMemcpyAsync(d_stage, host_wires, n * wire_stride * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess) S.rc = P2E_E_HIP;
        if (d_stage) S.outs.push_back({host_wires, d_stage, n * wire_stride * 8});
        wires = (uint64_t*)d_stage;
    }
    if (S.rc) return S.done(S.rc);
    ZERO_COUNTER(c);
    AssembleArgs A{cols, ld, aux, ld_aux, ux, ld_ux, ux_u32, gate, ld_gate, m->d_src, m->d_dst, m->count, wires, wire_stride, n};
    dim3 grid((unsigned)((n + 63) / 64), (unsigned)((m->count + 63) / 64));
    hipLaunchKernelGGL(k_assemble, grid, dim3(BS), 0, c->stream, A);
    c->have_phases = false;
    return S.done(finish_call(c));
}

static u64 gl_pow_host(u64 a, u64 e) {
    u64 r = 1;
    while (e) {
        if (e & 1) r = gl_mul(r, a);
        a = gl_mul(a, a);
        e >>= 1;
    }
    return r;
}
extern "C" long p2e_gate_internal_batch(p2e_ctx* c, int program, const uint64_t* aux, size_t ld_aux, uint64_t* gate, size_t ld_gate,
                                        size_t n) {
    if (bad_common(c, n, ld_aux) || program < 0 || program > 1 || !aux || !gate || ld_gate < n) {
        if (c && ld_gate < n) set_error("ld_gate < n");
        return P2E_E_INVALID;
    }
    if (n == 0) return 0;
    const DeviceProgram& DP = c->progs[program];
    Staged S(c);
    aux = S.in(aux, (size_t)DP.aux_tab.num_aux_cols * ld_aux * 8);
    gate = S.out(gate, (size_t)DP.num_gate_cols * ld_gate * 8);
    if (S.rc) return S.done(S.rc);
    ZERO_COUNTER(c);
    GateArgs A{};
    A.aux = aux;
    A.ald = ld_aux;
    A.gate = gate;
    A.gld = ld_gate;
    A.n = n;
    A.items = DP.d_gate_items;
    A.inv16[0] = 0;
    for (u64 d = 1; d < 16; d++) A.inv16[d] = gl_pow_host(d, P_GL - 2);
    const unsigned items = (unsigned)DP.gate_items.size();
    const bool wide_ok = (ld_gate % 2 == 0) && ((reinterpret_cast<uintptr_t>(gate) & 15) == 0) && !getenv("P2E_NARROW_STORES");
    const size_t n_wide = wide_ok ? (n / BS) * BS : 0;
    const dim3 gw((unsigned)(n_wide / BS), items), gt((unsigned)((n - n_wide + BS - 1) / BS), items);
    if (n_wide) hipLaunchKernelGGL(k_gate<1>, gw, dim3(BS), 0, c->stream, A, (size_t)0);
    if (n > n_wide) hipLaunchKernelGGL(k_gate<0>, gt, dim3(BS), 0, c->stream, A, n_wide);
    c->have_phases = false;
    return S.done(finish_call(c));
}
extern "C" long p2e_gate_internal_num_cols(int program) {
    if (program < 0 || program > 1) return P2E_E_INVALID;
    return (long)host_program(program).num_gate_cols;
}

extern "C" long p2e_ux_witness_batch(p2e_ctx* c, int program, const uint8_t* msg32, const uint8_t* r32, const uint8_t* s32,
                                     const uint8_t* pkx32, const uint8_t* pky32, const uint64_t* cols, size_t ld,
                                     const uint64_t* aux, size_t ld_aux, void* ux, int ux_u32, size_t ld_ux, size_t n, uint8_t* err) {
    if (bad_common(c, n, ld) || program < 0 || program > 1 || !msg32 || !pkx32 || !pky32 || (program == 0 && (!r32 || !s32)) ||
        !cols || !aux || !ux || !err || ld_aux < n || ld_ux < n) {
        if (c && (ld_aux < n || ld_ux < n)) set_error("ld_aux / ld_ux < n");
        return P2E_E_INVALID;
    }
    if (n == 0) return 0;
    const DeviceProgram& DP = c->progs[program];
    Staged S(c);
    msg32 = S.in(msg32, 32 * n);
    if (r32) r32 = S.in(r32, 32 * n);
    if (s32) s32 = S.in(s32, 32 * n);
    pkx32 = S.in(pkx32, 32 * n);
    pky32 = S.in(pky32, 32 * n);
    cols = S.in(cols, (size_t)DP.prog.num_cols * ld * 8);
    aux = S.in(aux, (size_t)DP.aux_tab.num_aux_cols * ld_aux * 8);
    ux = S.out((char*)ux, (size_t)DP.num_ux_cols * ld_ux * (ux_u32 ? 4 : 8));
    err = S.out(err, n);
    if (S.rc) return S.done(S.rc);
    if (int rc = ensure_scratch(c, n * sizeof(u32))) return S.done(rc);
    ZERO_COUNTER(c);
    u32* err32 = (u32*)c->scratch;
    HIP_TRY(hipMemsetAsync(err32, 0, n * sizeof(u32), c->stream));
    UxArgs A{};
    A.cols = cols;
    A.ld = ld;
    A.aux = aux;
    A.ald = ld_aux;
    A.ux = ux;
    A.uld = ld_ux;
    A.n = n;
    A.in[INPUT_PY] = pky32;
    A.in[INPUT_PX] = pkx32;
    A.in[INPUT_MSG] = msg32;
    A.in[INPUT_R] = r32 ? r32 : msg32;
    A.in[INPUT_S] = s32 ? s32 : msg32;
    A.consts = c->d_constv;
    A.items = DP.d_ux_items;
    A.err = err32;
    const unsigned gx = (unsigned)((n + BS - 1) / BS), items = (unsigned)DP.ux_items.size();
    const bool wide_ok = (ld_ux % 2 == 0) && ((reinterpret_cast<uintptr_t>(ux) & (ux_u32 ? 7 : 15)) == 0) && !getenv("P2E_NARROW_STORES");
    const size_t n_wide = wide_ok ? (n / BS) * BS : 0;
    const dim3 gw((unsigned)(n_wide / BS), items), gt((unsigned)((n - n_wide + BS - 1) / BS), items);
    if (ux_u32) {
        if (n_wide) hipLaunchKernelGGL(k_ux<3>, gw, dim3(BS), 0, c->stream, A, (size_t)0);
        if (n > n_wide) hipLaunchKernelGGL(k_ux<2>, gt, dim3(BS), 0, c->stream, A, n_wide);
    } else {
        if (n_wide) hipLaunchKernelGGL(k_ux<1>, gw, dim3(BS), 0, c->stream, A, (size_t)0);
        if (n > n_wide) hipLaunchKernelGGL(k_ux<0>, gt, dim3(BS), 0, c->stream, A, n_wide);
    }
    hipLaunchKernelGGL(k_finalize, dim3(gx), dim3(BS), 0, c->stream, err32, (const uint8_t*)nullptr, err, (uint8_t*)nullptr, n,
                       c->d_counter);
    c->have_phases = false;
    return S.done(finish_call(c));
}

// ====================================================================================================
// host-only entry points
// ====================================================================================================
extern "C" long p2e_ux_describe(int program, p2e_ux_desc* out, size_t cap) {
    if (program < 0 || program > 1) return P2E_E_INVALID;
    const DeviceProgram& P = host_program(program);
    for (size_t i = 0; i < P.ux_first.size() && i < cap && out; i++) {
        out[i].first_col = P.ux_first[i];
        out[i].num_cols = P.ux_count[i];
    }
    return (long)P.ux_first.size();
}
extern "C" long p2e_ux_num_cols(int program) {
    if (program < 0 || program > 1) return P2E_E_INVALID;
    return (long)host_program(program).num_ux_cols;
}
extern "C" long p2e_aux_describe(int program, p2e_aux_desc* out, size_t cap) {
    if (program < 0 || program > 1) return P2E_E_INVALID;
    const auto& g = host_program(program).aux_gens;
    for (size_t i = 0; i < g.size() && i < cap && out; i++) {
        out[i].kind = g[i].kind;
        out[i].first_col = g[i].col;
        out[i].num_cols = g[i].ncols;
        std::memset(out[i].label, 0, sizeof out[i].label);
        std::strncpy(out[i].label, g[i].label.c_str(), sizeof(out[i].label) - 1);
    }
    return (long)g.size();
}
extern "C" long p2e_compact_layout(int program, uint32_t* col_map, size_t cap, uint32_t* num_narrow, uint32_t* num_wide) {
    if (program < 0 || program > 1) return P2E_E_INVALID;
    const DeviceProgram& P = host_program(program);
    for (size_t i = 0; i < P.compact_map.size() && i < cap && col_map; i++) col_map[i] = P.compact_map[i];
    if (num_narrow) *num_narrow = P.num_narrow;
    if (num_wide) *num_wide = P.num_wide;
    return (long)P.compact_map.size();
}
extern "C" long p2e_aux_num_cols(int program) {
    if (program < 0 || program > 1) return P2E_E_INVALID;
    return host_program(program).aux_tab.num_aux_cols;
}
extern "C" long p2e_schedule_describe(int program, p2e_gen_desc* out, size_t cap) {
    if (program < 0 || program > 1) return P2E_E_INVALID;
    const auto& g = host_program(program).gens;
    for (size_t i = 0; i < g.size() && i < cap && out; i++) {
        out[i].kind = g[i].kind;
        out[i].field = g[i].field;
        out[i].first_col = g[i].col;
        out[i].num_cols = g[i].ncols;
        std::memset(out[i].label, 0, sizeof out[i].label);
        std::strncpy(out[i].label, g[i].label.c_str(), sizeof(out[i].label) - 1);
    }
    return (long)g.size();
}
extern "C" long p2e_schedule_wiring(int program, p2e_gen_wiring* out, size_t cap) {
    if (program < 0 || program > 1) return P2E_E_INVALID;
    const auto& g = host_program(program).gens;
    for (size_t i = 0; i < g.size() && i < cap && out; i++) {
        out[i].num_operands = g[i].nops;
        out[i].range_check = g[i].range_check ? 1 : 0;
        for (int k = 0; k < 4; k++) {
            out[i].src[k] = g[i].src[k];
            out[i].num_limbs[k] = g[i].nl[k];
        }
    }
    return (long)g.size();
}
extern "C" int p2e_wiring_const(uint32_t id, uint8_t out32[32]) {
    if (id >= NUM_CONSTV || !out32) return P2E_E_INVALID;
    const U256 v = host::ScheduleBuilder::const_value(id);
    std::memcpy(out32, v.w, 32);
    u32 l[NL];
    split29(v, l);
    int n = NL;
    while (n > 0 && l[n - 1] == 0) n--;
    return n;
}
extern "C" long p2e_schedule_num_cols(int program) {
    if (program < 0 || program > 1) return P2E_E_INVALID;
    return host_program(program).prog.num_cols;
}
extern "C" int p2e_synth_signatures(uint64_t seed, size_t first, size_t n, uint8_t* msg32, uint8_t* r32, uint8_t* s32,
                                    uint8_t* pkx32, uint8_t* pky32) {
    if (!msg32 || !r32 || !s32 || !pkx32 || !pky32) return P2E_E_INVALID;
#pragma omp parallel for schedule(dynamic, 64)
    for (long long i = 0; i < (long long)n; i++) {
        U256 msg, r, s;
        Aff pk;
        host::synth_signature(seed, first + (size_t)i, msg, r, s, pk);
        std::memcpy(msg32 + 32 * i, msg.w, 32);
        std::memcpy(r32 + 32 * i, r.w, 32);
        std::memcpy(s32 + 32 * i, s.w, 32);
        std::memcpy(pkx32 + 32 * i, pk.x.w, 32);
        std::memcpy(pky32 + 32 * i, pk.y.w, 32);
    }
    return 0;
}
#endif   // P2E_HAS(0)

#include "curve_api.inc"
