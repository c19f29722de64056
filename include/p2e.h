/* p2e -- MI355X-native batch witness generation for plonky2's secp256k1 ECDSA gadget.
 *
 * C ABI of libp2e_hip.so.  These entry points are what a plonky2-ecdsa (Rust) build would bind through
 * `extern "C"` to replace the bodies of its witness generators (see INTEGRATION.md for the stub).
 * Each entry point cites the reference interface it replaces; paths are relative to the reference
 * crate's src/ (Weobe/plonky2-ecdsa).
 *
 * DATA LAYOUT
 *   Goldilocks columns: canonical u64 values (< 2^64 - 2^32 + 1), column-major over the batch:
 *   element i of column c lives at base[c * ld + i] with ld >= n ("SoA": the 64 lanes of a wavefront
 *   are 64 consecutive batch elements, every access is one coalesced 512-byte transaction).
 *   A non-native value is 9 consecutive columns of 29-bit limbs, least significant first
 *   (gadgets/nonnative.rs:32 BITS = 29, gates/mul_nonnative.rs:37-39 num_limbs = 9).
 *   256-bit inputs of the fused entry points are packed 32-byte little-endian, element i at +32*i.
 *
 * MEMORY
 *   All data pointers are DEVICE pointers (hipMalloc / torch tensors) unless the context was created
 *   with P2E_CTX_HOST_POINTERS, in which case they are host pointers and the library stages them
 *   through its own device buffers (PCIe-inclusive; never the benchmarked configuration).
 *   The caller owns every buffer; the library owns only the p2e_ctx (stream, scratch, constant tables).
 *
 * ERRORS
 *   Return value: < 0 API misuse / HIP failure (p2e_last_error() has the text); >= 0 number of batch
 *   elements whose err byte is non-zero.  err[i] is a bit mask of P2E_ERR_* -- the conditions under
 *   which the reference generator panics.  Outputs of a flagged element are unspecified.
 *   Nothing ever unwinds or aborts across this boundary.
 *
 * THREADING  A p2e_ctx is not thread-safe: one context per (host thread, device).  Calls are
 *   synchronous on return unless P2E_CTX_ASYNC is set (then p2e_sync() completes them and returns the
 *   flagged-element count of the last call).  Every entry point runs on the context's device and restores the
 *   calling thread's current device before it returns.  A failed call (negative return) has released everything it
 *   staged and left no work queued on the context's streams; the context stays usable.
 */
#ifndef P2E_H
#define P2E_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P2E_FIELD_BASE 0   /* plonky2 Secp256K1Base   (point coordinates)      */
#define P2E_FIELD_SCALAR 1 /* plonky2 Secp256K1Scalar (msg, r, s, u1, u2, k1, k2) */
#define P2E_FIELD_P256_BASE 2   /* the crate's P256Base   (field/p256_base.rs)   -- single-generator entry points only */
#define P2E_FIELD_P256_SCALAR 3 /* the crate's P256Scalar (field/p256_scalar.rs) */

/* per-element error bits */
#define P2E_ERR_LIMB_RANGE 1       /* limb >= 2^29: gates/mul_nonnative.rs:262,271,275-276; biguint.rs:456,473 */
#define P2E_ERR_VALUE_GE_2_256 2   /* from_noncanonical_biguint panics (template field/p256_base.rs:121-130)  */
#define P2E_ERR_INVERSE_OF_ZERO 4  /* gadgets/nonnative.rs:863 inverse() of zero                               */
#define P2E_ERR_CARRY_RANGE 8      /* gates/mul_nonnative.rs:527 carry not < 2^34                              */
#define P2E_ERR_DIVISION_BY_ZERO 32 /* BigUintDivRemGenerator: b == 0 (BigUint::div_rem panics)                        */
#define P2E_ERR_QUOTIENT_RANGE 16  /* x*y/m does not fit the gate's nine q wires (x, y far above the modulus)  */

/* status codes (negative returns) */
#define P2E_E_INVALID (-1)   /* null pointer, ld < n, bad field id ...      */
#define P2E_E_HIP (-2)       /* a HIP call failed                             */
#define P2E_E_NO_DEVICE (-3) /* no gfx950 device / extension not usable       */
#define P2E_E_NOMEM (-4)

/* context flags */
#define P2E_CTX_HOST_POINTERS 1u
#define P2E_CTX_ASYNC 2u
/* Time every launch of a fused call with HIP events (p2e_last_phase_ms).  Off by default: a timing event is a barrier
 * packet in front of and behind every expansion launch, which costs nothing measurable at 2^16 signatures per call but
 * 6 % at 2^13 (2.04 -> 1.92 ms).  The column-block events of p2e_segments_* exist either way (without timestamps). */
#define P2E_CTX_PHASE_TIMING 4u

#define P2E_VERIFY_COLS 82615  /* hot-path generator outputs of one verify_secp256k1_message_circuit */
#define P2E_GLV_MUL_COLS 65243 /* ... of one glv_mul                                                 */

typedef struct p2e_ctx p2e_ctx;

/* ---- context ------------------------------------------------------------------------------------ */
/* `stream`: a hipStream_t to run on (e.g. torch's current stream), or NULL for a library-owned one.  The
 * library-owned stream is a blocking stream (hipStreamDefault): work the caller queued on the legacy default
 * stream before a call is ordered before it, and default-stream work queued after a call is ordered after it.
 * The fused entry points dispatch on five internal streams (two chain streams, a second inversion stream, two
 * expansion streams) which the context creates -- and binds to their hardware queues with one empty launch each --
 * back to back, in a fixed order; `stream` itself carries only the scalar phase and the finalisation of a fused call and
 * is ordered with the internal streams by events, so the call behaves like any other work queued on `stream`.  Creating
 * a context synchronises the legacy default stream once.  Export GPU_MAX_HW_QUEUES=8 before the process's first HIP call
 * so that the internal streams do not share a hardware queue with the caller's. */
int p2e_ctx_create(int device, unsigned flags, void *stream, p2e_ctx **out);
void p2e_ctx_destroy(p2e_ctx *ctx);
int p2e_sync(p2e_ctx *ctx);
const char *p2e_last_error(void);
/* bytes of device scratch the fused entry points need for a batch of n (allocated lazily, kept) */
size_t p2e_scratch_bytes(int program /*0 verify, 1 glv_mul*/, size_t n);
/* Timing of the last fused call, measured with hipEvents on the context's stream around every launch -- for contexts
 * created with P2E_CTX_PHASE_TIMING; otherwise every duration is 0 and only the launch and column counts are filled in.
 * The schedule is cut into segments whose Jacobian chains run on internal streams underneath; each
 * finished segment is expanded by k_expand (one curve op per workgroup row: window table, fixed-base
 * chain, trailing adds) and/or k_expand_runs (runs of MSM-loop iterations).
 * out[0] = ms of the scalar kernel, out[4] = ms of the whole call;
 * out[1], out[2], out[3] = k_expand: launches, witness columns per signature they wrote, summed ms;
 * out[5], out[6], out[7] = the same three for k_expand_runs; out[8], out[9], out[10] = for k_expand_fb_run (the
 * fixed-base windows as one run per signature: curve programs, and the verifier under P2E_FB_RUN=1).  The curve
 * programs report their kc_* kernels in the same slots.  Returns the number written (<= 12). */
int p2e_last_phase_ms(p2e_ctx *ctx, float *out, int cap);

/* ---- column blocks of a fused call, as they become final (the assembly step) -------- */
/* The reference fills a PartialWitness generator by generator; a batch consumer -- the RCCL exchange that assembles the
 * columns of a sharded batch on every rank, a D2H copy, a prover that starts on finished columns -- need not wait for
 * the whole call either.  Every fused entry point (p2e_ecdsa_verify_witness[_compact]_batch, p2e_glv_mul_witness
 * [_compact]_batch, p2e_curve_mul_witness[_compact]_batch, p2e_p256_verify_witness[_compact]_batch) leaves behind, per launch that writes
 * witness columns, the block of columns [first_col, first_col + num_cols) it completes and a HIP event recorded behind
 * it.  Blocks are disjoint, cover all columns of the program, and are listed in the order the launches were issued
 * (the scalar phase first, then the expansion of every schedule piece as its chain and inversion batch finish).  In
 * the compact container block k is rows [narrow_before(first_col), narrow_before(first_col + num_cols)) of the narrow
 * matrix and the same of the wide one (p2e_compact_layout is monotonic in the column index).
 * p2e_segments_describe: returns the number of blocks of the LAST fused call issued on ctx (valid from the moment that
 * call returns -- with P2E_CTX_ASYNC that is before anything has run -- until the next call on ctx).
 * p2e_segment_stream_wait: `stream` (a hipStream_t of the same device) waits until block k is final; nothing blocks on
 * the host.  p2e_segment_sync: the host blocks until block k is final.  Both return 0, or P2E_E_INVALID / P2E_E_HIP. */
typedef struct p2e_segment_desc {
    uint32_t first_col, num_cols;
} p2e_segment_desc;
long p2e_segments_describe(p2e_ctx *ctx, p2e_segment_desc *out, size_t cap);
int p2e_segment_stream_wait(p2e_ctx *ctx, int segment, void *stream);
int p2e_segment_sync(p2e_ctx *ctx, int segment);

/* ---- single generators (one reference run_once body each) ----------------------------------------- */
/* MulNonnativeGenerator::run_once gates/mul_nonnative.rs:249-324 followed by
 * CheckSumGenerator::run_once :513-531.  x, y: the gate's 9+9 input wires.  Outputs in gate wire
 * order (:41-59, :384-390): r[9], q[9], check_sum[17], then b[16] of the CheckSumGate row. */
long p2e_mul_witness_batch(p2e_ctx *ctx, int field, const uint64_t *x, const uint64_t *y, uint64_t *r,
                           uint64_t *q, uint64_t *check_sum, uint64_t *b, size_t n, size_t ld, uint8_t *err);
/* CheckSumGenerator::run_once alone on arbitrary a[17] (true Goldilocks division by 2^29). */
long p2e_checksum_witness_batch(p2e_ctx *ctx, const uint64_t *a, uint64_t *b, size_t n, size_t ld,
                                uint8_t *err);
/* NonNativeAdditionGenerator::run_once gadgets/nonnative.rs:626-645: sum[9], overflow. */
long p2e_add_witness_batch(p2e_ctx *ctx, int field, const uint64_t *a, const uint64_t *b, uint64_t *sum,
                           uint64_t *overflow, size_t n, size_t ld, uint8_t *err);
/* NonNativeSubtractionGenerator::run_once gadgets/nonnative.rs:792-810: diff[9], overflow. */
long p2e_sub_witness_batch(p2e_ctx *ctx, int field, const uint64_t *a, const uint64_t *b, uint64_t *diff,
                           uint64_t *overflow, size_t n, size_t ld, uint8_t *err);
/* NonNativeMultipleAddsGenerator::run_once gadgets/nonnative.rs:696-728: summands[k][9][ld], 1 <= k <= 8. */
long p2e_add_many_witness_batch(p2e_ctx *ctx, int field, const uint64_t *summands, int k, uint64_t *sum,
                                uint64_t *overflow, size_t n, size_t ld, uint8_t *err);
/* NonNativeInverseGenerator::run_once gadgets/nonnative.rs:857-872: inv[9], div[9]. */
long p2e_inv_witness_batch(p2e_ctx *ctx, int field, const uint64_t *x, uint64_t *inv, uint64_t *div, size_t n,
                           size_t ld, uint8_t *err);
/* BigUintDivRemGenerator::run_once gadgets/biguint.rs:508-518 (the generator behind rem_biguint / reduce,
 * gadgets/nonnative.rs:539-548; not on the ECDSA path, here so that every generator of the crate has an entry):
 * a[na][ld] (1 <= na <= 18 limbs), b[nb][ld] (1 <= nb <= 9) -> div[max(0, na - nb + 1)][ld], rem[nb][ld], the limb
 * counts div_rem_biguint allocates (:391-397).  err: a limb >= 2^29 (LIMB_RANGE; the reference would sum arbitrary
 * field elements), b == 0 (DIVISION_BY_ZERO), a quotient set_biguint_target rejects (LIMB_RANGE; this includes
 * the reference's convert_base quirk: a D-digit value is handed over as floor(32 D / 29) limbs, zero ones included). */
long p2e_biguint_div_rem_batch(p2e_ctx *ctx, const uint64_t *a, int na, const uint64_t *b, int nb, uint64_t *div,
                               uint64_t *rem, size_t n, size_t ld, uint8_t *err);
/* GLVDecompositionGenerator::run_once gadgets/glv.rs:128-142 (curve/glv.rs:39-77): k1[5], k2[5], signs. */
long p2e_glv_decompose_batch(p2e_ctx *ctx, const uint64_t *k, uint64_t *k1, uint64_t *k2, uint64_t *k1_neg,
                             uint64_t *k2_neg, size_t n, size_t ld, uint8_t *err);
/* set_biguint_target gadgets/biguint.rs:454-463 (convert_base :27-51): packed[n][32] -> limbs[9][ld]. */
long p2e_limb_split(p2e_ctx *ctx, const uint8_t *packed, uint64_t *limbs, size_t n, size_t ld);
/* get_biguint_target gadgets/biguint.rs:444-452: limbs[9][ld] -> packed[n][32]. */
long p2e_limb_pack(p2e_ctx *ctx, const uint64_t *limbs, uint8_t *packed, size_t n, size_t ld, uint8_t *err);

/* ---- fused schedules ------------------------------------------------------------------------------ */
/* Every hot-path generator output of verify_secp256k1_message_circuit (gadgets/ecdsa.rs:30-53) for n
 * signatures: cols[P2E_VERIFY_COLS][ld], generator registration order (p2e_schedule_describe).
 * valid[i] (nullable) = 1 iff the three connect constraints hold (curve_assert_valid, GLV
 * reconstruction, r == x), i.e. the signature verifies. */
long p2e_ecdsa_verify_witness_batch(p2e_ctx *ctx, const uint8_t *msg32, const uint8_t *r32, const uint8_t *s32,
                                    const uint8_t *pkx32, const uint8_t *pky32, uint64_t *cols, size_t n,
                                    size_t ld, uint8_t *err, uint8_t *valid);
/* The same circuit evaluated for its verdict only (a pre-filter for invalid signatures before
 * the witness is worth filling; curve/ecdsa.rs:42-62 verify_message on the GPU): valid[i] and err[i] exactly as
 * p2e_ecdsa_verify_witness_batch would set them, no columns written. */
long p2e_ecdsa_verify_batch(p2e_ctx *ctx, const uint8_t *msg32, const uint8_t *r32, const uint8_t *s32,
                            const uint8_t *pkx32, const uint8_t *pky32, size_t n, uint8_t *err, uint8_t *valid);
/* glv_mul(p, k) gadgets/glv.rs:87-104: cols[P2E_GLV_MUL_COLS][ld]. */
long p2e_glv_mul_witness_batch(p2e_ctx *ctx, const uint8_t *px32, const uint8_t *py32, const uint8_t *k32,
                               uint64_t *cols, size_t n, size_t ld, uint8_t *err, uint8_t *valid);

/* ---- built-in-generator columns ------------------------------------------- */
/* The values of the targets that plonky2's OWN generators fill between the hot-path generators of the same
 * circuit, derived on the GPU from the finished witness matrix `cols` (as written by
 * p2e_ecdsa_verify_witness_batch for program 0 / p2e_glv_mul_witness_batch for program 1), the constant tables
 * and the pk.y input, in the order the gadgets create the targets:
 *   split_nonnative_to_4_bit_limbs / _2_bit_limbs  gadgets/split_nonnative.rs:25-72  split_le_base bits of every
 *        limb, then (lower, upper, limb) per 4-bit digit / the limb per 2-bit digit
 *   per fixed-base window  gadgets/curve_fixed_base.rs:56-61  is_equal(limb, zero), not(.), the
 *        random_access_curve_points selection (gadgets/curve_windowed_mul.rs:74-118: 9 x limbs, 9 y limbs)
 *   per MSM digit  gadgets/curve_msm.rs:67-71  index = mul_add(four, limb_m, limb_n), the selection, is_equal, not
 *   per curve_conditional_add  gadgets/curve.rs:232-238  not(b), then the mul_biguint_by_bool products
 *        (gadgets/biguint.rs:360-374) sum.x*b, sum.y*b, p1.x*not_b, p1.y*not_b
 *   per nonnative_conditional_neg  gadgets/nonnative.rs:590-593  not(b), neg*b, x*not_b
 * aux[P2E_VERIFY_AUX_COLS or P2E_GLV_MUL_AUX_COLS][ld_aux], column-major like cols.  err[i] gets
 * P2E_ERR_LIMB_RANGE where a split scalar has a limb >= 2^29 (split_le_base has no witness then).  Wire
 * placement of these targets is not part of this library (plonky2's gate sources are not available offline). */
#define P2E_VERIFY_AUX_COLS 8959
#define P2E_GLV_MUL_AUX_COLS 4738
long p2e_aux_witness_batch(p2e_ctx *ctx, int program, const uint8_t *pky32, const uint64_t *cols, size_t ld,
                           uint64_t *aux, size_t ld_aux, size_t n, uint8_t *err);
/* The same pass inside the compact container: reads only its narrow matrix (every column this pass needs is a limb
 * or a flag), writes aux as u32 aux32[P2E_*_AUX_COLS][ld_aux] (every value is a limb, a bit, a digit or a flag). */
long p2e_aux_witness_compact_batch(p2e_ctx *ctx, int program, const uint8_t *pky32, const uint32_t *narrow,
                                   size_t ld_narrow, uint32_t *aux32, size_t ld_aux, size_t n, uint8_t *err);
typedef struct p2e_aux_desc {
    int32_t kind; /* 0 split4, 1 split2, 2 fixed-base window, 3 MSM digit, 4 conditional neg */
    uint32_t first_col, num_cols;
    char label[48];
} p2e_aux_desc;
long p2e_aux_describe(int program, p2e_aux_desc *out, size_t cap);
long p2e_aux_num_cols(int program);

/* Gate-internal values of the built-in gates on the path, the rest of SURVEY.md 8(f) rank 1: what plonky2's own
 * generators write into wires of their gates beyond the targets the gadgets get back -- per window, in call order,
 *   is_equal(index, zero) (gadgets/curve_fixed_base.rs:57, gadgets/curve_msm.rs:70): not(equal), the
 *        EqualityGenerator's inv = index^-1 in Goldilocks (0 for index 0), diff, diff * inv, diff * (diff * inv)
 *   random_access_curve_points (gadgets/curve_windowed_mul.rs:96-103): per selected limb (9 x, 9 y) the 4 bit
 *        wires of its RandomAccessGate op's access index, least significant first
 * (fixed-base windows: is_equal first; MSM digits: random_access first) = 77 values per window, derived from the aux
 * matrix alone.  gate[P2E_VERIFY_GATE_COLS or P2E_GLV_MUL_GATE_COLS][ld_gate], u64.  [upstream-from-memory: plonky2's
 * is_equal / EqualityGenerator / RandomAccessGenerator are not in the container; "parity unpinned"] */
#define P2E_VERIFY_GATE_COLS 10703
#define P2E_GLV_MUL_GATE_COLS 5621
long p2e_gate_internal_batch(p2e_ctx *ctx, int program, const uint64_t *aux, size_t ld_aux, uint64_t *gate, size_t ld_gate,
                             size_t n);
long p2e_gate_internal_num_cols(int program);

/* ---- constraint-block columns ---------------------------------------------- */
/* The values of the targets the plonky2_ux U29 gates fill INSIDE the constraint blocks of the non-native gadgets
 * (gadgets/nonnative.rs:262-273 add, :330-351 add_many, :373-386 sub, :518-530 inv, :462-463 mul's range check): what
 * add_biguint / sub_biguint / mul_biguint (gadgets/biguint.rs:240-323) receive back from add_many_ux, sub_ux, mul_ux
 * and add_uxs_with_carry (limb, carry / borrow), the mul_biguint_by_bool products modulus * overflow (:360-374), and
 * the cmp_biguint result of a range-checked gadget -- generator by generator, in the order the builder makes the
 * calls (p2e_ux_describe).  Derived on the GPU from the finished matrices `cols` and `aux` (p2e_aux_witness_batch),
 * the packed inputs and the circuit constants through the wiring of p2e_schedule_wiring.
 * ux[P2E_VERIFY_UX_COLS or P2E_GLV_MUL_UX_COLS][ld_ux]: u64 (ux_u32 = 0) or u32 (every value is < 2^29).
 * glv_mul program: k32 travels in msg32, r32 / s32 are ignored (may be NULL).  err[i] gets P2E_ERR_LIMB_RANGE where an
 * operand limb is not a U29 value.  plonky2_ux is not available offline: the gates are modelled by what they constrain
 * (oracle/check_circuit.py holds the same model; "parity unpinned"). */
#define P2E_VERIFY_UX_COLS 249385
#define P2E_GLV_MUL_UX_COLS 194361
long p2e_ux_witness_batch(p2e_ctx *ctx, int program, const uint8_t *msg32, const uint8_t *r32, const uint8_t *s32,
                          const uint8_t *pkx32, const uint8_t *pky32, const uint64_t *cols, size_t ld, const uint64_t *aux,
                          size_t ld_aux, void *ux, int ux_u32, size_t ld_ux, size_t n, uint8_t *err);
typedef struct p2e_ux_desc {
    uint32_t first_col, num_cols; /* this generator's block (num_cols = 0: none, e.g. an unchecked mul) */
} p2e_ux_desc;
/* one entry per generator, same index as p2e_schedule_describe; returns the generator count */
long p2e_ux_describe(int program, p2e_ux_desc *out, size_t cap);
long p2e_ux_num_cols(int program);

/* ---- wire-matrix assembly -------------------------------------------------- */
/* plonky2's prover consumes one matrix per proof, wire_values[num_wires][degree] (PartitionWitness::full_witness,
 * [upstream-from-memory]); a generator's targets are Target::wire(row, column) of the gate rows the builder placed
 * (gates/mul_nonnative.rs:208-226) or virtual targets.  The placement belongs to the circuit build on the Rust side,
 * so it is an INPUT here: a map from library columns to wire-matrix positions, built once per circuit from the
 * targets the gadgets returned.  One entry copies one value; the same source may feed
 * several positions (copy constraints: the x / y wires of a MulNonnativeGate row repeat the operand's limbs).
 *   src = P2E_WIRE_SRC_COLS | c, P2E_WIRE_SRC_AUX | c, P2E_WIRE_SRC_UX | c or P2E_WIRE_SRC_GATE | c   (column c of that matrix)
 *   dst = wire * degree + row        (element index inside one signature's wire matrix)
 * p2e_wire_map_create sorts a copy by dst (coalesced scatter) and keeps it on the device; entries must be in range
 * (dst < num_wires * degree) and no two entries may share a dst.
 * p2e_assemble_wires: wires[i * wire_stride + dst] = value of signature i, for every map entry; positions no entry
 * names are left as they are (zero-fill once, re-use the buffer).  wire_stride >= num_wires * degree elements.
 * A matrix whose entries the map does not use may be NULL.  ux is the u64 or u32 matrix of p2e_ux_witness_batch, gate
 * the matrix of p2e_gate_internal_batch. */
#define P2E_WIRE_SRC_COLS 0x00000000u
#define P2E_WIRE_SRC_AUX 0x40000000u
#define P2E_WIRE_SRC_UX 0x80000000u
#define P2E_WIRE_SRC_GATE 0xC0000000u
typedef struct p2e_wire_map_entry {
    uint32_t src, dst;
} p2e_wire_map_entry;
typedef struct p2e_wire_map p2e_wire_map;
int p2e_wire_map_create(p2e_ctx *ctx, int program, const p2e_wire_map_entry *entries, size_t count, uint32_t num_wires,
                        uint32_t degree, p2e_wire_map **out);
void p2e_wire_map_destroy(p2e_ctx *ctx, p2e_wire_map *map);
long p2e_assemble_wires(p2e_ctx *ctx, const p2e_wire_map *map, const uint64_t *cols, size_t ld, const uint64_t *aux,
                        size_t ld_aux, const void *ux, int ux_u32, size_t ld_ux, const uint64_t *gate, size_t ld_gate,
                        uint64_t *wires, size_t wire_stride, size_t n);

/* ---- layout helper --------------------------------------------------------------------------------- */
/* cols[ncols][ld] (column-major over the batch) -> rows[n][row_ld], one contiguous witness per signature:
 * what a per-signature PartialWitness fill (pw.set_biguint_target ... gadgets/biguint.rs:454-463 per target,
 * INTEGRATION.md section 3) wants to read.  row_ld >= ncols. */
long p2e_columns_to_rows(p2e_ctx *ctx, const uint64_t *cols, size_t ld, size_t n, size_t ncols, uint64_t *rows,
                         size_t row_ld);

/* Compact container for transfers (BASELINE config 5 is bound by the host link, not by the GPU): 57 % of the
 * columns only ever hold values < 2^32 (29-bit limbs, overflow words, flags; set through set_ux_target /
 * set_bool_target in the reference, gadgets/nonnative.rs:643-644,726-727) and are repacked as u32 narrow[num_narrow][ld_narrow];
 * the check_sum[17] and b[16] columns of every mul generator (gates/mul_nonnative.rs:305-322,518-527) stay u64
 * wide[num_wide][ld_wide]: 474 KB instead of 661 KB per verify.  col_map[c] (p2e_compact_layout, host only) is the
 * index of witness column c inside its matrix, with P2E_COMPACT_WIDE set for the wide one.  err[i] gets
 * P2E_ERR_LIMB_RANGE if a narrow column of signature i holds a value >= 2^32 (cannot happen for this library's
 * own output). */
#define P2E_COMPACT_WIDE 0x80000000u
long p2e_compact_layout(int program, uint32_t *col_map, size_t cap, uint32_t *num_narrow, uint32_t *num_wide);
long p2e_columns_compact(p2e_ctx *ctx, int program, const uint64_t *cols, size_t ld, size_t n, uint32_t *narrow,
                         size_t ld_narrow, uint64_t *wide, size_t ld_wide, uint8_t *err);
/* The fused schedules writing the compact container directly (same values, same err / valid semantics as
 * p2e_ecdsa_verify_witness_batch / p2e_glv_mul_witness_batch; 28 % fewer bytes leave the kernels):
 * narrow[num_narrow][ld_narrow] u32, wide[num_wide][ld_wide] u64, ld_narrow >= n, ld_wide >= n. */
long p2e_ecdsa_verify_witness_compact_batch(p2e_ctx *ctx, const uint8_t *msg32, const uint8_t *r32, const uint8_t *s32,
                                            const uint8_t *pkx32, const uint8_t *pky32, uint32_t *narrow,
                                            size_t ld_narrow, uint64_t *wide, size_t ld_wide, size_t n, uint8_t *err,
                                            uint8_t *valid);
long p2e_glv_mul_witness_compact_batch(p2e_ctx *ctx, const uint8_t *px32, const uint8_t *py32, const uint8_t *k32,
                                       uint32_t *narrow, size_t ld_narrow, uint64_t *wide, size_t ld_wide, size_t n,
                                       uint8_t *err, uint8_t *valid);

/* p2e_columns_to_rows for the compact container: one contiguous run of num_narrow u32 and one of num_wide u64 per
 * signature (rows_narrow[n][row_ld_narrow], rows_wide[n][row_ld_wide]). */
long p2e_compact_to_rows(p2e_ctx *ctx, int program, const uint32_t *narrow, size_t ld_narrow, const uint64_t *wide,
                         size_t ld_wide, size_t n, uint32_t *rows_narrow, size_t row_ld_narrow, uint64_t *rows_wide,
                         size_t row_ld_wide);

/* ---- schedule description (column -> generator map, host only, no GPU needed) ---------------------- */
typedef struct p2e_gen_desc {
    int32_t kind;  /* 0 add, 1 sub, 2 add_many, 3 mul(+checksum), 4 inv, 5 glv_decomposition */
    int32_t field; /* P2E_FIELD_* */
    uint32_t first_col, num_cols;
    char label[48]; /* gadget path, e.g. "glv_mul/msm/digit72" */
} p2e_gen_desc;
/* program 0 = verify circuit, 1 = glv_mul.  Writes up to cap entries; returns the total count. */
long p2e_schedule_describe(int program, p2e_gen_desc *out, size_t cap);
long p2e_schedule_num_cols(int program);
/* Operand wiring of generator g (same index as p2e_schedule_describe): where the targets the gadget passed to it live.
 * This is what the reference fixes by handing targets from one CircuitBuilderNonNative call to the next
 * (gadgets/curve.rs:160-243, gadgets/glv.rs:53-104, gadgets/ecdsa.rs:30-53); the Rust side has it as Target lists
 * (dependencies(), gadgets/nonnative.rs:618-624), a batch consumer needs it as columns.  src[k]:
 *   c < 2^29                     first limb column c of the witness matrix (limbs in consecutive columns)
 *   P2E_SRC_AUX | c              column c of the built-in-generator matrix (a mul_biguint_by_bool product, a
 *                                random_access_curve_points selection: p2e_aux_witness_batch)
 *   P2E_SRC_INPUT | slot         a caller input: 0 pk.y, 1 pk.x, 2 msg (glv_mul: k), 3 r, 4 s (9-limb virtual targets)
 *   P2E_SRC_CONST | id           a circuit constant (constant_biguint, gadgets/biguint.rs:165-175): p2e_wiring_const
 * num_limbs[k]: limbs that target has (constants: convert_base's count, zero has none -- quirk Q5; k1, k2: 5).
 * range_check: the gadget's range_check flag (one more cmp_biguint in its constraint block, nonnative.rs:180-190). */
#define P2E_SRC_AUX 0x20000000u
#define P2E_SRC_INPUT 0x40000000u
#define P2E_SRC_CONST 0x80000000u
typedef struct p2e_gen_wiring {
    int32_t num_operands; /* add, sub: 2; add_many: 4; mul: 2; inv: 1; glv: 1 */
    uint32_t src[4];
    uint8_t num_limbs[4];
    int32_t range_check;
} p2e_gen_wiring;
long p2e_schedule_wiring(int program, p2e_gen_wiring *out, size_t cap);
/* value of constant `id` as 32 little-endian bytes; returns its limb count (0 for the zero constant), < 0: unknown id */
int p2e_wiring_const(uint32_t id, uint8_t out32[32]);

/* ---- curve programs --------------------------------------------------------- */
/* The crate's other scalar-multiplication gadgets, on either of its curves, through the same phases as the built-in
 * programs:
 *   P2E_CP_WINDOWED_MUL  curve_scalar_mul_windowed(p, n, true)  gadgets/curve_windowed_mul.rs:131-173   98 185 columns
 *   P2E_CP_SCALAR_MUL    curve_scalar_mul(p, n, true)           gadgets/curve.rs:245-285               139 354 columns
 *   P2E_CP_VERIFY        verify_p256_message_circuit            gadgets/ecdsa.rs:55-78 (P-256 only)    115 557 columns
 * Both gadgets blind with a point drawn by rand() WHILE THE CIRCUIT IS BUILT (precompute_window
 * gadgets/curve_windowed_mul.rs:57, curve_scalar_mul gadgets/curve.rs:253), so their witness depends on the build:
 * a program object stands for one built circuit and takes that point (canonical affine coordinates, 32 little-endian
 * bytes each) as an argument -- the Rust side passes the point its builder drew.  Column order = generator
 * registration order as for the built-in programs (p2e_curve_program_describe / _wiring mirror
 * p2e_schedule_describe / _wiring; constant ids of a program resolve through p2e_curve_program_const: 2c / 2c + 1 =
 * x / y of constant point c, 64 + j = scalar constant j).  The other targets of the same circuits have the same three
 * passes as the built-in programs: p2e_curve_program_aux_witness_batch (built-in-generator targets, layout
 * p2e_curve_program_aux_describe: kinds 0 split4, 2 fixed-base window, 5 window of the windowed multiplication
 * [selected x (9), selected y (9), is_zero, should_add, not_b, sum.x*b, sum.y*b, p1.x*not_b, p1.y*not_b], 6 bit split,
 * 7 bit of curve_scalar_mul [not_bit, sum.x*bit, result.x*not_bit, sum.y*bit, result.y*not_bit]),
 * p2e_curve_program_gate_internal_batch and p2e_curve_program_ux_witness_batch (+ _ux_describe).
 * A program belongs to the context's device; u64 column matrix only for the fill. */
#define P2E_CURVE_SECP256K1 0
#define P2E_CURVE_P256 1
#define P2E_CP_WINDOWED_MUL 1
#define P2E_CP_SCALAR_MUL 2
#define P2E_CP_VERIFY 3
typedef struct p2e_curve_program p2e_curve_program;
int p2e_curve_program_create(p2e_ctx *ctx, int kind, int curve, const uint8_t *blind_x32, const uint8_t *blind_y32,
                             p2e_curve_program **out);
void p2e_curve_program_destroy(p2e_ctx *ctx, p2e_curve_program *prog);
long p2e_curve_program_num_cols(const p2e_curve_program *prog);
long p2e_curve_program_num_aux_cols(const p2e_curve_program *prog);
size_t p2e_curve_program_scratch_bytes(const p2e_curve_program *prog, size_t n);
long p2e_curve_program_describe(const p2e_curve_program *prog, p2e_gen_desc *out, size_t cap);
long p2e_curve_program_wiring(const p2e_curve_program *prog, p2e_gen_wiring *out, size_t cap);
long p2e_curve_program_aux_describe(const p2e_curve_program *prog, p2e_aux_desc *out, size_t cap);
int p2e_curve_program_const(const p2e_curve_program *prog, uint32_t id, uint8_t out32[32]);
long p2e_curve_program_num_gate_cols(const p2e_curve_program *prog);
long p2e_curve_program_num_ux_cols(const p2e_curve_program *prog);
long p2e_curve_program_ux_describe(const p2e_curve_program *prog, p2e_ux_desc *out, size_t cap);
/* inputs as for the program's fill (multiplication programs: the scalar in msg32, r32 = s32 = NULL) */
/* the wire map of p2e_wire_map_create for this program's matrices (then p2e_assemble_wires / p2e_wire_map_destroy) */
int p2e_curve_program_wire_map_create(p2e_ctx *ctx, const p2e_curve_program *prog, const p2e_wire_map_entry *entries,
                                      size_t count, uint32_t num_wires, uint32_t degree, p2e_wire_map **out);
long p2e_curve_program_aux_witness_batch(p2e_ctx *ctx, const p2e_curve_program *prog, const uint8_t *msg32, const uint8_t *r32,
                                         const uint8_t *s32, const uint8_t *pkx32, const uint8_t *pky32, const uint64_t *cols,
                                         size_t ld, uint64_t *aux, size_t ld_aux, size_t n, uint8_t *err);
long p2e_curve_program_gate_internal_batch(p2e_ctx *ctx, const p2e_curve_program *prog, const uint64_t *aux, size_t ld_aux,
                                           uint64_t *gate, size_t ld_gate, size_t n);
long p2e_curve_program_ux_witness_batch(p2e_ctx *ctx, const p2e_curve_program *prog, const uint8_t *msg32, const uint8_t *r32,
                                        const uint8_t *s32, const uint8_t *pkx32, const uint8_t *pky32, const uint64_t *cols,
                                        size_t ld, const uint64_t *aux, size_t ld_aux, void *ux, int ux_u32, size_t ld_ux, size_t n,
                                        uint8_t *err);
/* replaces the run_once bodies of every generator curve_scalar_mul_windowed / curve_scalar_mul registers for a batch
 * of (point, scalar) pairs; cols[num_cols][ld].  valid: always 1 unless flagged (the gadgets connect nothing). */
long p2e_curve_mul_witness_batch(p2e_ctx *ctx, const p2e_curve_program *prog, const uint8_t *px32, const uint8_t *py32,
                                 const uint8_t *k32, uint64_t *cols, size_t n, size_t ld, uint8_t *err, uint8_t *valid);
/* the same for verify_p256_message_circuit; valid = curve_assert_valid's connect and r == x both hold */
long p2e_p256_verify_witness_batch(p2e_ctx *ctx, const p2e_curve_program *prog, const uint8_t *msg32, const uint8_t *r32,
                                   const uint8_t *s32, const uint8_t *pkx32, const uint8_t *pky32, uint64_t *cols, size_t n,
                                   size_t ld, uint8_t *err, uint8_t *valid);
/* Both fills writing the COMPACT container instead (as p2e_ecdsa_verify_witness_compact_batch: u32 narrow[num_narrow]
 * [ld_narrow] for the columns that only ever hold values < 2^32 -- 29-bit limbs, overflow words, flags --, u64
 * wide[num_wide][ld_wide] for the check_sum / carry columns of the mul generators; both indices advance in generator
 * registration order).  p2e_curve_program_compact_layout: col_map[c] = narrow row of witness column c, or
 * P2E_COMPACT_WIDE | wide row; returns num_cols. */
long p2e_curve_mul_witness_compact_batch(p2e_ctx *ctx, const p2e_curve_program *prog, const uint8_t *px32, const uint8_t *py32,
                                         const uint8_t *k32, uint32_t *narrow, size_t ld_narrow, uint64_t *wide, size_t ld_wide,
                                         size_t n, uint8_t *err, uint8_t *valid);
long p2e_p256_verify_witness_compact_batch(p2e_ctx *ctx, const p2e_curve_program *prog, const uint8_t *msg32, const uint8_t *r32,
                                           const uint8_t *s32, const uint8_t *pkx32, const uint8_t *pky32, uint32_t *narrow,
                                           size_t ld_narrow, uint64_t *wide, size_t ld_wide, size_t n, uint8_t *err,
                                           uint8_t *valid);
long p2e_curve_program_compact_layout(const p2e_curve_program *prog, uint32_t *col_map, size_t cap, uint32_t *num_narrow,
                                      uint32_t *num_wide);
/* the P-256 verifier's verdict alone (the counterpart of p2e_ecdsa_verify_batch; native: curve/ecdsa.rs:42-62
 * verify_message with C = P256, with exactly the circuit's verdict): no witness, 2 bytes per signature */
long p2e_p256_verify_batch(p2e_ctx *ctx, const p2e_curve_program *prog, const uint8_t *msg32, const uint8_t *r32,
                           const uint8_t *s32, const uint8_t *pkx32, const uint8_t *pky32, size_t n, uint8_t *err,
                           uint8_t *valid);
/* synthetic valid signatures on a curve (host only; P2E_CURVE_*), same stream layout as p2e_synth_signatures */
int p2e_synth_signatures_curve(int curve, uint64_t seed, size_t first, size_t n, uint8_t *msg32, uint8_t *r32, uint8_t *s32,
                               uint8_t *pkx32, uint8_t *pky32);

/* ---- synthetic inputs (host only): valid signatures per curve/ecdsa.rs:25-40 sign_message with
 * sk, msg, nonce drawn from splitmix64(seed, i).  Host buffers of n*32 bytes each. -------------------- */
int p2e_synth_signatures(uint64_t seed, size_t first, size_t n, uint8_t *msg32, uint8_t *r32, uint8_t *s32,
                         uint8_t *pkx32, uint8_t *pky32);

#ifdef __cplusplus
}
#endif
#endif /* P2E_H */
