/* C restatement of the plonky2-ecdsa hot-path witness generators -- TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this.  The
 * product path (plonky2-ecdsa_amd/, include/p2e.h) never does.
 *
 * PARITY STATUS: "parity unpinned" against literal reference outputs (the reference holds no golden
 * vectors for these generators and cannot be built offline: SURVEY.md 8c).  Pinned instead by
 * (i) agreement with the independent Python big-int restatement oracle/p2e_ref.py on the committed
 * fixtures tests/golden/ (ii) the reference's constraint equations, re-evaluated in tests/.
 *
 * Data layout (same as include/p2e.h): Goldilocks-canonical u64 columns, column-major over the
 * batch: element (col, i) lives at base[col * ld + i], ld >= n.  256-bit inputs are packed 32-byte
 * little-endian, element i at base + 32*i.
 */
#ifndef P2E_ORACLE_H
#define P2E_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P2E_O_FIELD_BASE 0   /* Secp256K1Base   */
#define P2E_O_FIELD_SCALAR 1 /* Secp256K1Scalar */
#define P2E_O_FIELD_P256_BASE 2   /* field/p256_base.rs   */
#define P2E_O_FIELD_P256_SCALAR 3 /* field/p256_scalar.rs */

#define P2E_O_ERR_LIMB_RANGE 1
#define P2E_O_ERR_VALUE_GE_2_256 2
#define P2E_O_ERR_INVERSE_OF_ZERO 4
#define P2E_O_ERR_CARRY_RANGE 8
#define P2E_O_ERR_QUOTIENT_RANGE 16
#define P2E_O_ERR_DIVISION_BY_ZERO 32

#define P2E_O_VERIFY_COLS 82615
#define P2E_O_GLV_MUL_COLS 65243
#define P2E_O_VERIFY_AUX 8959
#define P2E_O_GLV_MUL_AUX 4738

/* each returns the number of elements with err != 0 */
/* gates/mul_nonnative.rs:249-324 + :513-531 : x[9][n], y[9][n] -> r[9], q[9], cs[17], b[16] */
long p2e_oracle_mul_witness(int field, const uint64_t *x, const uint64_t *y, uint64_t *r, uint64_t *q,
                            uint64_t *cs, uint64_t *b, size_t n, size_t ld, uint8_t *err);
/* gates/mul_nonnative.rs:513-531 alone: a[17][n] -> b[16][n] (true Goldilocks field division) */
long p2e_oracle_checksum_witness(const uint64_t *a, uint64_t *b, size_t n, size_t ld, uint8_t *err);
/* gadgets/nonnative.rs:626-645 : a[9][n], b[9][n] -> sum[9][n], overflow[n] */
long p2e_oracle_add_witness(int field, const uint64_t *a, const uint64_t *b, uint64_t *sum, uint64_t *ov,
                            size_t n, size_t ld, uint8_t *err);
/* gadgets/nonnative.rs:792-810 */
long p2e_oracle_sub_witness(int field, const uint64_t *a, const uint64_t *b, uint64_t *diff, uint64_t *ov,
                            size_t n, size_t ld, uint8_t *err);
/* gadgets/nonnative.rs:696-728 : summands[k][9][n] */
long p2e_oracle_add_many_witness(int field, const uint64_t *summands, int k, uint64_t *sum, uint64_t *ov,
                                 size_t n, size_t ld, uint8_t *err);
/* gadgets/nonnative.rs:857-872 : x[9][n] -> inv[9][n], div[9][n] */
long p2e_oracle_inv_witness(int field, const uint64_t *x, uint64_t *inv, uint64_t *div, size_t n, size_t ld,
                            uint8_t *err);
/* gadgets/glv.rs:128-142 : k[9][n] -> k1[5][n], k2[5][n], k1_neg[n], k2_neg[n] */
long p2e_oracle_glv_decompose(const uint64_t *k, uint64_t *k1, uint64_t *k2, uint64_t *k1_neg,
                              uint64_t *k2_neg, size_t n, size_t ld, uint8_t *err);
/* gadgets/biguint.rs:508-518 BigUintDivRemGenerator: a[na][n] (na <= 18), b[nb][n] (nb <= 9) ->
 * div[max(0, na - nb + 1)][n], rem[nb][n].  Limbs must be < 2^29 (ERR_LIMB_RANGE), b != 0 (ERR_DIVISION_BY_ZERO),
 * div must fit its limbs (ERR_LIMB_RANGE: the reference's set_biguint_target assert). */
long p2e_oracle_div_rem(const uint64_t *a, int na, const uint64_t *b, int nb, uint64_t *div, uint64_t *rem, size_t n,
                        size_t ld, uint8_t *err);
/* gadgets/biguint.rs:27-51,454-463 : packed 32-byte LE -> limbs[9][n] and back (pack flags limbs >= 2^29
 * or value >= 2^256) */
long p2e_oracle_limb_split(const uint8_t *packed, uint64_t *limbs, size_t n, size_t ld);
long p2e_oracle_limb_pack(const uint64_t *limbs, uint8_t *packed, size_t n, size_t ld, uint8_t *err);

/* gadgets/ecdsa.rs:30-53 : cols[82615][n].  flags[i] bit0 = all three connect constraints hold
 * (curve_assert_valid, glv k reconstruction, r == x). nthreads <= 0 -> omp default. */
long p2e_oracle_verify_witness(const uint8_t *msg, const uint8_t *r, const uint8_t *s, const uint8_t *pkx,
                               const uint8_t *pky, uint64_t *cols, size_t n, size_t ld, uint8_t *err,
                               uint8_t *flags, int nthreads);
/* The same columns by the OPTIMISED CPU variant (BASELINE.md section 2 variant b; bench.py cpu_baseline_optimised):
 * signatures are walked in lock-step groups of `group` (<= 256, <= 0 -> 64) and every inverse generator's
 * FF::inverse() of a group is one Fermat ladder + 3 multiplications per signature (Montgomery's trick across
 * signatures at the same schedule step); everything else is the faithful-cost algorithm.  Returns -1 if the
 * per-thread coroutine stacks cannot be allocated. */
long p2e_oracle_verify_witness_lockstep(const uint8_t *msg, const uint8_t *r, const uint8_t *s, const uint8_t *pkx,
                                        const uint8_t *pky, uint64_t *cols, size_t n, size_t ld, uint8_t *err,
                                        uint8_t *flags, int nthreads, int group);
/* ... additionally recording aux[8959][n] (see p2e_oracle_verify_witness_aux below); aux == NULL: not recorded */
long p2e_oracle_verify_witness_aux_lockstep(const uint8_t *msg, const uint8_t *r, const uint8_t *s, const uint8_t *pkx,
                                            const uint8_t *pky, uint64_t *cols, size_t n, size_t ld, uint64_t *aux, size_t ald,
                                            uint8_t *err, uint8_t *flags, int nthreads, int group);
/* gadgets/glv.rs:87-104 : cols[65243][n] */
long p2e_oracle_glv_mul_witness(const uint8_t *px, const uint8_t *py, const uint8_t *k, uint64_t *cols,
                                size_t n, size_t ld, uint8_t *err, uint8_t *flags, int nthreads);

/* The same walks, additionally recording aux[8959][n] / aux[4738][n]: the values of the targets that plonky2's
 * BUILT-IN generators fill on this path, in the order the gadgets create them:
 * split_le_base bits and the 4-/2-bit digits built from them (gadgets/split_nonnative.rs:25-72), is_equal / not
 * results and the random_access_curve_points selection per window (gadgets/curve_fixed_base.rs:56-61,
 * gadgets/curve_msm.rs:67-71, gadgets/curve_windowed_mul.rs:74-118), not(b) and the mul_biguint_by_bool products
 * of every conditional add / negation (gadgets/curve.rs:225-243, gadgets/nonnative.rs:584-596,
 * gadgets/biguint.rs:360-374).  aux == NULL: not recorded. */
long p2e_oracle_verify_witness_aux(const uint8_t *msg, const uint8_t *r, const uint8_t *s, const uint8_t *pkx,
                                   const uint8_t *pky, uint64_t *cols, size_t n, size_t ld, uint64_t *aux,
                                   size_t ald, uint8_t *err, uint8_t *flags, int nthreads);
long p2e_oracle_glv_mul_witness_aux(const uint8_t *px, const uint8_t *py, const uint8_t *k, uint64_t *cols,
                                    size_t n, size_t ld, uint64_t *aux, size_t ald, uint8_t *err, uint8_t *flags,
                                    int nthreads);

/* ---- curve programs: the crate's other scalar multiplications, on both curves ------------
 * kind 1: curve_scalar_mul_windowed gadgets/curve_windowed_mul.rs:131-173 (precompute_window :52-72) -- 98 185 columns
 * kind 2: curve_scalar_mul          gadgets/curve.rs:245-285                                         -- 139 354 columns
 * kind 3: verify_p256_message_circuit gadgets/ecdsa.rs:55-78 (curve must be 1)                       -- 115 557 columns
 * curve 0 = secp256k1, 1 = P-256 (curve/p256.rs:15-57).  blind = the point the gadget draws with rand() while the
 * circuit is built (curve_windowed_mul.rs:57, curve.rs:253): an input here, like in include/p2e.h
 * p2e_curve_program_create.  Kinds 1, 2 read (k, -, -, p.x, p.y) from (msg, r, s, px, py); kind 3 reads all five.
 * aux (NULL: not recorded): the built-in-generator values of the same circuit in gadget creation order, as for the
 * built-in programs above (4 221 / 9 918 / 8 442 values).  flags: kinds 1, 2 always 1; kind 3 the two connects
 * (curve_assert_valid, r == x).  lockstep_group > 0: the optimised-CPU variant (one Fermat ladder per group and inverse
 * op), same columns bit for bit.  Returns the number of flagged elements, -1 on bad arguments, -2 if the coroutine
 * stacks cannot be allocated. */
#define P2E_O_CP_WINDOWED_MUL 1
#define P2E_O_CP_SCALAR_MUL 2
#define P2E_O_CP_VERIFY 3
long p2e_oracle_curve_program(int kind, int curve, const uint8_t *blind_x32, const uint8_t *blind_y32,
                              const uint8_t *msg, const uint8_t *r, const uint8_t *s, const uint8_t *px,
                              const uint8_t *py, uint64_t *cols, size_t n, size_t ld, uint64_t *aux, size_t ald,
                              uint8_t *err, uint8_t *flags, int nthreads, int lockstep_group);
/* column counts of a curve program by a dry walk: returns num_cols, *num_aux = built-in-generator values */
long p2e_oracle_curve_program_num_cols(int kind, int curve, long *num_aux);

/* constants computed at init (for tests): rando = keccak256(0u64 LE) as LE scalar * G */
void p2e_oracle_rando(uint8_t x32[32], uint8_t y32[32]);
int p2e_oracle_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
