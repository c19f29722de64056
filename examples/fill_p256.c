/* Minimal C client of the curve-program part of the C ABI (include/p2e.h): the witness columns of a batch of P-256
 * ECDSA verifications (verify_p256_message_circuit, gadgets/ecdsa.rs:55-78) and the verdict-only pre-filter.
 *     gcc -std=c11 -Iinclude examples/fill_p256.c -Lplonky2-ecdsa_amd -lp2e_hip -o fill_p256
 *     GPU_MAX_HW_QUEUES=8 LD_LIBRARY_PATH=plonky2-ecdsa_amd:/opt/rocm/lib ./fill_p256 512
 * (tests/test_host.py compiles this file to keep it in step with the header.) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "p2e.h"

int main(int argc, char **argv) {
    size_t n = argc > 1 ? (size_t)strtoull(argv[1], NULL, 10) : 128;
    uint8_t *in[5];
    for (int k = 0; k < 5; k++) in[k] = malloc(32 * n);
    if (p2e_synth_signatures_curve(P2E_CURVE_P256, 4, 0, n, in[0], in[1], in[2], in[3], in[4])) return 1;
    in[0][32 * (n / 2)] ^= 1;   /* one tampered message */

    p2e_ctx *ctx = NULL;
    if (p2e_ctx_create(0, P2E_CTX_HOST_POINTERS, NULL, &ctx)) {
        fprintf(stderr, "p2e_ctx_create: %s\n", p2e_last_error());   /* no GPU: there is no CPU fallback */
        return 2;
    }
    /* the point precompute_window drew with rand() when THIS circuit was built (gadgets/curve_windowed_mul.rs:57): the
     * Rust side passes its builder's; any point of the curve serves here -- a synthetic public key */
    uint8_t blind[5][32];
    if (p2e_synth_signatures_curve(P2E_CURVE_P256, 99, 0, 1, blind[0], blind[1], blind[2], blind[3], blind[4])) return 1;
    p2e_curve_program *prog = NULL;
    if (p2e_curve_program_create(ctx, P2E_CP_VERIFY, P2E_CURVE_P256, blind[3], blind[4], &prog)) {
        fprintf(stderr, "p2e_curve_program_create: %s\n", p2e_last_error());
        return 3;
    }
    const size_t ncols = (size_t)p2e_curve_program_num_cols(prog), ld = n;
    uint64_t *cols = malloc(ncols * ld * sizeof *cols);   /* column-major: cols[c * ld + i] */
    uint8_t *err = malloc(n), *valid = malloc(n), *err2 = malloc(n), *valid2 = malloc(n);
    long bad = p2e_p256_verify_witness_batch(ctx, prog, in[0], in[1], in[2], in[3], in[4], cols, n, ld, err, valid);
    long bad2 = p2e_p256_verify_batch(ctx, prog, in[0], in[1], in[2], in[3], in[4], n, err2, valid2);   /* verdict only */
    if (bad < 0 || bad2 < 0) {
        fprintf(stderr, "p2e: %s\n", p2e_last_error());
        return 4;
    }
    size_t verified = 0, agree = 0;
    for (size_t i = 0; i < n; i++) verified += valid[i], agree += valid[i] == valid2[i];
    printf("%zu P-256 fills x %zu columns, %ld flagged, %zu signatures verify, pre-filter agrees on %zu\n", n, ncols, bad, verified, agree);

    long ngen = p2e_curve_program_describe(prog, NULL, 0);
    p2e_gen_desc *gens = malloc((size_t)ngen * sizeof *gens);
    p2e_curve_program_describe(prog, gens, (size_t)ngen);
    printf("%ld generators; first '%s' (kind %d, %u columns), last '%s'\n", ngen, gens[0].label, gens[0].kind, gens[0].num_cols,
           gens[ngen - 1].label);

    p2e_curve_program_destroy(ctx, prog);
    p2e_ctx_destroy(ctx);
    for (int k = 0; k < 5; k++) free(in[k]);
    free(cols), free(err), free(valid), free(err2), free(valid2), free(gens);
    return 0;
}
