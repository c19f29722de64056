/* Minimal C client of the C ABI (include/p2e.h): fill the witness columns of a batch of ECDSA verifications from
 * host memory, then read one generator's outputs through the column map.  Plain C11, no HIP headers needed:
 *     gcc -std=c11 -Iinclude examples/fill_batch.c -Lplonky2-ecdsa_amd -lp2e_hip -o fill_batch
 *     GPU_MAX_HW_QUEUES=8 LD_LIBRARY_PATH=plonky2-ecdsa_amd:/opt/rocm/lib ./fill_batch 1024
 * (tests/test_host.py compiles this file to keep it in step with the header.) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "p2e.h"

int main(int argc, char **argv) {
    size_t n = argc > 1 ? (size_t)strtoull(argv[1], NULL, 10) : 256;
    /* synthetic valid signatures (curve/ecdsa.rs:25-40 sign_message with splitmix64 keys), packed 32-byte LE */
    uint8_t *in[5];
    for (int k = 0; k < 5; k++) in[k] = malloc(32 * n);
    if (p2e_synth_signatures(4, 0, n, in[0], in[1], in[2], in[3], in[4])) return 1;

    p2e_ctx *ctx = NULL;
    if (p2e_ctx_create(0, P2E_CTX_HOST_POINTERS, NULL, &ctx)) {
        fprintf(stderr, "p2e_ctx_create: %s\n", p2e_last_error());   /* no GPU: there is no CPU fallback */
        return 2;
    }
    const size_t ncols = (size_t)p2e_schedule_num_cols(0), ld = n;
    uint64_t *cols = malloc(ncols * ld * sizeof *cols);   /* column-major: cols[c * ld + i] */
    uint8_t *err = malloc(n), *valid = malloc(n);
    long bad = p2e_ecdsa_verify_witness_batch(ctx, in[0], in[1], in[2], in[3], in[4], cols, n, ld, err, valid);
    if (bad < 0) {
        fprintf(stderr, "p2e_ecdsa_verify_witness_batch: %s\n", p2e_last_error());
        return 3;
    }
    size_t verified = 0;
    for (size_t i = 0; i < n; i++) verified += valid[i];
    printf("%zu fills x %zu columns, %ld flagged, %zu signatures verify\n", n, ncols, bad, verified);

    /* the column map: generator g of the circuit wrote columns [first_col, first_col + num_cols) */
    long ngen = p2e_schedule_describe(0, NULL, 0);
    p2e_gen_desc *gens = malloc((size_t)ngen * sizeof *gens);
    p2e_schedule_describe(0, gens, (size_t)ngen);
    const p2e_gen_desc *last = &gens[ngen - 1];   /* final curve_add's y3 = sub_nonnative(...): 9 limbs + overflow */
    printf("last generator '%s' kind %d: signature 0 limbs", last->label, last->kind);
    for (uint32_t k = 0; k < last->num_cols; k++) printf(" %llu", (unsigned long long)cols[(last->first_col + k) * ld]);
    printf("\n");

    p2e_ctx_destroy(ctx);
    for (int k = 0; k < 5; k++) free(in[k]);
    free(cols), free(err), free(valid), free(gens);
    return 0;
}
