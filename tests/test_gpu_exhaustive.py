"""GPU (-m gpu): EVERY signature and EVERY column of the BASELINE shapes against the C oracle.

BASELINE.json configs[3] is one batch of 2^16 verifies: on one GPU that is n = 65 536 (the large-batch plan:
lane per signature, run expansion), and sharded over 8 / 4 / 2 GPUs it is n = 8 192 (four lanes per signature),
16 384 and 32 768 (mid-size plan: alternating split inversion batches, 12-iteration runs) per GPU.  Each shape is
filled once through the C ABI into the u64 matrix and once into the compact container, and both are compared with
oracle/p2e_oracle.c's lock-step walk (bit-identical to the faithful walk:
tests/test_oracle.py::test_optimised_cpu_variant_is_bit_identical) on all n x 82 615 elements, in chunks of 4 096
signatures; the comparison itself runs on the GPU (the oracle's chunk is uploaded, nothing of the product's output is
sampled).  The 8 959 built-in-generator columns derived from each container are compared the same way.  Two signatures
per batch are tampered so the verdict column is exercised too.  Nothing here reads
/root/reference.  Gadget: verify_secp256k1_message_circuit, gadgets/ecdsa.rs:30-53."""
import numpy as np
import pytest

import oracle_c

pytestmark = pytest.mark.gpu

CHUNK = 4096


def _first_difference(got, want, base):
    ne = (got != want).nonzero()
    c, i = int(ne[0, 0]), int(ne[0, 1])
    return f"{ne.shape[0]} differing elements; first: column {c}, signature {base + i}: got {int(got[c, i])} want {int(want[c, i])}"


def compare_every_signature(sigs, n, matrices, err, valid, chunk=CHUNK, aux_matrices=()):
    """matrices / aux_matrices: lists of (name, check(chunk_start, chunk_stop, want_int64_cuda) -> None) for the hot-path
    columns / the built-in-generator columns.  Walks the whole batch."""
    import torch
    e, v = err.cpu().numpy(), valid.cpu().numpy()
    checked = 0
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        want, want_aux, werr, wflags = oracle_c.verify_witness_aux_lockstep(*[x[a:b] for x in sigs])
        assert np.array_equal(e[a:b], werr), f"err flags differ in [{a}, {b})"
        assert np.array_equal(v[a:b], wflags), f"verdicts differ in [{a}, {b})"
        want_t = torch.from_numpy(want.view(np.int64)).cuda()
        for _name, check in matrices:
            check(a, b, want_t)
        aux_t = torch.from_numpy(want_aux.view(np.int64)).cuda()
        for _name, check in aux_matrices:
            check(a, b, aux_t)
        checked += (b - a) * want.shape[0]
        del want_t, want, aux_t, want_aux
    return checked


@pytest.mark.parametrize("log2n", [13, 14, 15, 16])
def test_every_signature_and_column_of_the_baseline_shapes(log2n):
    import torch
    import plonky2_ecdsa_amd as p2e
    n = 1 << log2n
    sigs = p2e.synth_signatures(seed=4, n=n)                    # bench.py's batch (seed 4)
    sigs[2][n // 3, 5] ^= 0x40                                  # a tampered s
    sigs[0][n - 1, 0] ^= 1                                      # a tampered message, in the last lane of the batch
    ctx = p2e.Context(device=0)
    dev = [torch.from_numpy(a).cuda() for a in sigs]

    # ---- the u64 matrix (default padded stride, as bench.py fills it)
    cols, err, valid, bad = ctx.ecdsa_verify_witness_batch(*dev)
    torch.cuda.synchronize()
    assert bad == 0 and int(valid.sum()) == n - 2

    # ---- the compact container, written directly by the fused kernels
    cmap, nn, nw = p2e.compact_layout(p2e.PROGRAM_VERIFY)
    nar, wid, cerr, cvalid, cbad = ctx.ecdsa_verify_witness_compact_batch(*dev)
    torch.cuda.synchronize()
    assert cbad == 0 and torch.equal(cvalid, valid) and torch.equal(cerr, err)
    is_wide = (cmap & p2e.COMPACT_WIDE) != 0
    narrow_cols = torch.from_numpy(np.nonzero(~is_wide)[0]).cuda()       # witness column of narrow row k: layout is in
    wide_cols = torch.from_numpy(np.nonzero(is_wide)[0]).cuda()          # registration order, so the k-th narrow column
    assert np.array_equal(cmap[~is_wide], np.arange(nn)) and np.array_equal(cmap[is_wide] & 0x7FFFFFFF, np.arange(nw))

    def check_u64(a, b, want):
        got = cols[:, a:b]
        assert torch.equal(got, want), "u64 matrix: " + _first_difference(got, want, a)

    def check_compact(a, b, want):
        got_n = nar[:, a:b].to(torch.int64) & 0xFFFFFFFF
        want_n = want[narrow_cols]
        assert torch.equal(got_n, want_n), "compact container, narrow: " + _first_difference(got_n, want_n, a)
        got_w, want_w = wid[:, a:b], want[wide_cols]
        assert torch.equal(got_w, want_w), "compact container, wide: " + _first_difference(got_w, want_w, a)

    # ---- the built-in-generator columns derived from both containers (p2e_aux_witness_batch / _compact_batch)
    aux, aerr, abad = ctx.aux_witness_batch(p2e.PROGRAM_VERIFY, dev[4], cols, n=n, ld=cols.stride(0))
    aux32, a32err, a32bad = ctx.aux_witness_compact_batch(p2e.PROGRAM_VERIFY, dev[4], nar, n=n, ld_narrow=nar.stride(0))
    torch.cuda.synchronize()
    assert abad == 0 and a32bad == 0

    def check_aux(a, b, want_aux):
        got = aux[:, a:b]
        assert torch.equal(got, want_aux), "aux matrix: " + _first_difference(got, want_aux, a)

    def check_aux32(a, b, want_aux):
        got = aux32[:, a:b].to(torch.int64) & 0xFFFFFFFF
        assert torch.equal(got, want_aux), "aux matrix of the compact container: " + _first_difference(got, want_aux, a)

    checked = compare_every_signature(sigs, n, [("u64", check_u64), ("compact", check_compact)], err, valid,
                                      aux_matrices=[("aux", check_aux), ("aux32", check_aux32)])
    assert checked == n * p2e.VERIFY_COLS
