"""GPU (-m gpu): the HIP kernels through the C ABI against the oracle and the golden fixtures.
Bit-exact (integer / Goldilocks work).  Nothing here reads /root/reference."""
import os

import numpy as np
import pytest

import oracle_c
import p2e_ref as R
import parity_checks as pc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    from backends import GpuBackend
    return GpuBackend()


@pytest.fixture(scope="module")
def ora():
    from backends import OracleBackend
    return OracleBackend()


@pytest.mark.parametrize("check", pc.ALL_PRIM_CHECKS, ids=lambda f: f.__name__)
def test_generators_match_golden(gpu, check):
    check(gpu)


def test_verify_matches_golden(gpu):
    pc.check_verify_golden(gpu)


def test_glv_mul_matches_golden(gpu):
    pc.check_glv_mul_golden(gpu)


def _rand_limbs(rng, n, bound):
    vals = [int(rng.integers(0, 2**63)) ** 5 % bound for _ in range(n)]
    return oracle_c.limbs_cols(vals)


@pytest.mark.parametrize("field", [0, 1])
def test_cfg2_mul_batch_4096(gpu, ora, field):
    """BASELINE config 2: 2^12 NonNativeMultiplication witnesses."""
    import plonky2_ecdsa_amd as p2e
    m = R.MODULI[field]
    rng = R.SplitMix64(2 + field)
    x = oracle_c.limbs_cols([rng.below(m) for _ in range(4096)])
    y = oracle_c.limbs_cols([rng.below(m) for _ in range(4096)])
    want = ora.mul(field, x, y)
    got = gpu.mul(field, x, y)
    for g, w in zip(got, want):
        assert np.array_equal(np.asarray(g), w)
    assert not np.asarray(got[4]).any()


@pytest.mark.parametrize("field", [0, 1])
def test_inv_add_sub_random(gpu, ora, field):
    m = R.MODULI[field]
    rng = R.SplitMix64(40 + field)
    a = oracle_c.limbs_cols([rng.below(m) for _ in range(777)])
    b = oracle_c.limbs_cols([rng.below(m) for _ in range(777)])
    for name in ("add", "sub"):
        for g, w in zip(getattr(gpu, name)(field, a, b), getattr(ora, name)(field, a, b)):
            assert np.array_equal(np.asarray(g), w)
    for g, w in zip(gpu.inv(field, a), ora.inv(field, a)):
        assert np.array_equal(np.asarray(g), w)
    k = oracle_c.limbs_cols([rng.below(R.N) for _ in range(777)])
    for g, w in zip(gpu.glv(k), ora.glv(k)):
        assert np.array_equal(np.asarray(g), w)


def test_structured_operands(gpu, ora):
    """Same structured-operand sweep as the CPU emulation, on the real kernels (asm multiply / squaring)."""
    pc.check_structured_mul_add_sub(gpu, ora, count=20000)


@pytest.mark.parametrize("plan", ["quad", "lane_per_signature", "quad_split8", "quad_op_by_op"])
def test_verify_random_batch_ragged(ora, monkeypatch, plan):
    """n = 301: not a multiple of the wavefront / workgroup size (nor of the 64 signatures a quad workgroup holds).
    Both forms of phases A / B: four lanes per signature with split inversion batches (the default below 24 576
    signatures per call, csrc/quad.hpp) and one lane per signature (what a 2^16 batch runs)."""
    import plonky2_ecdsa_amd as p2e
    from backends import GpuBackend
    monkeypatch.setenv("P2E_QUAD_MAX_N", "0" if plan == "lane_per_signature" else "1000000")
    if plan == "quad_split8":
        monkeypatch.setenv("P2E_BINV_SPLIT_LOG2", "3")
    if plan == "quad_op_by_op":                    # round-2 form of the four-lane plan: every op expanded on its own
        monkeypatch.setenv("P2E_RUN_ITERS_SMALL", "0")
    gpu = GpuBackend()
    arrs = p2e.synth_signatures(seed=77, n=301)
    want, werr, wflags = ora.verify(*arrs)
    got, err, valid = gpu.verify(*arrs)
    assert not np.asarray(err).any() and np.asarray(valid).all() and wflags.all()
    assert np.array_equal(np.asarray(got).view(np.uint64), want)


@pytest.mark.parametrize("n,ld_pad", [(1000, 0), (1000, 1), (768, 6)])
def test_verify_wide_and_tail_launches(ora, n, ld_pad):
    """n = 1000: three full workgroups take the 16-byte paired-store kernels, the 232-signature tail the
    8-byte ones; an odd column stride (ld_pad = 1) must fall back to 8-byte stores everywhere."""
    import torch
    import plonky2_ecdsa_amd as p2e
    sigs = p2e.synth_signatures(seed=123, n=n)
    want, _, _ = ora.verify(*sigs)
    ctx = p2e.Context(device=0)
    dev = [torch.from_numpy(a).cuda() for a in sigs]
    ld = n + ld_pad
    big = torch.zeros((p2e.VERIFY_COLS, ld), dtype=torch.int64, device="cuda")
    cols, err, valid, bad = ctx.ecdsa_verify_witness_batch(*dev, cols=big[:, :n], ld=ld)
    torch.cuda.synchronize()
    assert bad == 0 and bool(valid.cpu().numpy().all())
    host = big.cpu().numpy().view(np.uint64)
    assert np.array_equal(host[:, :n], want)
    assert not host[:, n:].any()


@pytest.mark.parametrize("run_iters", [1, 4, 9, 73])
def test_verify_run_expansion_any_run_length(ora, monkeypatch, run_iters):
    """Below 49 152 signatures the library expands op by op; force the run-walking kernels (k_expand_runs, what a
    2^16 batch uses) on a small ragged batch, at several run lengths, for both programs."""
    import plonky2_ecdsa_amd as p2e
    monkeypatch.setenv("P2E_RUNS_MIN_N", "0")
    monkeypatch.setenv("P2E_RUN_ITERS", str(run_iters))
    # ... and, for two of the run lengths, the fixed-base windows as one run per signature (k_expand_fb_run: off by
    # default in this program)
    monkeypatch.setenv("P2E_FB_RUN", "1" if run_iters in (4, 73) else "0")
    n = 700                                             # two full workgroups (paired stores) + a 188-signature tail
    sigs = p2e.synth_signatures(seed=31 + run_iters, n=n)
    ctx = p2e.Context(device=0, host_pointers=True)
    want, _, wflags = ora.verify(*sigs)
    got, err, valid = ctx.ecdsa_verify_witness_batch(*sigs)[:3]
    assert not np.asarray(err).any() and np.asarray(valid).all() and wflags.all()
    assert np.array_equal(np.asarray(got).view(np.uint64), want)
    rng = R.SplitMix64(900 + run_iters)
    k = oracle_c.pack256([rng.below(R.N) for _ in range(n)])
    want, _, _ = ora.glv_mul(sigs[3], sigs[4], k)
    got, err, valid = ctx.glv_mul_witness_batch(sigs[3], sigs[4], k)[:3]
    assert not np.asarray(err).any()
    assert np.array_equal(np.asarray(got).view(np.uint64), want)


def test_aux_matches_golden(gpu):
    pc.check_aux_golden(gpu)


@pytest.mark.parametrize("program", [0, 1])
def test_aux_random_batch_ragged(gpu, ora, program):
    """Built-in-generator columns (bool selects, bits / digits, random-access selections) derived on the GPU from its
    own witness matrix, against the oracle's independent walk; n = 301 is not a multiple of the workgroup size."""
    import plonky2_ecdsa_amd as p2e
    sigs = p2e.synth_signatures(seed=55, n=301)
    if program == 0:
        inputs = sigs
    else:
        rng = R.SplitMix64(56)
        inputs = [sigs[3], sigs[4], oracle_c.pack256([rng.below(R.N) for _ in range(301)])]
    wcols, want, werr = ora.aux(program, inputs)
    cols, got, err = gpu.aux(program, inputs)
    assert not werr.any() and not err.any()
    assert np.array_equal(cols, wcols) and np.array_equal(got, want)


def test_cfg3_glv_mul_1024(gpu, ora):
    """BASELINE config 3: 2^10 glv_mul witness fills."""
    import plonky2_ecdsa_amd as p2e
    sig = p2e.synth_signatures(seed=3, n=1024)
    rng = R.SplitMix64(303)
    k = oracle_c.pack256([rng.below(R.N) for _ in range(1024)])
    want, werr, wflags = ora.glv_mul(sig[3], sig[4], k)
    got, err, valid = gpu.glv_mul(sig[3], sig[4], k)
    assert not np.asarray(err).any() and np.asarray(valid).all()
    assert np.array_equal(np.asarray(got).view(np.uint64), want)


@pytest.mark.parametrize("runs", [False, True], ids=["op_by_op", "run_expansion"])
def test_edge_inputs_and_error_flags(gpu, ora, monkeypatch, runs):
    if runs:   # the expansion a 2^16 batch uses (k_expand_runs), forced on this small batch
        import plonky2_ecdsa_amd as p2e
        monkeypatch.setenv("P2E_RUNS_MIN_N", "0")

        class _G:
            ctx = p2e.Context(device=0, host_pointers=True)
            verify = lambda self, *a: self.ctx.ecdsa_verify_witness_batch(*[np.ascontiguousarray(x, np.uint8) for x in a])[:3]
        gpu = _G()
    else:
        monkeypatch.setenv("P2E_RUN_ITERS_SMALL", "0")
    sig = list(R.synth_signature_at(5, 0))
    rx, ry = R.rando_point()
    cases = [tuple(sig)]
    bad = list(sig)
    bad[1] = (bad[1] + 1) % R.N
    cases.append(tuple(bad))                                      # does not verify
    cases.append((sig[0], sig[1], 1, sig[3], sig[4]))             # s = 1
    cases.append((sig[0], sig[1], sig[2], R.GX, R.GY))            # pk = G
    cases.append((sig[0], sig[1], sig[2], rx, (-ry) % R.P))       # first table add hits x2 == x1 -> inverse of zero
    cases.append((sig[0], sig[1], 0, sig[3], sig[4]))             # s = 0 -> inverse of zero (mod n)
    cases.append((2**256 - 1, 2**256 - 1, 2**256 - 1, 2**256 - 1, 2**256 - 1))  # non-canonical everything
    cases.append(R.synth_signature_at(5, 1))
    arrs = [oracle_c.pack256([c[k] for c in cases]) for k in range(5)]
    want, werr, wflags = ora.verify(*arrs)
    got, err, valid = gpu.verify(*arrs)
    err, valid = np.asarray(err), np.asarray(valid)
    assert np.array_equal(err != 0, werr != 0), (err, werr)
    ok = werr == 0
    assert np.array_equal(np.asarray(got).view(np.uint64)[:, ok], want[:, ok])
    assert np.array_equal(valid[ok], wflags[ok]) and not valid[~ok].any()
    assert err[4] & R.ERR_INVERSE_OF_ZERO and err[5] & R.ERR_INVERSE_OF_ZERO


def test_device_pointers_ld_and_async():
    """Device-pointer mode on torch's stream, output written into a column SLICE of a larger matrix
    (ld > n: how a rank fills its part of an assembled matrix), asynchronous context."""
    import torch
    import plonky2_ecdsa_amd as p2e
    n, ld, off = 200, 512, 100
    sigs = p2e.synth_signatures(seed=8, n=n)
    want, _, _ = oracle_c.verify_witness(*sigs)
    ctx = p2e.Context(device=0, asynchronous=True)
    dev = [torch.from_numpy(a).cuda() for a in sigs]
    big = torch.full((p2e.VERIFY_COLS, ld), -1, dtype=torch.int64, device="cuda")
    view = big[:, off:off + n]
    err = torch.empty(n, dtype=torch.uint8, device="cuda")
    valid = torch.empty(n, dtype=torch.uint8, device="cuda")
    rc = ctx._L.p2e_ecdsa_verify_witness_batch(ctx._h, *[p2e._ptr(d) for d in dev],
                                               p2e.C.c_void_p(view.data_ptr()), p2e.C.c_size_t(n), p2e.C.c_size_t(ld),
                                               p2e._ptr(err), p2e._ptr(valid))
    assert rc == 0
    assert ctx.sync() == 0
    host = big.cpu().numpy().view(np.uint64)
    assert np.array_equal(host[:, off:off + n], want)
    assert (host[:, :off] == np.uint64(2**64 - 1)).all() and (host[:, off + n:] == np.uint64(2**64 - 1)).all()
    assert bool(valid.cpu().numpy().all()) and not err.cpu().numpy().any()


def test_columns_to_rows(gpu):
    """Layout helper: column-major witness -> one contiguous row per signature (ragged sizes, padded strides)."""
    import torch
    import plonky2_ecdsa_amd as p2e
    rng = np.random.default_rng(9)
    for n, ncols, ld in ((1, 1, 1), (70, 131, 70), (301, 517, 320), (64, 64, 64)):
        cols = rng.integers(0, 2**63, size=(ncols, ld), dtype=np.int64).astype(np.uint64)
        rows = gpu.ctx.columns_to_rows(cols, n=n, ld=ld)
        assert np.array_equal(np.asarray(rows), cols[:, :n].T)
    # device pointers, on the real witness
    sigs = p2e.synth_signatures(seed=55, n=130)
    ctx = p2e.Context(device=0)
    cols, err, valid, bad = ctx.ecdsa_verify_witness_batch(*[torch.from_numpy(a).cuda() for a in sigs])
    rows = ctx.columns_to_rows(cols)
    torch.cuda.synchronize()
    assert torch.equal(rows, cols.t())


@pytest.mark.parametrize("n,pad", [(600, 0), (601, 0), (512, 1)])
def test_columns_compact_round_trip(n, pad):
    """Compact transfer container (u32 narrow + u64 wide matrices): expanding it on the host gives the witness matrix
    back; odd batch sizes / odd strides take the single-element path; a value that does not fit is flagged."""
    import torch
    import plonky2_ecdsa_amd as p2e
    sigs = p2e.synth_signatures(seed=61, n=n)
    ctx = p2e.Context(device=0)
    dev = [torch.from_numpy(a).cuda() for a in sigs]
    ld = n + pad
    big = torch.zeros((p2e.VERIFY_COLS, ld), dtype=torch.int64, device="cuda")
    cols, err, valid, bad = ctx.ecdsa_verify_witness_batch(*dev, cols=big[:, :n], ld=ld)
    assert bad == 0
    _m, nn, nw = p2e.compact_layout(0)
    nar = torch.full((nn, n + 2 * pad), -1, dtype=torch.int32, device="cuda")
    wid = torch.full((nw, n + 4 * pad), -1, dtype=torch.int64, device="cuda")
    _, _, cerr, cbad = ctx.columns_compact(0, big, n=n, ld=ld, narrow=nar, wide=wid, ld_narrow=n + 2 * pad, ld_wide=n + 4 * pad)
    torch.cuda.synchronize()
    assert cbad == 0 and int(cerr.sum()) == 0
    back = p2e.compact_expand(0, nar[:, :n].cpu().numpy().view(np.uint32), wid[:, :n].cpu().numpy())
    assert np.array_equal(back, big[:, :n].cpu().numpy().view(np.uint64))
    assert bool((nar[:, n:] == -1).all()) and bool((wid[:, n:] == -1).all())       # padding untouched
    big[7, 5] = 1 << 32                                                             # r limb of the first mul: narrow
    _, _, cerr, cbad = ctx.columns_compact(0, big, n=n, ld=ld, narrow=nar, wide=wid, ld_narrow=n + 2 * pad, ld_wide=n + 4 * pad)
    flagged = np.nonzero(cerr.cpu().numpy())[0].tolist()
    assert cbad == 1 and flagged == [5]


@pytest.mark.parametrize("runs", [False, True], ids=["op_by_op", "run_expansion"])
def test_fused_compact_output_matches_oracle(ora, monkeypatch, runs):
    """The fused schedules writing the compact container directly (u32 narrow + u64 wide matrices): expanded on the
    host it is the oracle's matrix.  n = 700: two full workgroups (paired 8- / 16-byte stores) + a ragged tail; padded
    strides; both expansion kernels; both programs."""
    import torch
    import plonky2_ecdsa_amd as p2e
    monkeypatch.setenv("P2E_RUNS_MIN_N" if runs else "P2E_RUN_ITERS_SMALL", "0")
    n = 700
    sigs = p2e.synth_signatures(seed=71, n=n)
    ctx = p2e.Context(device=0)
    dev = [torch.from_numpy(a).cuda() for a in sigs]
    _m, nn, nw = p2e.compact_layout(0)
    nar = torch.full((nn, n + 2), -1, dtype=torch.int32, device="cuda")
    wid = torch.full((nw, n + 4), -1, dtype=torch.int64, device="cuda")
    _, _, err, valid, bad = ctx.ecdsa_verify_witness_compact_batch(*dev, narrow=nar, wide=wid, ld_narrow=n + 2, ld_wide=n + 4)
    torch.cuda.synchronize()
    assert bad == 0 and int(valid.sum()) == n
    want, _, _ = ora.verify(*sigs)
    got = p2e.compact_expand(0, nar[:, :n].cpu().numpy().view(np.uint32), wid[:, :n].cpu().numpy())
    assert np.array_equal(got, want)
    assert bool((nar[:, n:] == -1).all()) and bool((wid[:, n:] == -1).all())
    # the built-in-generator pass inside the container: narrow matrix in, u32 matrix out
    aux32 = torch.full((p2e.VERIFY_AUX_COLS, n + 6), -1, dtype=torch.int32, device="cuda")
    _, aerr, abad = ctx.aux_witness_compact_batch(0, dev[4], nar, n=n, ld_narrow=n + 2, aux32=aux32, ld_aux=n + 6)
    torch.cuda.synchronize()
    assert abad == 0 and bool((aux32[:, n:] == -1).all())
    assert np.array_equal(aux32[:, :n].cpu().numpy().view(np.uint32).astype(np.uint64), ora.aux(0, sigs)[1])
    # one contiguous compact witness per signature
    rn, rw = ctx.compact_to_rows(0, nar, wid, n, ld_narrow=n + 2, ld_wide=n + 4)
    assert torch.equal(rn, nar[:, :n].t()) and torch.equal(rw, wid[:, :n].t())
    # and it is the same container p2e_columns_compact makes from the standard matrix
    cols, _, _, _ = ctx.ecdsa_verify_witness_batch(*dev)
    nar2, wid2, _, _ = ctx.columns_compact(0, cols, n=n, ld=cols.stride(0))
    assert torch.equal(nar2, nar[:, :n]) and torch.equal(wid2, wid[:, :n])
    # glv_mul program, host pointers (odd stride n = 700 is even; use the staged path)
    hctx = p2e.Context(device=0, host_pointers=True)
    rng = R.SplitMix64(72)
    k = oracle_c.pack256([rng.below(R.N) for _ in range(n)])
    gn, gw, gerr, _gvalid, gbad = hctx.glv_mul_witness_compact_batch(sigs[3], sigs[4], k)
    assert gbad == 0
    assert np.array_equal(p2e.compact_expand(1, gn, gw), ora.glv_mul(sigs[3], sigs[4], k)[0])


def test_verdict_only_matches_full_fill(ora):
    """p2e_ecdsa_verify_batch (pre-filter: no columns written) sets valid / err like the witness fill, on a ragged batch
    with tampered signatures and an inverse-of-zero element."""
    import plonky2_ecdsa_amd as p2e
    n = 333
    arrs = [a.copy() for a in p2e.synth_signatures(seed=91, n=n)]
    for i in range(0, n, 5):
        arrs[i % 3][i, 0] ^= 1                                     # tamper msg / r / s
    rx, ry = R.rando_point()
    arrs[3][17] = np.frombuffer(int(rx).to_bytes(32, "little"), dtype=np.uint8)
    arrs[4][17] = np.frombuffer(int((-ry) % R.P).to_bytes(32, "little"), dtype=np.uint8)
    ctx = p2e.Context(device=0, host_pointers=True)
    _cols, ferr, fvalid, _ = ctx.ecdsa_verify_witness_batch(*arrs)
    verr, vvalid, vbad = ctx.ecdsa_verify_batch(*arrs)
    _want, werr, wflags = ora.verify(*arrs)
    assert np.array_equal(verr != 0, werr != 0) and np.array_equal(verr != 0, ferr != 0) and vbad == int((verr != 0).sum())
    assert np.array_equal(vvalid, fvalid) and np.array_equal(vvalid[werr == 0], wflags[werr == 0])
    assert verr[17] & R.ERR_INVERSE_OF_ZERO and 0 < int(vvalid.sum()) < n


@pytest.mark.parametrize("n", [1, 2, 63, 65, 257])
def test_tiny_and_odd_batches_every_output(n):
    """Batch sizes below / around a wavefront and a workgroup: u64 matrix (device and staged host pointers), compact
    container, built-in-generator columns and the verdict-only call against the oracle."""
    import torch
    import plonky2_ecdsa_amd as p2e
    sigs = p2e.synth_signatures(seed=1000 + n, n=n)
    want, waux, _werr, _wflags = oracle_c.verify_witness_aux(*sigs)
    ctx = p2e.Context(device=0)
    dev = [torch.from_numpy(a).cuda() for a in sigs]
    cols, _err, valid, bad = ctx.ecdsa_verify_witness_batch(*dev)
    aux, _aerr, abad = ctx.aux_witness_batch(0, dev[4], cols, n=n, ld=cols.stride(0))
    nar, wid, _cerr, cvalid, cbad = ctx.ecdsa_verify_witness_compact_batch(*dev)
    _verr, vvalid, vbad = ctx.ecdsa_verify_batch(*dev)
    torch.cuda.synchronize()
    assert bad + abad + cbad + vbad == 0 and int(valid.sum()) == n == int(vvalid.sum()) == int(cvalid.sum())
    assert np.array_equal(cols.cpu().numpy().view(np.uint64), want)
    assert np.array_equal(aux.cpu().numpy().view(np.uint64), waux)
    assert np.array_equal(p2e.compact_expand(0, nar.cpu().numpy().view(np.uint32), wid.cpu().numpy()), want)
    hcols = p2e.Context(device=0, host_pointers=True).ecdsa_verify_witness_batch(*sigs)[0]
    assert np.array_equal(np.asarray(hcols).view(np.uint64), want)


def test_plain_c_client(tmp_path):
    """examples/fill_batch.c: a C11 program on the C ABI alone (no torch, no Python in the process)."""
    import subprocess
    from test_host import _build_c_example
    exe, env = _build_c_example(tmp_path)
    r = subprocess.run([exe, "300"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "300 fills x 82615 columns, 0 flagged, 300 signatures verify" in r.stdout


def test_plain_c_client_of_the_p256_verifier(tmp_path):
    """examples/fill_p256.c: curve program create / fill / verdict-only pre-filter / describe from plain C"""
    import subprocess
    from test_host import _build_c_example
    exe, env = _build_c_example(tmp_path, "fill_p256")
    r = subprocess.run([exe, "300"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "300 P-256 fills x 115557 columns, 0 flagged, 299 signatures verify, pre-filter agrees on 300" in r.stdout
    assert "4828 generators" in r.stdout


def test_api_misuse_returns_status_not_crash():
    import plonky2_ecdsa_amd as p2e
    ctx = p2e.Context(device=0, host_pointers=True)
    x = np.zeros((9, 4), dtype=np.uint64)
    rc = ctx._L.p2e_mul_witness_batch(ctx._h, p2e.C.c_int(7), p2e._ptr(x), p2e._ptr(x), p2e._ptr(x), p2e._ptr(x),
                                      p2e._ptr(x), p2e._ptr(x), p2e.C.c_size_t(4), p2e.C.c_size_t(4), p2e._ptr(x))
    assert rc == -1
    rc = ctx._L.p2e_add_witness_batch(ctx._h, p2e.C.c_int(0), p2e._ptr(x), p2e._ptr(x), p2e._ptr(x), p2e._ptr(x),
                                      p2e.C.c_size_t(8), p2e.C.c_size_t(4), p2e._ptr(x))   # ld < n
    assert rc == -1
    L, c = ctx._L, p2e.C
    py, e = np.zeros((4, 32), dtype=np.uint8), np.zeros(4, dtype=np.uint8)
    cols, aux = np.zeros((p2e.VERIFY_COLS, 4), dtype=np.uint64), np.zeros((p2e.VERIFY_AUX_COLS, 4), dtype=np.uint64)
    ok_args = lambda prog=0, ld_aux=4, a=aux: (ctx._h, c.c_int(prog), p2e._ptr(py), p2e._ptr(cols), c.c_size_t(4), p2e._ptr(a),
                                               c.c_size_t(ld_aux), c.c_size_t(4), p2e._ptr(e))
    assert L.p2e_aux_witness_batch(*ok_args(prog=2)) == -1            # unknown program
    assert L.p2e_aux_witness_batch(*ok_args(ld_aux=3)) == -1          # ld_aux < n
    assert L.p2e_aux_witness_batch(*ok_args(a=None)) == -1            # null output
    assert L.p2e_aux_witness_batch(*ok_args()) >= 0                   # an all-zero matrix is a valid (if meaningless) input


def test_mid_size_batch_between_the_two_plans():
    """24 576 < n < 49 152 (the N = 2 point of the strong-scaled shape): lane-per-signature chains, but the inversion
    batches of consecutive pieces alternate between two streams and are cut into two sub-ranges each (k_batch_inv_split
    behind k_chains).  33 000 + 77 signatures (ragged), both programs: flags, and 64 sampled signatures on every column."""
    import torch
    import plonky2_ecdsa_amd as p2e
    n = 33000 + 77
    sigs = p2e.synth_signatures(seed=15, n=n)
    sigs[2][1234, 0] ^= 1                                       # one tampered s
    ctx = p2e.Context(device=0)
    dev = [torch.from_numpy(a).cuda() for a in sigs]
    cols, err, valid, bad = ctx.ecdsa_verify_witness_batch(*dev)
    torch.cuda.synchronize()
    v = valid.cpu().numpy()
    assert bad == 0 and int(err.sum()) == 0 and v.sum() == n - 1 and v[1234] == 0
    sample = np.unique(np.concatenate([np.linspace(0, n - 1, 62).astype(np.int64), [1234, n - 1]]))
    want, werr, wflags = oracle_c.verify_witness(*[a[sample] for a in sigs])
    got = cols[:, torch.from_numpy(sample).cuda()].cpu().numpy().view(np.uint64)
    assert np.array_equal(got, want) and np.array_equal(v[sample], wflags)
    rng = R.SplitMix64(4242)
    k = oracle_c.pack256([rng.below(R.N) for _ in range(len(sample))])
    kk = np.zeros((n, 32), np.uint8)
    kk[:] = k[0]
    kk[sample] = k
    gcols, gerr, _, gbad = ctx.glv_mul_witness_batch(dev[3], dev[4], torch.from_numpy(kk).cuda())
    torch.cuda.synchronize()
    want, _, _ = oracle_c.glv_mul_witness(sigs[3][sample], sigs[4][sample], k)
    assert gbad == 0 and np.array_equal(gcols[:, torch.from_numpy(sample).cuda()].cpu().numpy().view(np.uint64), want)


def test_full_size_batch_properties():
    """BASELINE metric config: 2^16 verifies on one GPU.  Size-independent properties:
    every synthetic signature verifies (the r == x connect constraint at the END of the 3.5k-op chain),
    no error flags, every 29-bit limb column is < 2^29, and 48 sampled signatures are bit-exact vs the oracle."""
    import torch
    import plonky2_ecdsa_amd as p2e
    n = 1 << 16
    sigs = p2e.synth_signatures(seed=4, n=n)
    ctx = p2e.Context(device=0)
    dev = [torch.from_numpy(a).cuda() for a in sigs]
    cols, err, valid, bad = ctx.ecdsa_verify_witness_batch(*dev)
    torch.cuda.synchronize()
    assert bad == 0 and int(valid.sum()) == n and int(err.sum()) == 0
    sched = p2e.schedule_describe(p2e.PROGRAM_VERIFY)
    limb_rows = []
    for kind, field, c0, nc, label in sched:
        if kind == "mul":
            limb_rows += list(range(c0, c0 + 18))
        elif kind == "inv":
            limb_rows += list(range(c0, c0 + 18))
        elif kind in ("add", "sub", "add_many"):
            limb_rows += list(range(c0, c0 + 9))
    idx = torch.tensor(limb_rows[::7], device="cuda")          # every 7th limb column: 7.6k columns x 65536
    assert int((cols[idx] >> 29).ne(0).sum()) == 0
    sample = np.linspace(0, n - 1, 48).astype(np.int64)
    want, want_aux, _, _ = oracle_c.verify_witness_aux(*[a[sample] for a in sigs])
    got = cols[:, torch.from_numpy(sample).cuda()].cpu().numpy().view(np.uint64)
    assert np.array_equal(got, want)
    # built-in-generator columns of the same batch, padded stride, device pointers
    ld = cols.stride(0)
    aux_full = torch.zeros((p2e.VERIFY_AUX_COLS, n + 16), dtype=torch.int64, device="cuda")
    aux, aerr, abad = ctx.aux_witness_batch(p2e.PROGRAM_VERIFY, dev[4], cols, n=n, ld=ld, aux=aux_full[:, :n], ld_aux=n + 16)
    torch.cuda.synchronize()
    assert abad == 0 and int(aerr.sum()) == 0
    assert np.array_equal(aux_full[:, torch.from_numpy(sample).cuda()].cpu().numpy().view(np.uint64), want_aux)
    assert int(aux_full[:, n:].ne(0).sum()) == 0
    # size-independent properties: every bit / bool column is 0 or 1, the 66 + 73 selections are 29-bit limbs
    items = p2e.aux_describe(p2e.PROGRAM_VERIFY)
    k, first, ncols, _ = items[0]
    assert k == "split4" and int((aux_full[first:first + 261] >> 1).ne(0).sum()) == 0
    sel_rows = [r for kind, f, _n, _l in items if kind in ("fixed_base_window", "msm_digit")
                for r in range(f + (2 if kind == "fixed_base_window" else 1), f + (2 if kind == "fixed_base_window" else 1) + 18)]
    assert int((aux_full[torch.tensor(sel_rows[::5], device="cuda")] >> 29).ne(0).sum()) == 0
    # constraint-block (U29 gate) columns and gate-internal values of the same batch: 65 GB of u32, every value a U29
    # value; three sampled signatures go through the constraint replay with all four matrices attached
    import check_circuit as CC
    del aux_full, idx
    torch.cuda.empty_cache()
    ux, uerr, ubad = ctx.ux_witness_batch(p2e.PROGRAM_VERIFY, dev, cols, aux)
    gate = ctx.gate_internal_batch(p2e.PROGRAM_VERIFY, aux)
    torch.cuda.synchronize()
    assert ubad == 0 and int(uerr.sum()) == 0
    assert all(int((ux[c0:c0 + 8192] >> 29).ne(0).sum()) == 0 for c0 in range(0, ux.shape[0], 8192))   # in slices: 65 GB
    for i in (0, 30000, n - 1):
        ti = torch.tensor([i], device="cuda")
        c = CC.check_verify(cols[:, ti].cpu().numpy().view(np.uint64)[:, 0], *CC.unpack_inputs(sigs, i),
                            aux=aux[:, ti].cpu().numpy().view(np.uint64)[:, 0], ux=ux[:, ti].cpu().numpy().view(np.uint32)[:, 0])
        assert np.array_equal(gate[:, ti].cpu().numpy().view(np.uint64)[:, 0], np.array(c.gate, dtype=np.uint64))


def test_gpu_output_passes_the_constraint_replay():
    """No oracle VALUES involved: the HIP kernels' columns (and built-in-generator columns) of fresh signatures go
    straight into the reference's constraint equations (oracle/check_circuit.py: every add / sub / add_many / inv
    relation with its operands, both mul gates, the range checks, GLV, the three connects), and a corrupted GPU
    column is rejected."""
    import torch
    import check_circuit as CC
    import plonky2_ecdsa_amd as p2e
    n = 300
    sigs = p2e.synth_signatures(seed=2024, n=n)
    ctx = p2e.Context(device=0)
    dev = [torch.from_numpy(a).cuda() for a in sigs]
    cols, err, valid, bad = ctx.ecdsa_verify_witness_batch(*dev)
    aux, _aerr, abad = ctx.aux_witness_batch(0, dev[4], cols, n=n, ld=cols.stride(0))
    torch.cuda.synchronize()
    assert bad == 0 and abad == 0 and int(valid.sum()) == n
    host, haux = cols.cpu().numpy().view(np.uint64), aux.cpu().numpy().view(np.uint64)
    for i in (0, 63, 64, 255, 299):                       # both half-waves, a workgroup boundary, the ragged tail
        ins = CC.unpack_inputs(sigs, i)
        c = CC.check_verify(host[:, i], *ins, aux=haux[:, i])
        assert len(c.gens) == 3555
    bad_cols = host[:, 7].copy()
    bad_cols[41234] ^= np.uint64(1)
    with pytest.raises(CC.ConstraintViolation):
        CC.check_verify(bad_cols, *CC.unpack_inputs(sigs, 7))
    # glv_mul program
    rng = R.SplitMix64(2025)
    ks = [rng.below(R.N) for _ in range(n)]
    gcols, gerr, _gvalid, gbad = ctx.glv_mul_witness_batch(dev[3], dev[4], torch.from_numpy(oracle_c.pack256(ks)).cuda())
    torch.cuda.synchronize()
    assert gbad == 0
    ghost = gcols.cpu().numpy().view(np.uint64)
    for i in (1, 298):
        px, py = CC.unpack_inputs(sigs[3:5], i)
        CC.check_glv_mul(ghost[:, i], px, py, ks[i])


@pytest.mark.parametrize("container", ["u64", "compact", "rows"])
def test_cfg5_streamed_chunks(container):
    """BASELINE config 5 (witness columns streamed to the host prover; one GPU's leg of it): 2^16 + 300 signatures in
    2^13-signature chunks through the double-buffered D2H pipeline (plonky2_ecdsa_amd.stream.HostStreamer), the last
    chunk ragged.  EVERY chunk is compared with the oracle on the HOST copy (first / middle / last signature of the chunk,
    every column), all signatures verify, nothing is flagged."""
    import torch
    import plonky2_ecdsa_amd as p2e
    from plonky2_ecdsa_amd.stream import HostStreamer
    chunk, total = 1 << 13, (1 << 16) + 300
    sigs = p2e.synth_signatures(seed=5, n=total)
    dev = [torch.from_numpy(a).cuda() for a in sigs]
    hs = HostStreamer(device=0, chunk=chunk, container=container)
    seen = []

    def consumer(ch):
        idx = np.unique(np.array([0, ch.n // 2, ch.n - 1]))
        want, werr, wflags = oracle_c.verify_witness(*[a[ch.first + idx] for a in sigs])
        if container == "compact":
            got = p2e.compact_expand(0, ch.narrow[:, idx].contiguous().numpy().view(np.uint32), ch.wide[:, idx].contiguous().numpy())
        elif container == "rows":
            got = ch.rows[idx].t().contiguous().numpy().view(np.uint64)
        else:
            got = ch.cols[:, idx].contiguous().numpy().view(np.uint64)
        assert np.array_equal(got, want), f"chunk {ch.index}: host copy differs from the oracle"
        assert not ch.err.any() and ch.valid.all() and not werr.any() and wflags.all()
        seen.append((ch.index, ch.first, ch.n))

    st = hs.run(dev, consumer)
    assert seen == [(k, k * chunk, min(chunk, total - k * chunk)) for k in range(9)] and seen[-1][2] == 300
    assert st["flagged"] == 0 and st["valid"] == total and st["chunks"] == 9


def test_default_strides_follow_the_array():
    """the default output of ecdsa_verify_witness_batch at a power-of-two n >= 4096 is a (82615, n) VIEW of a
    padded matrix (row stride n + 16).  Every helper that takes it must read its stride from the array."""
    import torch
    import plonky2_ecdsa_amd as p2e
    n = 4096
    sigs = p2e.synth_signatures(seed=404, n=n)
    ctx = p2e.Context(device=0)
    dev = [torch.from_numpy(a).cuda() for a in sigs]
    cols, _err, valid, bad = ctx.ecdsa_verify_witness_batch(*dev)
    assert bad == 0 and cols.shape == (p2e.VERIFY_COLS, n) and cols.stride(0) == n + 16
    idx = np.array([0, 1, 2047, 4095])
    want, waux, _, _ = oracle_c.verify_witness_aux(*[a[idx] for a in sigs])
    rows = ctx.columns_to_rows(cols)                                   # default ld
    aux, _aerr, abad = ctx.aux_witness_batch(0, dev[4], cols)          # default n, ld, ld_aux
    nar, wid, _cerr, cbad = ctx.columns_compact(0, cols)               # default n, ld
    fnar, fwid, _e, fvalid, fbad = ctx.ecdsa_verify_witness_compact_batch(*dev)   # sliced like the u64 output
    faux, _e2, fabad = ctx.aux_witness_compact_batch(0, dev[4], fnar)
    torch.cuda.synchronize()
    assert abad == cbad == fbad == fabad == 0 and int(fvalid.sum()) == n
    tidx = torch.from_numpy(idx).cuda()
    assert np.array_equal(rows[tidx].t().cpu().numpy().view(np.uint64), want)
    assert np.array_equal(aux[:, tidx].cpu().numpy().view(np.uint64), waux)
    assert fnar.shape[1] == n and fwid.shape[1] == n and fnar.stride(0) == n + 16
    for a, b in ((nar, wid), (fnar, fwid)):
        got = p2e.compact_expand(0, a[:, tidx].cpu().numpy().view(np.uint32), b[:, tidx].cpu().numpy())
        assert np.array_equal(got, want)
    assert np.array_equal(faux[:, tidx].cpu().numpy().view(np.uint32).astype(np.uint64), waux)
    with pytest.raises(p2e.P2EError):
        ctx.columns_to_rows(cols.t())                                  # batch dimension not dense


def test_failed_calls_clean_up_and_the_context_survives():
    """Failure paths of the boundary: a staged (host-pointer) call that cannot allocate returns P2E_E_NOMEM without
    touching the host buffers or crashing, leaks nothing that blocks the next call, and a following valid call on the
    same context succeeds; the same for a device-pointer call whose scratch does not fit."""
    import torch
    import plonky2_ecdsa_amd as p2e
    c = p2e.C
    n = 300
    sigs = p2e.synth_signatures(seed=77, n=n)
    want, _, _ = oracle_c.verify_witness(*[a[:4] for a in sigs])
    hctx = p2e.Context(device=0, host_pointers=True)
    huge = 1 << 44                                                     # 2^44 * 32 B of staging: cannot exist
    small = np.zeros((4, 32), dtype=np.uint8)
    out, e = np.zeros((p2e.VERIFY_COLS, 4), dtype=np.uint64), np.zeros(4, dtype=np.uint8)
    rc = hctx._L.p2e_ecdsa_verify_witness_batch(hctx._h, *[p2e._ptr(small)] * 5, p2e._ptr(out), c.c_size_t(huge), c.c_size_t(huge),
                                                p2e._ptr(e), p2e._ptr(e))
    assert rc == -4 and b"allocation" in hctx._L.p2e_last_error()      # P2E_E_NOMEM
    x = np.zeros((9, 4), dtype=np.uint64)
    rc = hctx._L.p2e_inv_witness_batch(hctx._h, c.c_int(0), p2e._ptr(x), p2e._ptr(x), p2e._ptr(x), c.c_size_t(huge), c.c_size_t(huge),
                                       p2e._ptr(e))
    assert rc == -4
    got, err, valid, bad = hctx.ecdsa_verify_witness_batch(*[a[:4] for a in sigs])
    assert bad == 0 and valid.all() and np.array_equal(np.asarray(got).view(np.uint64), want)
    # device pointers: the scratch for 2^40 signatures does not fit the HBM
    dctx = p2e.Context(device=0)
    dev = [torch.from_numpy(a).cuda() for a in sigs]
    cols = torch.empty((p2e.VERIFY_COLS, 4), dtype=torch.int64, device="cuda")
    de = torch.empty(4, dtype=torch.uint8, device="cuda")
    rc = dctx._L.p2e_ecdsa_verify_witness_batch(dctx._h, *[p2e._ptr(d) for d in dev], p2e._ptr(cols), c.c_size_t(1 << 40),
                                                c.c_size_t(1 << 40), p2e._ptr(de), p2e._ptr(de))
    assert rc == -4
    got, err, valid, bad = dctx.ecdsa_verify_witness_batch(*dev)
    torch.cuda.synchronize()
    assert bad == 0 and int(valid.sum()) == n and np.array_equal(got[:, :4].cpu().numpy().view(np.uint64), want)
    # the caller's current device is left alone (single-GPU box: still 0) and the library reports a sane status
    assert torch.cuda.current_device() == 0 and dctx.sync() == 0


@pytest.mark.parametrize("program", [0, 1])
def test_constraint_block_columns_match_the_replay_model(program):
    """SURVEY 8(f) rank 2 (p2e_ux_witness_batch): the U29-gate values of every constraint block, derived on the GPU from
    its own witness + aux matrices through the wiring table, against the constraint replay's model (which recomputes
    them from the reference's gadgets while checking every constraint on the same GPU witness).  n = 300: a full
    workgroup of paired stores + a ragged tail; u32 and u64 outputs, padded strides."""
    import torch
    import check_circuit as CC
    import plonky2_ecdsa_amd as p2e
    n = 300
    sigs = p2e.synth_signatures(seed=3030, n=n)
    ctx = p2e.Context(device=0)
    dev = [torch.from_numpy(a).cuda() for a in sigs]
    if program == 0:
        cols, _e, valid, bad = ctx.ecdsa_verify_witness_batch(*dev)
        inputs, ks = dev, None
    else:
        rng = R.SplitMix64(3031)
        ks = [rng.below(R.N) for _ in range(n)]
        kd = torch.from_numpy(oracle_c.pack256(ks)).cuda()
        cols, _e, valid, bad = ctx.glv_mul_witness_batch(dev[3], dev[4], kd)
        inputs = [dev[3], dev[4], kd]
    aux, _ae, abad = ctx.aux_witness_batch(program, dev[4], cols)
    k = p2e.ux_num_cols(program)
    big32 = torch.full((k, n + 6), -1, dtype=torch.int32, device="cuda")
    ux32, e32, b32 = ctx.ux_witness_batch(program, inputs, cols, aux, ux=big32[:, :n])
    ux64, e64, b64 = ctx.ux_witness_batch(program, inputs, cols, aux, u32=False)
    torch.cuda.synchronize()
    assert bad == 0 and abad == 0 and b32 == 0 and b64 == 0 and ux64.shape == (k, n)
    assert bool((big32[:, n:] == -1).all())                                 # padding untouched
    h32 = ux32.cpu().numpy().view(np.uint32)
    assert np.array_equal(h32.astype(np.uint64), ux64.cpu().numpy().view(np.uint64)) and int(h32.max()) < 1 << 29
    host, haux = cols.cpu().numpy().view(np.uint64), aux.cpu().numpy().view(np.uint64)
    for i in (0, 255, 256, 299):
        if program == 0:
            c = CC.check_verify(host[:, i], *CC.unpack_inputs(sigs, i), aux=haux[:, i], ux=h32[:, i])
        else:
            px, py = CC.unpack_inputs(sigs[3:5], i)
            c = CC.check_glv_mul(host[:, i], px, py, ks[i], aux=haux[:, i], ux=h32[:, i])
        assert len(c.ux) == k
    # host-pointer (staged) path on a tiny batch, u32
    hctx = p2e.Context(device=0, host_pointers=True)
    sl = [a[:3] for a in sigs]
    if program == 0:
        hc, _, _, _ = hctx.ecdsa_verify_witness_batch(*sl)
        hin = sl
    else:
        hk = oracle_c.pack256(ks[:3])
        hc, _, _, _ = hctx.glv_mul_witness_batch(sl[3], sl[4], hk)
        hin = [sl[3], sl[4], hk]
    ha, _, _ = hctx.aux_witness_batch(program, sl[4], hc)
    hu, he, hb = hctx.ux_witness_batch(program, hin, hc, ha)
    assert hb == 0 and np.array_equal(np.asarray(hu), h32[:, :3])


@pytest.mark.parametrize("program", [0, 1])
def test_gate_internal_values_on_the_gpu(program):
    """p2e_gate_internal_batch (equality-gadget internals, RandomAccessGate index bits per window) from the GPU's own aux
    matrix against the constraint replay's model; n = 300, padded stride, staged host-pointer path."""
    import torch
    import check_circuit as CC
    import plonky2_ecdsa_amd as p2e
    n = 300
    sigs = p2e.synth_signatures(seed=4040, n=n)
    ctx = p2e.Context(device=0)
    dev = [torch.from_numpy(a).cuda() for a in sigs]
    if program == 0:
        cols, _e, _v, bad = ctx.ecdsa_verify_witness_batch(*dev)
        ks = None
    else:
        rng = R.SplitMix64(4041)
        ks = [rng.below(R.N) for _ in range(n)]
        cols, _e, _v, bad = ctx.glv_mul_witness_batch(dev[3], dev[4], torch.from_numpy(oracle_c.pack256(ks)).cuda())
    aux, _ae, abad = ctx.aux_witness_batch(program, dev[4], cols)
    k = p2e.VERIFY_GATE_COLS if program == 0 else p2e.GLV_MUL_GATE_COLS
    big = torch.full((k, n + 10), -1, dtype=torch.int64, device="cuda")
    gate = ctx.gate_internal_batch(program, aux, gate=big[:, :n])
    torch.cuda.synchronize()
    assert bad == 0 and abad == 0 and bool((big[:, n:] == -1).all())
    host, hg = cols.cpu().numpy().view(np.uint64), gate.cpu().numpy().view(np.uint64)
    for i in (0, 255, 256, 299):
        if program == 0:
            c = CC.check_verify(host[:, i], *CC.unpack_inputs(sigs, i))
        else:
            px, py = CC.unpack_inputs(sigs[3:5], i)
            c = CC.check_glv_mul(host[:, i], px, py, ks[i])
        assert np.array_equal(hg[:, i], np.array(c.gate, dtype=np.uint64))
    hctx = p2e.Context(device=0, host_pointers=True)
    assert np.array_equal(np.asarray(hctx.gate_internal_batch(program, aux.cpu().numpy().view(np.uint64)[:, :7].copy())), hg[:, :7])


def test_assemble_wires_against_numpy():
    """SURVEY 8(f) rank 3 (p2e_assemble_wires): every column of the three matrices scattered into one plonky2 wire matrix
    per signature through a host-supplied (column -> wire * degree + row) map -- here the synthetic standard_ecc_config
    placement of plonky2_ecdsa_amd.wiremap (378 878 entries, 136 wires x 8192 rows, mul-gate rows with repeated sources)
    and a random permutation map -- against the same scatter in numpy.  n = 96: one full tile of signatures + a ragged one."""
    import torch
    import plonky2_ecdsa_amd as p2e
    from plonky2_ecdsa_amd.wiremap import synthetic_wire_map, WIRE_SRC_AUX, WIRE_SRC_UX
    n = 96
    sigs = p2e.synth_signatures(seed=888, n=n)
    ctx = p2e.Context(device=0)
    dev = [torch.from_numpy(a).cuda() for a in sigs]
    cols, _e, valid, bad = ctx.ecdsa_verify_witness_batch(*dev)
    aux, _ae, abad = ctx.aux_witness_batch(0, dev[4], cols)
    ux, _ue, ubad = ctx.ux_witness_batch(0, dev, cols, aux)                   # u32
    assert bad == abad == ubad == 0 and int(valid.sum()) == n
    hc, ha, hu = cols.cpu().numpy().view(np.uint64), aux.cpu().numpy().view(np.uint64), ux.cpu().numpy().view(np.uint32)
    hg = None

    def reference(src, dst, cells):
        out = np.zeros((n, cells), dtype=np.uint64)
        kind, col = src >> 30, src & 0x3FFFFFFF
        for k, m in ((0, hc), (1, ha), (2, hu), (3, hg)):
            sel = kind == k
            if sel.any():
                out[:, dst[sel]] = m[col[sel]].T
        return out

    gate = ctx.gate_internal_batch(0, aux)
    torch.cuda.synchronize()
    hg = gate.cpu().numpy().view(np.uint64)
    src, dst, nw, deg = synthetic_wire_map(0, with_gate=True)
    assert nw == 136 and len(src) == 378878 + p2e.VERIFY_GATE_COLS
    wm = ctx.wire_map(0, src, dst, nw, deg)
    wires = ctx.assemble_wires(wm, cols, aux, ux, gate)
    torch.cuda.synchronize()
    assert np.array_equal(wires.cpu().numpy().view(np.uint64), reference(src, dst, nw * deg))
    # untouched cells keep the caller's values; a random map (sources repeated, destinations scattered); u64 ux matrix
    rng = np.random.default_rng(5)
    cells = 1 << 18
    cnt = 100_001
    rsrc = np.concatenate([rng.integers(0, p2e.VERIFY_COLS, cnt // 3), WIRE_SRC_AUX | rng.integers(0, p2e.VERIFY_AUX_COLS, cnt // 3),
                           WIRE_SRC_UX | rng.integers(0, p2e.VERIFY_UX_COLS, cnt - 2 * (cnt // 3))]).astype(np.uint32)
    rdst = rng.permutation(cells)[:cnt].astype(np.uint32)
    wm2 = ctx.wire_map(0, rsrc, rdst, 64, cells // 64)
    ux64, _, _ = ctx.ux_witness_batch(0, dev, cols, aux, u32=False)
    pre = torch.full((n, cells + 5), -7, dtype=torch.int64, device="cuda")
    ctx.assemble_wires(wm2, cols, aux, ux64, wires=pre[:, :cells])
    torch.cuda.synchronize()
    got = pre.cpu().numpy()
    want = reference(rsrc, rdst, cells).view(np.int64)
    named = np.zeros(cells, dtype=bool)
    named[rdst] = True
    assert np.array_equal(got[:, :cells][:, named], want[:, named])
    assert (got[:, :cells][:, ~named] == -7).all() and (got[:, cells:] == -7).all()
    # map validation: duplicate destination, destination / source out of range
    for s_, d_ in (([1, 2], [5, 5]), ([1], [64 * 16]), ([p2e.VERIFY_COLS], [0]), ([WIRE_SRC_UX | p2e.VERIFY_UX_COLS], [0])):
        with pytest.raises(p2e.P2EError):
            ctx.wire_map(0, np.array(s_, np.uint32), np.array(d_, np.uint32), 64, 16)
    # a map that reads only the witness matrix needs no other matrix
    only = ctx.wire_map(0, np.arange(100, dtype=np.uint32), np.arange(100, dtype=np.uint32)[::-1].copy(), 10, 10)
    w3 = ctx.assemble_wires(only, cols)
    assert np.array_equal(w3.cpu().numpy().view(np.uint64)[:, ::-1][:, :100], hc[:100].T)
    # staged host-pointer path
    hctx = p2e.Context(device=0, host_pointers=True)
    hw = hctx.assemble_wires(hctx.wire_map(0, rsrc, rdst, 64, cells // 64), hc[:, :5].copy(), ha[:, :5].copy(), hu[:, :5].copy())
    assert np.array_equal(np.asarray(hw)[:, named], want[:5][:, named].view(np.uint64))


def test_per_launch_timing_is_opt_in_and_changes_no_value():
    """include/p2e.h P2E_CTX_PHASE_TIMING: a context created with it reports a duration for the scalar kernel, for the
    expansion launches and for the whole call (p2e_last_phase_ms); the default context reports the same launch and
    column counts with zero durations -- and both write the same witness."""
    import torch
    import plonky2_ecdsa_amd as p2e
    n = 2048 + 5
    dev = [torch.from_numpy(a).cuda() for a in p2e.synth_signatures(seed=77, n=n)]
    plain, timed = p2e.Context(device=0), p2e.Context(device=0, phase_timing=True)
    a = plain.ecdsa_verify_witness_batch(*dev)
    b = timed.ecdsa_verify_witness_batch(*dev)
    torch.cuda.synchronize()
    assert a[3] == 0 and b[3] == 0 and torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])
    pa, pb = plain.last_phase_ms(), timed.last_phase_ms()
    for k in ("expand_launches", "expand_cols", "runs_launches", "runs_cols", "fbrun_launches", "fbrun_cols"):
        assert pa[k] == pb[k]
    assert pa["expand_cols"] + pa["runs_cols"] + pa["fbrun_cols"] > 80000
    assert pa["scalar"] == 0 and pa["total"] == 0 and pa["expand"] == 0 and pa["runs"] == 0
    assert pb["scalar"] > 0 and pb["total"] > pb["scalar"] and pb["expand"] + pb["runs"] > 0


@pytest.mark.parametrize("plan", ["four_lanes", "lane_per_signature_runs"])
def test_column_blocks_are_final_when_their_event_fires(monkeypatch, plan):
    """include/p2e.h p2e_segments_describe / p2e_segment_stream_wait (what the pipelined assembly of a
    sharded batch stands on).  The blocks of an ASYNCHRONOUS fill must be disjoint, cover every column, and block k must
    hold its final values as soon as its event has fired: a copy stream waits for each block's event and copies the block
    out of the (poisoned) output while the later launches are still running; the copies together must be the oracle's
    matrix.  Both containers; the four-lane plan (two expansion streams) and the run plan."""
    import torch
    import plonky2_ecdsa_amd as p2e
    if plan == "lane_per_signature_runs":
        monkeypatch.setenv("P2E_QUAD_MAX_N", "0")
        monkeypatch.setenv("P2E_RUNS_MIN_N", "0")
    n = 3000 + 77
    sigs = p2e.synth_signatures(seed=91, n=n)
    want, _, _ = oracle_c.verify_witness_lockstep(*sigs)
    want_t = torch.from_numpy(want.view(np.int64)).cuda()
    work, copy = torch.cuda.Stream(), torch.cuda.Stream()
    ctx = p2e.Context(device=0, stream=work.cuda_stream, asynchronous=True)
    dev = [torch.from_numpy(a).cuda() for a in sigs]
    cols = torch.full((p2e.VERIFY_COLS, n + 1), -1, dtype=torch.int64, device="cuda")
    snap = torch.full((p2e.VERIFY_COLS, n), -2, dtype=torch.int64, device="cuda")
    err = torch.zeros(n, dtype=torch.uint8, device="cuda")
    valid = torch.zeros(n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.ecdsa_verify_witness_batch(*dev, cols=cols[:, :n], err=err, valid=valid, ld=n + 1)
    segs = ctx.segments()
    covered = np.zeros(p2e.VERIFY_COLS, dtype=np.int32)
    with torch.cuda.stream(copy):
        for k, (c0, nc) in enumerate(segs):
            covered[c0:c0 + nc] += 1
            ctx.segment_stream_wait(k, copy.cuda_stream)
            snap[c0:c0 + nc].copy_(cols[c0:c0 + nc, :n], non_blocking=True)
    assert (covered == 1).all() and len(segs) >= 6
    copy.synchronize()
    assert ctx.sync() == 0
    assert torch.equal(snap, want_t) and torch.equal(cols[:, :n], want_t) and bool((cols[:, n] == -1).all())
    # the host-blocking form, and the compact container (a block of columns = a block of rows of each matrix)
    from plonky2_ecdsa_amd.dist import compact_row_ranges
    cmap, nn, nw = p2e.compact_layout(0)
    nar = torch.full((nn, n + 1), -1, dtype=torch.int32, device="cuda")
    wid = torch.full((nw, n + 1), -1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ctx.ecdsa_verify_witness_compact_batch(*dev, narrow=nar[:, :n], wide=wid[:, :n], err=err, valid=valid, ld_narrow=n + 1, ld_wide=n + 1)
    segs2 = ctx.segments()
    nr, wr = compact_row_ranges(cmap, segs2)
    got_n, got_w = [], []
    for k in range(len(segs2)):
        ctx.segment_sync(k)
        with torch.cuda.stream(copy):
            got_n.append((nr[k], nar[nr[k][0]:nr[k][0] + nr[k][1], :n].clone()))
            got_w.append((wr[k], wid[wr[k][0]:wr[k][0] + wr[k][1], :n].clone()))
    copy.synchronize()
    assert ctx.sync() == 0
    is_wide = (cmap & p2e.COMPACT_WIDE) != 0
    want_n = (want_t[torch.from_numpy(np.nonzero(~is_wide)[0]).cuda()])
    want_w = want_t[torch.from_numpy(np.nonzero(is_wide)[0]).cuda()]
    for (r0, k_), blk in got_n:
        assert torch.equal(blk.to(torch.int64) & 0xFFFFFFFF, want_n[r0:r0 + k_])
    for (r0, k_), blk in got_w:
        assert torch.equal(blk, want_w[r0:r0 + k_])
    assert sum(k_ for (_, k_), _ in got_n) == nn and sum(k_ for (_, k_), _ in got_w) == nw
    with pytest.raises(p2e.P2EError):
        ctx.segment_sync(len(segs2))
    ctx.close()


@pytest.mark.parametrize("layout", ["MFB2", "2P1BFM"])
def test_other_stream_layouts_give_the_same_witness(ora, monkeypatch, layout):
    """P2E_STREAM_LAYOUT (csrc/p2e_hip.hip p2e_ctx_create): the creation / queue-binding order of the library's streams, with
    ("1") or without an own first expansion stream (without: the round-2 form, expansions on the caller's stream), spare
    streams ("P").  A tuning knob -- whatever it is set to, every plan must produce the oracle's matrix."""
    import torch
    import plonky2_ecdsa_amd as p2e
    monkeypatch.setenv("P2E_STREAM_LAYOUT", layout)
    ctx = p2e.Context(device=0)
    for n, env in ((700, None), (700, ("P2E_QUAD_MAX_N", "0"))):
        if env:
            monkeypatch.setenv(*env)
            ctx = p2e.Context(device=0)
        sigs = p2e.synth_signatures(seed=123, n=n)
        dev = [torch.from_numpy(a).cuda() for a in sigs]
        cols, err, valid, bad = ctx.ecdsa_verify_witness_batch(*dev)
        torch.cuda.synchronize()
        want, _, wflags = ora.verify(*sigs)
        assert bad == 0 and np.array_equal(cols.cpu().numpy().view(np.uint64), want) and np.array_equal(valid.cpu().numpy(), wflags)
        assert len(ctx.segments()) >= 6
