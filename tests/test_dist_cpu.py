"""CPU, world_size 2, gloo: the N>1 path of bench.py / plonky2-ecdsa_amd/dist.py -- contiguous sharding
of the signature batch and the single all_gather that assembles the witness columns.  The per-rank
compute is the CPU emulation of the kernels (test-only); on the GPU box the same code runs the HIP
library.  -m "not gpu"."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, q):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import plonky2_ecdsa_amd as p2e
    from plonky2_ecdsa_amd import dist as pd
    from backends import EmuBackend, OracleBackend
    start, end = pd.shard_bounds(total, rank, world)
    sigs = p2e.synth_signatures(seed=31, n=end - start, first=start)   # each rank synthesises ITS range
    cols, err, valid = EmuBackend().verify(*sigs)
    assert not err.any() and valid.all()
    local = torch.from_numpy(cols.view(np.int64))
    gathered = pd.all_gather_columns(local, total)
    assert gathered.shape == (world, p2e.VERIFY_COLS, -(-total // world))
    ok = True
    if rank == 0:
        all_sigs = p2e.synth_signatures(seed=31, n=total)
        want, _, _ = OracleBackend().verify(*all_sigs)
        for c in (0, 17, 344, 17141, 50000, 82614):
            got = pd.global_column(gathered, c, total).numpy().view(np.uint64)
            ok = ok and np.array_equal(got, want[c])
        for j in range(total):
            r, i = pd.shard_of(j, total, world)
            ok = ok and np.array_equal(gathered[r, :, i].numpy().view(np.uint64), want[:, j])
    # the shard as the fused entry points leave it: the [:, :n] view of a padded (cols, n + pad) buffer, and that
    # buffer unsliced with n_local given.  Equal shards gather straight from the buffer (no packing copy), unequal
    # ones are packed; either way the result is the same matrix.
    n_local = end - start
    big = torch.full((p2e.VERIFY_COLS, n_local + 3), -1, dtype=torch.int64)
    big[:, :n_local] = local
    g2 = pd.all_gather_columns(big[:, :n_local], total)
    g3 = pd.all_gather_columns(big, total, n_local=n_local, out=pd.gather_buffer(big[:, :n_local], world))
    ok = ok and g2.shape == gathered.shape and g3.shape == gathered.shape
    for r in range(world):
        s_, e_ = pd.shard_bounds(total, r, world)
        ok = ok and torch.equal(g2[r, :, :e_ - s_], gathered[r, :, :e_ - s_]) and torch.equal(g3[r, :, :e_ - s_], gathered[r, :, :e_ - s_])
    try:
        pd.all_gather_columns(big, total, n_local=n_local + 4)
        ok = False
    except ValueError:
        pass
    # the same assembly in the compact container (u32 narrow + u64 wide matrices)
    cmap, nn, nw = p2e.compact_layout(0)
    narrow, wide, cerr, _ = EmuBackend().compact(0, sigs, nn, nw)
    gn, gw = pd.all_gather_compact(torch.from_numpy(narrow.view(np.int32)), torch.from_numpy(wide.view(np.int64)), total)
    assert not cerr.any() and gn.shape == (world, nn, -(-total // world)) and gw.shape[1] == nw
    if rank == 0:
        for c in (0, 17, 18, 50, 344, 17141, 50000, 82614):     # limbs, check_sum and carry columns
            got = pd.global_column_compact(gn, gw, cmap, c, total).numpy().view(np.uint64)
            ok = ok and np.array_equal(got, want[c])
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok)))


def test_shard_bounds_cover_batch():
    sys.path.insert(0, ROOT)
    from plonky2_ecdsa_amd import dist as pd
    for total in (1, 5, 8, 65536, 65537):
        for world in (1, 2, 3, 8):
            spans = [pd.shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1
            for j in (0, total // 2, total - 1):
                r, i = pd.shard_of(j, total, world)
                assert spans[r][0] + i == j


@pytest.mark.timeout(600)
@pytest.mark.parametrize("total", [5, 4], ids=["uneven_3_2", "even_2_2"])
def test_two_rank_gloo_shard_and_gather(total):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    res = sorted(q.get(timeout=10) for _ in range(world))
    assert res == [(0, True), (1, True)] and all(p.exitcode == 0 for p in procs)
