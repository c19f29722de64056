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
    # ---- the assembly pipelined with the fill (ColumnAssembly / assemble_fill): blocks of columns exchanged in the order a
    # producer completes them, straight into the final (world, cols, ld) layout.  The producer here is a stand-in for an
    # asynchronous p2e.Context: block k of the staged shard only holds its values once segment_sync(k) has been called
    # (poison before), and the blocks complete out of column order, as the launch plans' do.
    class FakeCtx:
        def __init__(self, finished, stage, blocks):
            self.finished, self.stage, self.blocks, self.synced = finished, stage, blocks, []

        def segments(self):
            return list(self.blocks)

        def segment_sync(self, k):
            c0, nc = self.blocks[k]
            self.stage[c0:c0 + nc] = self.finished[c0:c0 + nc]
            self.synced.append(k)

    cuts = [0, 344, 17141, 17345, 30000, 30001, 61000, p2e.VERIFY_COLS]
    blocks = [(cuts[i], cuts[i + 1] - cuts[i]) for i in (0, 2, 5, 1, 4, 3, 6)]       # issue order != column order
    n_max = -(-total // world)
    ld = n_max + 2
    asm = pd.ColumnAssembly([torch.full((world, p2e.VERIFY_COLS, ld), -7, dtype=torch.int64)], total)
    stage = torch.full((p2e.VERIFY_COLS, ld), -9, dtype=torch.int64)
    finished = torch.full((p2e.VERIFY_COLS, ld), -9, dtype=torch.int64)
    finished[:, :n_local] = local
    fake = FakeCtx(finished, stage, blocks)
    segs = pd.assemble_fill(fake, asm, issue=lambda: None, host_stage=[stage])
    ok = ok and segs == blocks and sorted(fake.synced) == list(range(len(blocks)))
    got_all = asm.matrices[0]
    for r in range(world):
        s_, e_ = pd.shard_bounds(total, r, world)
        ok = ok and torch.equal(got_all[r, :, :e_ - s_], gathered[r, :, :e_ - s_])
        if e_ - s_ < n_max:
            ok = ok and bool((gathered[r, :, e_ - s_:n_max] == 0).all())        # all_gather_columns: the pad column is zero
    if rank == 0:
        for c in (0, 343, 344, 17141, 30000, 82614):
            ok = ok and np.array_equal(asm.global_column(c).numpy().view(np.uint64), want[c])
    # ranks that disagree on the block list (shards straddling a launch-plan threshold): one block behind the last event
    asm2 = pd.ColumnAssembly([torch.full((world, p2e.VERIFY_COLS, ld), -7, dtype=torch.int64)], total)
    stage2 = torch.full((p2e.VERIFY_COLS, ld), -9, dtype=torch.int64)
    fake2 = FakeCtx(finished, stage2, blocks if rank == 0 else [(0, 17141), (17141, p2e.VERIFY_COLS - 17141)])
    pd.assemble_fill(fake2, asm2, issue=lambda: None, host_stage=[stage2])
    for r in range(world):
        s_, e_ = pd.shard_bounds(total, r, world)
        ok = ok and torch.equal(asm2.matrices[0][r, :, :e_ - s_], gathered[r, :, :e_ - s_])
    # the compact container through the same path: a block of columns is a block of rows of each matrix
    asm3 = pd.ColumnAssembly([torch.full((world, nn, ld), -7, dtype=torch.int32), torch.full((world, nw, ld), -7, dtype=torch.int64)], total)
    st_n, st_w = torch.full((nn, ld), -9, dtype=torch.int32), torch.full((nw, ld), -9, dtype=torch.int64)
    fin_n, fin_w = st_n.clone(), st_w.clone()
    fin_n[:, :n_local] = torch.from_numpy(narrow.view(np.int32))
    fin_w[:, :n_local] = torch.from_numpy(wide.view(np.int64))
    nar_r, wid_r = pd.compact_row_ranges(cmap, blocks)

    class FakeCompact(FakeCtx):
        def segment_sync(self, k):
            (a, na), (b, nb) = nar_r[k], wid_r[k]
            st_n[a:a + na] = fin_n[a:a + na]
            st_w[b:b + nb] = fin_w[b:b + nb]

    pd.assemble_fill(FakeCompact(None, None, blocks), asm3, issue=lambda: None, compact_map=cmap, host_stage=[st_n, st_w])
    ok = ok and sum(x[1] for x in nar_r) == nn and sum(x[1] for x in wid_r) == nw
    if rank == 0:
        for c in (0, 17, 18, 50, 344, 17141, 50000, 82614):
            got = pd.global_column_compact(asm3.matrices[0][:, :, :n_max], asm3.matrices[1][:, :, :n_max], cmap, c, total).numpy().view(np.uint64)
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
