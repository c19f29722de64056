"""GPU (-m gpu): the assembly of a sharded batch with REAL contexts -- two gloo ranks sharing the one GPU of the box
(the N > 1 code path of bench.py's `strong` object; on a multi-GPU node the same code runs one rank per GPU over RCCL).

Each rank fills its shard of ONE global batch through an asynchronous context and hands the column blocks over as their
events fire (p2e_segments_describe / p2e_segment_sync -> dist.assemble_fill, host path: device stage, per-block copy,
grouped isend/irecv); rank 0 then compares the WHOLE assembled matrix -- every column of every signature of both shards
-- with the C oracle.  Uneven shards (n odd), the u64 matrix and the compact container.  SURVEY.md 8(e)."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, compact, q):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import plonky2_ecdsa_amd as p2e
    from plonky2_ecdsa_amd import dist as pd
    import oracle_c
    torch.cuda.set_device(0)
    start, end = pd.shard_bounds(total, rank, world)
    n = end - start
    n_max = -(-total // world)
    ld = n_max + 2
    sigs = p2e.synth_signatures(seed=23, n=n, first=start)
    dev = [torch.from_numpy(a).cuda() for a in sigs]
    st = torch.cuda.Stream()
    ctx = p2e.Context(device=0, stream=st.cuda_stream, asynchronous=True)
    err = torch.zeros(n, dtype=torch.uint8, device="cuda")
    valid = torch.zeros(n, dtype=torch.uint8, device="cuda")
    cmap = None
    if compact:
        cmap, nn, nw = p2e.compact_layout(0)
        shapes = [((nn, ld), torch.int32), ((nw, ld), torch.int64)]
    else:
        shapes = [((p2e.VERIFY_COLS, ld), torch.int64)]
    mats = [torch.full((world,) + sh, -7, dtype=dt) for sh, dt in shapes]                 # host assembly (gloo)
    stage = [torch.full(sh, -9, dtype=dt, device="cuda") for sh, dt in shapes]            # what the fill writes
    asm = pd.ColumnAssembly(mats, total)
    torch.cuda.synchronize()

    def issue():
        if compact:
            ctx.ecdsa_verify_witness_compact_batch(*dev, narrow=stage[0][:, :n], wide=stage[1][:, :n], err=err, valid=valid,
                                                   ld_narrow=ld, ld_wide=ld)
        else:
            ctx.ecdsa_verify_witness_batch(*dev, cols=stage[0][:, :n], err=err, valid=valid, ld=ld)

    segs = pd.assemble_fill(ctx, asm, issue, compact_map=cmap, host_stage=stage)
    why = []
    ok = ctx.sync() == 0 and int(valid.sum()) == n and len(segs) >= 6
    if not ok:
        why.append(f"fill: valid {int(valid.sum())} of {n}, {len(segs)} blocks")
    if rank == 0:
        want, werr, wflags = oracle_c.verify_witness_lockstep(*p2e.synth_signatures(seed=23, n=total))
        ok = ok and not werr.any() and wflags.all()
        for r in range(world):
            s_, e_ = pd.shard_bounds(total, r, world)
            if compact:
                got = p2e.compact_expand(0, mats[0][r, :, :e_ - s_].numpy().view(np.uint32), mats[1][r, :, :e_ - s_].numpy())
            else:
                got = mats[0][r, :, :e_ - s_].numpy().view(np.uint64)
            if not np.array_equal(got, want[:, s_:e_]):
                bad_cols = np.nonzero((got != want[:, s_:e_]).any(axis=1))[0]
                why.append(f"shard of rank {r}: {len(bad_cols)} columns differ, first {bad_cols[:5]}")
                ok = False
            if e_ - s_ < n_max and not all(bool((m[r, :, e_ - s_:n_max] == 0).all()) for m in mats):
                why.append(f"pad column of rank {r}'s shorter shard is not zero")       # the shorter shard's pad column
                ok = False
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()
    q.put((rank, bool(ok), "; ".join(why)))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("compact", [False, True], ids=["u64", "compact"])
def test_two_ranks_assemble_their_shards_block_by_block(compact):
    import torch.multiprocessing as mp
    world, total = 2, 2 * 1500 + 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, compact, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(400)
    res = sorted(q.get(timeout=10) for _ in range(world))
    assert res == [(0, True, ""), (1, True, "")] and all(p.exitcode == 0 for p in procs)
