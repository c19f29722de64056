"""Multi-GPU sharding of a signature batch: one process per GPU (torch.distributed, backend "nccl" =
RCCL over xGMI on ROCm; "gloo" in the CPU tests and in rehearsals where ranks share a GPU).

The path is embarrassingly parallel over signatures (one circuit instance per signature, no
cross-signature data flow), so ranks take contiguous ranges of the batch and there is NO collective on the data path.
The only exchange the north star names is the *assembly* of the witness columns on every rank.  Two ways to do it:

``ColumnAssembly`` (the one bench.py's strong mode uses): every rank owns ONE tensor ``(world, cols, ld)``; its fill
writes straight into ``[rank]`` (the C ABI takes any ``ld``, so no packing copy), and the blocks of columns are
exchanged AS THE KERNELS FINISH THEM (include/p2e.h p2e_segments_describe: one block per launch that completes
columns, each with its own HIP event): for block ``[c0, c0 + nc)`` a communication stream waits for the block's event
and then one grouped send/recv (``batch_isend_irecv`` -> ncclGroupStart / ncclSend x (world-1) / ncclRecv x (world-1) /
ncclGroupEnd) moves ``[rank, c0:c0+nc, :]`` to every peer's ``[rank, c0:c0+nc, :]``.  That is the direct (one-shot)
all-gather on the fully connected xGMI mesh -- every link carries exactly one shard, no ring hops -- and it lands in
the final layout: signature ``j`` of the global batch is ``[r, :, i]`` with ``r, i = shard_of(j)``, no transposition,
because each shard is already column-major over ITS signatures.  The exchange of finished blocks overlaps the
expansion of later ones; what stays exposed is (gather time - fill time), see DESIGN.md section 6 for the xGMI figures.

``all_gather_columns`` (kept): ONE all_gather_into_tensor after the fill, straight from the padded output buffer.
"""
from __future__ import annotations


def shard_bounds(total: int, rank: int, world: int):
    """Contiguous range [start, end) of the global batch owned by ``rank`` (sizes differ by at most 1)."""
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_of(j: int, total: int, world: int):
    """(rank, local index) of global signature j."""
    base, rem = divmod(total, world)
    cut = rem * (base + 1)
    if j < cut:
        return j // (base + 1), j % (base + 1)
    return rem + (j - cut) // base, (j - cut) % base


def _dense_base(local):
    """The dense (cols, ld) matrix a column-slice view (cols, n <= ld) lives in, without copying; None if the view
    is not of that shape (then the caller packs it)."""
    import torch

    cols, n = local.shape
    if n > 1 and local.stride(1) != 1:
        return None
    ld = local.stride(0) if cols > 1 else n
    if ld == n:
        return local if local.is_contiguous() else None
    if ld < n:
        return None
    have = local.untyped_storage().nbytes() // local.element_size() - local.storage_offset()
    if have < cols * ld:
        return None
    return torch.as_strided(local, (cols, ld), (ld, 1))


def gather_buffer(local, world: int):
    """An output buffer all_gather_columns can reuse across calls (same shape rules as the call itself)."""
    import torch

    base = _dense_base(local)
    rows, ld = (base.shape if base is not None else local.shape)
    return torch.empty((world * rows, ld), dtype=local.dtype, device=local.device)


def all_gather_columns(local, total: int, group=None, n_local=None, out=None):
    """local: this rank's shard of the column matrix, (cols, n_local) int64/uint64/int32 -- typically the
    ``[:, :n]`` view of a padded (cols, ld) output buffer of the fused entry points, which is gathered AS IS (one
    all_gather_into_tensor straight from the buffer, no packing copy: 43 GB at 2^16) whenever every rank has the same
    ld; the pad columns are dropped from the result by a view.  ``n_local`` (default: this rank's shard size from
    shard_bounds) is the number of valid signatures in ``local``; a wider ``local`` (an unsliced padded buffer) is
    accepted and sliced.  Returns (world, cols, n_max) with n_max = the largest shard; the column past a shorter
    shard's last signature is ZERO on both paths (uneven totals: a consumer may reduce over the whole tensor)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_max = -(-total // world)
    if n_local is None:
        s, e = shard_bounds(total, rank, world)
        n_local = e - s
    if n_local > n_max or n_local > local.shape[1]:
        raise ValueError(f"n_local {n_local} exceeds the largest shard {n_max} or the matrix ({local.shape[1]} columns)")
    local = local[:, :n_local]
    cols = local.shape[0]
    base = _dense_base(local)
    ld = base.shape[1] if base is not None else -1
    if ld < n_max:
        ld = -1
    # the no-copy path needs ONE ld on all ranks: agree on it (a two-element all-reduce)
    probe = torch.tensor([ld, -ld], dtype=torch.int64, device=local.device)
    dist.all_reduce(probe, op=dist.ReduceOp.MAX, group=group)
    same_ld = int(probe[0]) == ld and int(probe[1]) == -ld and ld > 0
    if same_ld:
        src = base
        if n_local < n_max:
            src[:, n_local:n_max] = 0      # the pad a shorter shard contributes to the result (one column)
    else:
        ld = n_max
        src = torch.zeros((cols, n_max), dtype=local.dtype, device=local.device)
        src[:, :n_local] = local
    if out is None or tuple(out.shape) != (world * cols, ld) or out.dtype != local.dtype:
        out = torch.empty((world * cols, ld), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, src, group=group)  # concatenation along dim 0
    return out.view(world, cols, ld)[:, :, :n_max]


def global_column(gathered, col: int, total: int):
    """Column ``col`` of the assembled witness, signatures in global order: (total,)."""
    import torch

    world = gathered.shape[0]
    parts = []
    for r in range(world):
        s, e = shard_bounds(total, r, world)
        parts.append(gathered[r, col, : e - s])
    return torch.cat(parts)


def all_gather_compact(narrow, wide, total: int, group=None):
    """The same assembly for the compact container (include/p2e.h p2e_columns_compact): two collectives, one per
    matrix, 28 % fewer bytes over xGMI than the u64 matrix.  Returns (world, num_narrow, n_max) int32 and
    (world, num_wide, n_max) int64."""
    return all_gather_columns(narrow, total, group), all_gather_columns(wide, total, group)


def global_column_compact(gathered_narrow, gathered_wide, col_map, col: int, total: int):
    """Column ``col`` of the assembled witness from the gathered compact container, widened to int64: (total,).
    ``col_map`` is plonky2_ecdsa_amd.compact_layout(program)[0]."""
    import torch

    m = int(col_map[col])
    if m & 0x80000000:
        return global_column(gathered_wide, m & 0x7FFFFFFF, total)
    return global_column(gathered_narrow, m, total).to(torch.int64) & 0xFFFFFFFF


# ---- assembly pipelined with the fill ------------------------------------------------------------------------------
def compact_row_ranges(col_map, segments):
    """Rows of the narrow / wide matrices of the compact container that hold the witness columns of each block of
    ``segments`` [(first_col, num_cols)]: ([(n0, n_rows)], [(w0, w_rows)]).  The layout advances both indices in
    registration order (include/p2e.h p2e_compact_layout), so a block of columns is a block of rows in each."""
    import numpy as np

    m = np.asarray(col_map, dtype=np.uint32)
    is_wide = (m & 0x80000000) != 0
    wide_before = np.concatenate([[0], np.cumsum(is_wide)])
    narrow_before = np.concatenate([[0], np.cumsum(~is_wide)])
    nar = [(int(narrow_before[c0]), int(narrow_before[c0 + nc] - narrow_before[c0])) for c0, nc in segments]
    wid = [(int(wide_before[c0]), int(wide_before[c0 + nc] - wide_before[c0])) for c0, nc in segments]
    return nar, wid


class ColumnAssembly:
    """The assembled column matrix of a sharded batch, exchanged block by block while the fill is still running.

    ``matrices``: one ``(world, rows, ld)`` tensor per matrix of the container (u64 matrix: one; compact container:
    narrow int32 and wide int64), identical shapes on every rank, on the GPU for "nccl" and on the host for "gloo".
    The caller's fill writes this rank's shard into ``local_view(k)`` = ``matrices[k][rank]`` (device tensors: directly,
    with ``ld = local_view(k).stride(0)``), then hands the blocks over with ``exchange`` as they become final:

        asm = ColumnAssembly([buf], total)                                  # buf: (world, 82615, ld) int64
        ctx.ecdsa_verify_witness_batch(*inputs, cols=asm.local_view(0)[:, :n], ld=asm.ld(0), ...)   # asynchronous context
        for k, (c0, nc) in enumerate(ctx.segments()):
            asm.exchange([(c0, nc)], ready=lambda st, k=k: ctx.segment_stream_wait(k, st))
        asm.wait()

    ``ready(stream_handle)`` is called with the raw handle of the communication stream and must make that STREAM wait
    for the block (no host blocking), or be None when the block is final already.  On the host path (CPU tensors) pass
    ``ready=lambda _: ctx.segment_sync(k)`` followed by the copy into ``local_view``.
    """

    def __init__(self, matrices, total: int, group=None):
        import torch.distributed as dist

        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.total = total
        self.matrices = list(matrices)
        for m in self.matrices:
            if m.dim() != 3 or m.shape[0] != self.world or not m.is_contiguous():
                raise ValueError("ColumnAssembly wants contiguous (world, rows, ld) tensors")
        s, e = shard_bounds(total, self.rank, self.world)
        self.n_local = e - s
        self.n_max = -(-total // self.world)
        if any(m.shape[2] < self.n_max for m in self.matrices):
            raise ValueError("ld is smaller than the largest shard")
        self.on_device = self.matrices[0].is_cuda
        self._stream = None
        self._reqs = []
        self.bytes_received = 0
        if self.n_local < self.n_max:              # the one pad column a shorter shard contributes: zero, not stale
            for m in self.matrices:
                m[self.rank, :, self.n_local:self.n_max] = 0

    def ld(self, k: int = 0) -> int:
        return self.matrices[k].shape[2]

    def local_view(self, k: int = 0):
        return self.matrices[k][self.rank]

    def comm_stream(self):
        if self._stream is None and self.on_device:
            import torch

            self._stream = torch.cuda.Stream(device=self.matrices[0].device)
        return self._stream

    def exchange(self, blocks, ready=None):
        """blocks: one (first_row, num_rows) per matrix (``None`` / zero rows: nothing of that matrix in this block).
        Sends this rank's rows to every peer and posts the receives of theirs; returns at once."""
        import torch
        import torch.distributed as dist

        if self.world == 1:
            return
        ops = []
        for m, blk in zip(self.matrices, blocks):
            if not blk or blk[1] == 0:
                continue
            r0, nr = blk
            mine = m[self.rank, r0:r0 + nr]
            for d in range(1, self.world):         # peer order rotated by rank: no two ranks start on the same link
                peer = (self.rank + d) % self.world
                src = (self.rank - d) % self.world
                ops.append(dist.P2POp(dist.isend, mine, peer, self.group))
                ops.append(dist.P2POp(dist.irecv, m[src, r0:r0 + nr], src, self.group))
                self.bytes_received += nr * m.shape[2] * m.element_size()
        if not ops:
            return
        if self.on_device:
            st = self.comm_stream()
            with torch.cuda.stream(st):            # the collective is ordered behind whatever this stream waits for
                if ready is not None:
                    ready(st.cuda_stream)
                self._reqs += dist.batch_isend_irecv(ops)
        else:
            if ready is not None:
                ready(None)
            self._reqs += dist.batch_isend_irecv(ops)

    def wait(self):
        """All posted exchanges are complete (and, on the device, visible to the current stream)."""
        import torch

        for r in self._reqs:
            r.wait()
        self._reqs = []
        if self.on_device and self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)

    def global_column(self, col_row: int, k: int = 0):
        return global_column(self.matrices[k][:, :, : self.n_max], col_row, self.total)


def assemble_fill(ctx, asm: ColumnAssembly, issue, compact_map=None, host_stage=None):
    """Issue one fused fill on ``ctx`` (an ASYNCHRONOUS context: the call must return before the kernels have run) and
    exchange its column blocks through ``asm`` as they complete.

    issue(): makes the fused call, writing into ``asm.local_view(k)`` (device assembly) or into ``host_stage[k]``
    (host assembly: device tensors shaped like ``asm.local_view(k)``, copied to the host block by block).
    compact_map: plonky2_ecdsa_amd.compact_layout(program)[0] when ``asm`` holds the compact container (narrow, wide).
    Returns the list of blocks [(first_col, num_cols)]."""
    import torch.distributed as dist

    issue()
    segs = ctx.segments()
    # every rank must post the same sequence of blocks.  The block list follows the launch plan, which the library picks
    # by batch size: shards that straddle a plan threshold (sizes differ by one) may disagree -- then the whole matrix
    # goes as ONE block behind the last event (a 12-byte agreement per fill, on the host)
    mine = [tuple(x) for x in segs]
    if asm.world > 1 and getattr(asm, "_agreed_blocks", None) != mine:      # (agreed once per assembly and block list)
        everyone = [None] * asm.world
        dist.all_gather_object(everyone, mine, group=asm.group)
        if all(e == mine for e in everyone):
            asm._agreed_blocks = mine
        else:
            rows = [m.shape[1] for m in asm.matrices]
            last = len(segs) - 1

            def ready_all(st):
                for k in range(last + 1):
                    if st is not None:
                        ctx.segment_stream_wait(k, st)
                    else:
                        ctx.segment_sync(k)
                if st is None:
                    for j in range(len(rows)):
                        asm.local_view(j)[:, :asm.n_local].copy_(host_stage[j][:, :asm.n_local])
            asm.exchange([(0, r) for r in rows], ready=ready_all)
            asm.wait()
            return segs
    if compact_map is not None:
        nar, wid = compact_row_ranges(compact_map, segs)
        per_block = [[nar[k], wid[k]] for k in range(len(segs))]
    else:
        per_block = [[segs[k]] for k in range(len(segs))]
    for k, blocks in enumerate(per_block):
        if asm.on_device:
            asm.exchange(blocks, ready=lambda st, k=k: ctx.segment_stream_wait(k, st))
        else:
            def ready(_st, k=k, blocks=blocks):
                ctx.segment_sync(k)
                for j, blk in enumerate(blocks):
                    if blk and blk[1]:
                        r0, nr = blk
                        # (only the shard's own signatures: the pad column of a shorter shard stays zero)
                        asm.local_view(j)[r0:r0 + nr, :asm.n_local].copy_(host_stage[j][r0:r0 + nr, :asm.n_local])
            asm.exchange(blocks, ready=ready)
    asm.wait()
    return segs
