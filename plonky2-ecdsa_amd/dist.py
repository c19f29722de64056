"""Multi-GPU sharding of a signature batch: one process per GPU (torch.distributed, backend "nccl" =
RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The path is embarrassingly parallel over signatures (one circuit instance per
signature, no cross-signature data flow), so ranks take contiguous ranges of the batch and there is NO
collective on the data path.  The only exchange the north star names is the *assembly* of the witness
columns on every rank: ``all_gather_columns`` does it with ONE all_gather_into_tensor per call (one
large collective instead of one per column: xGMI is point-to-point, 7 links per GPU, so a single big
gather keeps every link busy), producing ``(world, cols, n_local)``; signature ``j`` of the global batch
is ``(rank, :, i)`` with ``rank, i = divmod-like shard_of(j)``.  No transposition is needed because every
rank's shard is already column-major over ITS signatures.
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
    accepted and sliced.  Returns (world, cols, n_max) with n_max = the largest shard; shorter shards are padded on
    the right (zeros on the packing path, the buffer's own pad otherwise)."""
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
