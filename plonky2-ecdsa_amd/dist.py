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


def all_gather_columns(local, total: int, group=None):
    """local: (cols, n_local) int64/uint64 tensor of this rank's shard.  Returns (world, cols, n_max)
    where n_max = max shard size (shards shorter than n_max are zero padded on the right)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    n_max = -(-total // world)
    cols, n_local = local.shape
    if n_local != n_max:
        padded = torch.zeros((cols, n_max), dtype=local.dtype, device=local.device)
        padded[:, :n_local] = local
        local = padded
    out = torch.empty((world * cols, n_max), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)  # concatenation along dim 0
    return out.view(world, cols, n_max)


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
