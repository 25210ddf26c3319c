"""Row-tile sharding of the framebuffer across ranks and the single end-of-render gather.

The path shards into independent units (pixels; RNG streams are keyed by the global pixel index and the
sample index), so rank g renders rows [rows[g], rows[g+1]) with no data-path collective; the only
collective is one gather of the float4 row tiles to rank 0 (RCCL over xGMI with backend "nccl";
the CPU tests use "gloo")."""
import torch
import torch.distributed as dist


def partition_rows(height, world):
    """Contiguous row tiles; sizes differ by at most one row."""
    return [(g * height) // world for g in range(world + 1)]


def interleaved_shard(height, world, rank):
    """Rank g renders rows g, g+world, g+2*world ... (rtw_params row0=g, row1=height, row_stride=world): every rank
    gets the same mix of cheap and expensive rows, which contiguous tiles do not (ceiling vs floor of a Cornell box:
    27 % imbalance at 8 tiles). Returns (row0, row1, row_stride, n_rows)."""
    n = max(0, (height - rank + world - 1) // world)
    return rank, height, world, n


def gather_interleaved(tile, height, rank, world, dst=0):
    """tile: (max_rows, W, 4), valid rows first. One gather; rank dst re-interleaves the rows into the (H, W, 4) frame."""
    if world == 1:
        return tile[:height]
    gathered = [torch.empty_like(tile) for _ in range(world)] if rank == dst else None
    dist.gather(tile, gathered, dst=dst)
    if rank != dst:
        return None
    full = torch.empty((height,) + tuple(tile.shape[1:]), dtype=tile.dtype, device=tile.device)
    for g in range(world):
        n = interleaved_shard(height, world, g)[3]
        full[g::world] = gathered[g][:n]
    return full


def gather_tiles(tile, rows, rank, world, dst=0):
    """tile: (max_rows, W, 4) tensor whose first rows[rank+1]-rows[rank] rows are valid.
    Returns the assembled (H, W, 4) framebuffer on rank dst, None elsewhere. One collective."""
    if world == 1:
        return tile[: rows[1] - rows[0]]
    gathered = [torch.empty_like(tile) for _ in range(world)] if rank == dst else None
    dist.gather(tile, gathered, dst=dst)
    if rank != dst:
        return None
    return torch.cat([gathered[g][: rows[g + 1] - rows[g]] for g in range(world)], dim=0)
