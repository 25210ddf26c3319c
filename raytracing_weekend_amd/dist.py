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
