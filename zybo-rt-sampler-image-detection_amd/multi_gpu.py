"""Direction-sharded heat-maps over the GPUs of one node (no counterpart in the reference,
which is a single process).

Every rank holds the full steering tables and the same frames, steers them to its own contiguous shard
[lo, hi) of the flat direction grid (bf_das_device's dir_begin/dir_end) and ONE all-gather per batch assembles the
full maps.  The only collective on the data path is that gather (RCCL over xGMI when the backend is "nccl"; the CPU
tests run the same code over gloo with the oracle as the per-shard compute)."""
import numpy as np


def shard_capacity(n_dirs, world):
    """Directions per rank, padded so that every rank contributes an equally sized block to the all-gather."""
    return (n_dirs + world - 1) // world


def shard_range(n_dirs, world, rank):
    """Contiguous [lo, hi) of flat directions owned by `rank` (the last ranks may own fewer, or none)."""
    cap = shard_capacity(n_dirs, world)
    lo = min(rank * cap, n_dirs)
    return lo, min(lo + cap, n_dirs)


def assemble(gathered, n_dirs):
    """[world, frames, cap] (all-gather output) -> [frames, n_dirs] full maps, padding columns dropped."""
    world, frames, cap = gathered.shape
    full = gathered.permute(1, 0, 2).reshape(frames, world * cap)
    return full[:, :n_dirs]


def sharded_heatmaps(compute_shard, frames, n_dirs, group=None, device=None):
    """Run `compute_shard(lo, hi) -> float32 tensor [frames, hi-lo]` on every rank's shard and all-gather.

    Returns float32 [frames, n_dirs] on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_range(n_dirs, world, rank)
    cap = shard_capacity(n_dirs, world)
    part = torch.zeros((frames, cap), dtype=torch.float32, device=device)
    if hi > lo:
        part[:, : hi - lo] = compute_shard(lo, hi)
    if world == 1:
        return part[:, :n_dirs]
    gathered = torch.empty((world * frames, cap), dtype=torch.float32, device=device)   # rank-major concatenation
    dist.all_gather_into_tensor(gathered, part, group=group)
    return assemble(gathered.view(world, frames, cap), n_dirs)
