"""Image-space sharding across GPUs: strips of `strip_rows` rows dealt
round-robin to ranks, one gather per frame, analytic un-interleave.  Pure index math plus a
torch.distributed all_gather (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU
tests).  The reference is single-GPU (main.cpp:200); this replaces its one
vkCmdTraceRaysKHR(W,H,1) grid (hello_vulkan.cpp:1446) by N disjoint row sets."""
import numpy as np

from . import abi

STRIP_ROWS = 16


def make_shard(width, height, world_size, rank, strip_rows=STRIP_ROWS):
    if world_size <= 1:
        return abi.Shard(width, height, 0, 1, 0)
    return abi.Shard(width, height, strip_rows, world_size, rank)


def shard_row_indices(height, world_size, rank, strip_rows=STRIP_ROWS):
    """Global row index of every local row of `rank`, in local-buffer order."""
    if world_size <= 1:
        return np.arange(height, dtype=np.int64)
    rows = []
    nstrips = (height + strip_rows - 1) // strip_rows
    for s in range(rank, nstrips, world_size):
        y0 = s * strip_rows
        rows.append(np.arange(y0, min(y0 + strip_rows, height), dtype=np.int64))
    return np.concatenate(rows) if rows else np.zeros(0, np.int64)


def max_shard_rows(height, world_size, strip_rows=STRIP_ROWS):
    return max(len(shard_row_indices(height, world_size, r, strip_rows)) for r in range(world_size))


def gather_image(local, height, world_size, rank, strip_rows=STRIP_ROWS, group=None):
    """all_gather the per-rank strip buffers and un-interleave into the full [H, W, 4] image.
    `local` is a torch tensor [rows_r, W, 4] on any device; every rank returns the full image."""
    import torch
    import torch.distributed as dist

    if world_size <= 1:
        return local
    W = local.shape[1]
    cap = max_shard_rows(height, world_size, strip_rows)
    padded = torch.zeros((cap, W, 4), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    parts = [torch.empty_like(padded) for _ in range(world_size)]
    dist.all_gather(parts, padded, group=group)
    full = torch.empty((height, W, 4), dtype=local.dtype, device=local.device)
    for r in range(world_size):
        idx = torch.from_numpy(shard_row_indices(height, world_size, r, strip_rows)).to(local.device)
        full[idx] = parts[r][: idx.shape[0]]
    return full
