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


class ImageGatherer:
    """Per-frame gather of the rank-local strip buffers into the full image, with everything that does not change from
    frame to frame prepared once: the padded send buffer, the receive buffer, the full image and the row permutation
    live on the device, so a frame costs one copy, one all_gather and one indexed copy, all enqueued asynchronously
    (no host synchronisation, no per-frame allocation or host-to-device index upload)."""

    def __init__(self, height, width, world_size, rank, device, dtype=None, strip_rows=STRIP_ROWS, group=None):
        import torch

        self.world, self.rank, self.group = world_size, rank, group
        dtype = dtype or torch.float32
        self.cap = max_shard_rows(height, world_size, strip_rows)
        self.rows_local = len(shard_row_indices(height, world_size, rank, strip_rows))
        self.padded = torch.zeros((self.cap, width, 4), dtype=dtype, device=device)
        self.recv = torch.empty((world_size, self.cap, width, 4), dtype=dtype, device=device)
        self.full = torch.empty((height, width, 4), dtype=dtype, device=device)
        src, dst = [], []
        for r in range(world_size):
            g = shard_row_indices(height, world_size, r, strip_rows)
            src.append(r * self.cap + np.arange(len(g), dtype=np.int64))
            dst.append(g)
        self.src = torch.from_numpy(np.concatenate(src)).to(device)
        self.dst = torch.from_numpy(np.concatenate(dst)).to(device)
        self._side = self._ready = self._done = None
        self._pending = False

    def gather(self, local, overlap=False):
        """Gathers this frame.  overlap=False: the returned image is ordered on the current stream like any torch op.
        overlap=True (device tensors): only the copy of the local strips is ordered on the current stream; the collective and
        the un-interleave run on a side stream, so the next frame's kernels overlap them.  The image is complete after
        `wait()` (or any device-wide synchronisation); the next call waits for the previous gather before reusing buffers."""
        import torch
        import torch.distributed as dist

        on_device = self.padded.is_cuda
        if on_device and self._side is None:
            self._side = torch.cuda.Stream(device=self.padded.device)
            self._ready = torch.cuda.Event()
            self._done = torch.cuda.Event()
        if on_device:
            main = torch.cuda.current_stream(self.padded.device)
            if self._pending:
                main.wait_event(self._done)  # the previous gather still reads `padded` / writes `full`
            self.padded[: self.rows_local].copy_(local, non_blocking=True)
            self._ready.record(main)
            with torch.cuda.stream(self._side):
                self._side.wait_event(self._ready)
                self._collect(dist)
                self._done.record(self._side)
            self._pending = True
            if not overlap:
                self.wait()
        else:
            self.padded[: self.rows_local].copy_(local)
            self._collect(dist)
        return self.full

    def wait(self):
        """Orders the current stream after the last gather."""
        import torch

        if self._pending:
            torch.cuda.current_stream(self.padded.device).wait_event(self._done)
            self._pending = False

    def _collect(self, dist):
        if dist.get_backend(self.group) == "nccl":
            dist.all_gather_into_tensor(self.recv, self.padded, group=self.group)
        else:
            parts = list(self.recv.unbind(0))
            dist.all_gather(parts, self.padded, group=self.group)
        flat = self.recv.view(self.world * self.cap, *self.recv.shape[2:])
        self.full.index_copy_(0, self.dst, flat.index_select(0, self.src))


_gatherers = {}


def gather_image(local, height, world_size, rank, strip_rows=STRIP_ROWS, group=None, overlap=False):
    """all_gather the per-rank strip buffers and un-interleave into the full [H, W, 4] image.
    `local` is a torch tensor [rows_r, W, 4] on any device; every rank returns the full image (a buffer that the next
    call with the same geometry overwrites).  overlap=True: see ImageGatherer.gather."""
    if world_size <= 1:
        return local
    key = (height, local.shape[1], world_size, rank, str(local.device), local.dtype, strip_rows, id(group))
    g = _gatherers.get(key)
    if g is None:
        g = _gatherers[key] = ImageGatherer(height, local.shape[1], world_size, rank, local.device, local.dtype, strip_rows, group)
    return g.gather(local, overlap=overlap)
