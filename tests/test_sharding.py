"""Multi-GPU path on CPU: strip index math and the gather over torch.distributed (gloo, world 2)."""
import os
import sys

import numpy as np
import pytest

from vkrt_amd.sharding import gather_image, max_shard_rows, shard_row_indices

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("H", [1, 16, 17, 100, 1080, 2160])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_strips_partition_the_image(H, world):
    rows = [shard_row_indices(H, world, r) for r in range(world)]
    allr = np.concatenate(rows)
    assert sorted(allr.tolist()) == list(range(H))
    for r in rows:
        assert np.all(np.diff(r) > 0)
    assert max_shard_rows(H, world) == max(len(r) for r in rows)
    if world > 1 and H >= 16 * world:
        assert max(len(r) for r in rows) - min(len(r) for r in rows) <= 16


def _worker(rank, world, port, H, W, out_dir):
    import torch
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    import vkrt_amd  # noqa: F401
    from vkrt_amd.sharding import gather_image, shard_row_indices

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows = shard_row_indices(H, world, rank)
    # "render": pixel value encodes its global coordinates, as the kernel's seeds do
    y = torch.from_numpy(rows).float()[:, None].expand(len(rows), W)
    x = torch.arange(W).float()[None, :].expand(len(rows), W)
    local = torch.stack([x, y, x * 0 + rank, x * 0 + 1], dim=-1).contiguous()
    full = gather_image(local, H, world, rank)
    np.save(os.path.join(out_dir, f"full_{rank}.npy"), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_gather_reassembles_image(tmp_path):
    import torch.multiprocessing as mp

    world, H, W = 2, 70, 12
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, H, W, str(tmp_path)), nprocs=world, join=True)
    a = np.load(tmp_path / "full_0.npy")
    b = np.load(tmp_path / "full_1.npy")
    assert np.array_equal(a, b)
    yy, xx = np.mgrid[0:H, 0:W]
    assert np.array_equal(a[..., 0], xx) and np.array_equal(a[..., 1], yy)
    owner = (yy // 16) % world
    assert np.array_equal(a[..., 2], owner)


@pytest.mark.gpu
def test_overlapped_gather_on_rccl_single_rank():
    """The side-stream path of ImageGatherer with the real RCCL backend (one rank is all a 1-GPU box allows): the gather
    of frame f overlaps whatever the caller enqueues next, buffers are not reused before the previous gather finished, and
    after wait() the image is the local strips of the frame that was gathered."""
    import torch
    import torch.distributed as dist

    from vkrt_amd.sharding import ImageGatherer

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        H, W = 1080, 1920
        g = ImageGatherer(H, W, 1, 0, torch.device("cuda:0"))
        local = torch.zeros((H, W, 4), device="cuda:0")
        for f in range(5):
            local.fill_(float(f + 1))                     # "frame f"
            full = g.gather(local, overlap=True)
            local.mul_(0.5)                               # the caller goes on modifying its buffer at once
            big = torch.randn(4096, 4096, device="cuda:0") @ torch.randn(4096, 4096, device="cuda:0")  # next frame's work
            g.wait()
            assert torch.all(full == float(f + 1)).item(), f
        full = g.gather(local, overlap=False)
        assert torch.all(full == 2.5).item()
        del big
    finally:
        dist.destroy_process_group()
