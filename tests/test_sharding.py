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


# ---- C++ host side of the gather (host/strip_gather.h): same strip deal, row map and un-interleave ---------------------
def test_cpp_strip_layout_matches_the_abi_shard_math():
    """StripLayout (the C++ host's row map) against vkrt_shard_rows (the ABI) and sharding.py, incl. ragged heights."""
    import ctypes as C

    from vkrt_amd import abi, host_py
    from vkrt_amd.renderer import load_library
    from vkrt_amd.sharding import max_shard_rows, shard_row_indices

    lib = load_library()
    for H in (1, 15, 16, 17, 70, 200, 1080, 2160):
        for world in (1, 2, 3, 4, 8):
            for rank in range(world):
                rows = shard_row_indices(H, world, rank) if world > 1 else np.arange(H)
                assert host_py.strip_rows(H, world, rank) == len(rows)
                if world > 1:
                    sh = abi.Shard(64, H, 16, world, rank)
                    assert host_py.strip_rows(H, world, rank) == int(lib.vkrt_shard_rows(C.byref(sh)))
                for local in (0, len(rows) // 2, len(rows) - 1):
                    if len(rows):
                        assert host_py.strip_source(H, world, int(rows[local])) == (rank, local)
            if world > 1:
                assert host_py.strip_rows(H, world, 0) == max_shard_rows(H, world)  # rank 0's buffer is the padded size


def _cpp_map_worker(rank, world, port, H, W, out_dir):
    """all_gather of padded strip buffers over gloo, un-interleaved with the C++ host's row map (what StripGather's kernel does)."""
    import torch
    import torch.distributed as dist

    from vkrt_amd import host_py
    from vkrt_amd.sharding import shard_row_indices

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows = shard_row_indices(H, world, rank)
    cap = host_py.strip_rows(H, world, 0)
    send = torch.zeros((cap, W, 4))
    y = torch.from_numpy(rows).float()[:, None].expand(len(rows), W)
    x = torch.arange(W).float()[None, :].expand(len(rows), W)
    send[: len(rows)] = torch.stack([x, y, x * 0 + rank, x * 0 + 1], dim=-1)
    recv = [torch.zeros_like(send) for _ in range(world)]
    dist.all_gather(recv, send)
    recv = torch.stack(recv)
    full = torch.zeros((H, W, 4))
    for gy in range(H):
        r, l = host_py.strip_source(H, world, gy)
        full[gy] = recv[r, l]
    np.save(os.path.join(out_dir, f"cpp_full_{rank}.npy"), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_gather_with_the_cpp_row_map(tmp_path):
    import torch.multiprocessing as mp

    world, H, W = 2, 70, 12
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_cpp_map_worker, args=(world, port, H, W, str(tmp_path)), nprocs=world, join=True)
    a = np.load(tmp_path / "cpp_full_0.npy")
    assert np.array_equal(a, np.load(tmp_path / "cpp_full_1.npy"))
    yy, xx = np.mgrid[0:H, 0:W]
    assert np.array_equal(a[..., 0], xx) and np.array_equal(a[..., 1], yy) and np.array_equal(a[..., 2], (yy // 16) % world)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 8])
def test_cpp_unpack_kernel_reassembles_sharded_render(world, cornell_flat):
    """k_unpack_strips (the HIP un-interleave of the C++ host) on real strip buffers: N shards rendered by vkrt_pathtrace,
    stacked as an all-gather would deliver them, unpacked, equal the unsharded image bit for bit."""
    import torch

    from conftest import default_camera
    from vkrt_amd import host_py
    from vkrt_amd.flat_scene import make_push_constants
    from vkrt_amd.renderer import Renderer
    from vkrt_amd.sharding import make_shard

    W, H = 256, 200
    cam = default_camera(W, H)
    pc = make_push_constants(samples=2, depth=3, frame=0, lights_count=1)
    r = Renderer(cornell_flat, device=0, build="sah")
    full = r.pathtrace(pc, cam, W, H, seed=9)
    cap = host_py.strip_rows(H, world, 0)
    gathered = torch.zeros((world, cap, W, 4), device="cuda:0")
    for rank in range(world):
        part = r.pathtrace(pc, cam, W, H, seed=9, shard=make_shard(W, H, world, rank))
        gathered[rank, : part.shape[0]] = part
    out = host_py.unpack_strips(gathered, torch.empty_like(full), world)
    torch.cuda.synchronize()
    r.close()
    assert torch.equal(out, full)


@pytest.mark.gpu
def test_vkrt_render_ranks_goes_through_rccl(tmp_path):
    """`vkrt_render --ranks 1`: the C++ host's multi-process path (fork before any GPU work, RCCL id file, ncclCommInitRank,
    ncclAllGather, unpack kernel) with the one rank a 1-GPU box allows; the gathered image equals the plain single-process one."""
    import json
    import subprocess
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import atrium
    import gltf_export
    import imgdiff

    flat, _ = atrium.build_atrium(4000, seed=5, with_textures=False)
    gltf_export.export_gltf(flat, str(tmp_path / "scene.gltf"))
    for name in ("a", "b"):
        cfg = {"scenes": ["scene.gltf"], "scene": 0, "vsync": False, "width": 160, "height": 90, "samples": 2, "depth": 3, "frames": 2,
               "camera": {"eye": [-12.5, 4.2, 0.6], "center": [6.0, 3.6, -0.4], "up": [0, 1, 0], "fov": 60}, "output": str(tmp_path / name)}
        (tmp_path / f"{name}.json").write_text(json.dumps(cfg))
    exe = os.path.join(ROOT, "vk-raytracing-engine_amd", "vkrt_render")
    p = subprocess.run([exe, "--config", str(tmp_path / "a.json")], capture_output=True, text=True, timeout=180)
    assert p.returncode == 0, p.stderr
    q = subprocess.run([exe, "--config", str(tmp_path / "b.json"), "--ranks", "1"], capture_output=True, text=True, timeout=180)
    assert q.returncode == 0, q.stderr + q.stdout
    assert "RCCL all-gather" in q.stdout
    a, _ = imgdiff.read_image(str(tmp_path / "a.pfm"))
    b, _ = imgdiff.read_image(str(tmp_path / "b.pfm"))
    assert np.array_equal(a, b)
