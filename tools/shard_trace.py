"""One rank's share of the 3840x2160 / 16 spp / depth 8 frame (shard 0 of 8), a warm-up frame and two measured ones: the program
rocprofv3 --kernel-trace is pointed at to see the anatomy of a small shard's frame (per-launch durations and the gaps between
launches).  PROBE_SHARDS / PROBE_RANK / VKRT_* options via the environment."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import vkrt_amd
from vkrt_amd import host_py
from vkrt_amd.flat_scene import make_push_constants
from vkrt_amd.renderer import Renderer
from vkrt_amd.sharding import make_shard
import atrium

W, H = int(os.environ.get("PROBE_W", 3840)), int(os.environ.get("PROBE_H", 2160))
N, RANK = int(os.environ.get("PROBE_SHARDS", 8)), int(os.environ.get("PROBE_RANK", 0))
flat, info = atrium.build_atrium(262144, seed=1)
cam = host_py.global_uniforms(width=W, height=H, **atrium.DEFAULT_CAMERA)
r = Renderer(flat, device=0, build="ploc")
shard = make_shard(W, H, N, RANK)
r.reserve(shard)
img = None
for f in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    img = r.pathtrace(make_push_constants(samples=16, depth=8, frame=f, lights_count=8), cam, W, H, seed=f, shard=shard, image=img)
    torch.cuda.synchronize()
    print(f"frame {f}: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
