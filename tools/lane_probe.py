"""Lanes of vkrt_pathtrace_frames on one GPU: milliseconds per frame of one rank's share of the 3840x2160 / 16 spp / depth 8 frame
(shard PROBE_RANK of PROBE_SHARDS; PROBE_SHARDS=1 = the whole PROBE_W x PROBE_H frame) for a list of lane configurations
"F:S:framesPerCall" (frames in flight, sub-frames, frames per call; framesPerCall 0 = single-frame vkrt_pathtrace calls; with F > 1
the library runs one lane per frame in flight and S only applies to calls of one frame).  One JSON line per configuration; efficiency = (whole-frame time of
the FIRST configuration run with PROBE_SHARDS=1 ... ) is left to the reader: run the whole frame as its own configuration list."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import vkrt_amd
from vkrt_amd import abi, host_py
from vkrt_amd.flat_scene import make_push_constants
from vkrt_amd.renderer import Renderer
from vkrt_amd.sharding import make_shard
import atrium

W, H = int(os.environ.get("PROBE_W", 3840)), int(os.environ.get("PROBE_H", 2160))
N, RANK = int(os.environ.get("PROBE_SHARDS", 8)), int(os.environ.get("PROBE_RANK", 0))
SPP, DEPTH = int(os.environ.get("PROBE_SPP", 16)), 8
FRAMES = int(os.environ.get("PROBE_FRAMES", 8))
REPS = int(os.environ.get("PROBE_REPS", 2))
variant = os.environ.get("PROBE_VARIANT", "default")
flat, info = atrium.build_atrium(262144, seed=1, **({} if variant == "default" else {"variant": variant}))
cam = host_py.global_uniforms(width=W, height=H, **atrium.DEFAULT_CAMERA)
shard = make_shard(W, H, N, RANK)
configs = [tuple(int(v) for v in c.split(":")) for c in os.environ.get("PROBE_CONFIGS", "1:3:0,3:1:6,2:1:6").split(",")]
r = Renderer(flat, device=0, build=os.environ.get("BUILD", "ploc"))
for F, S, PER in configs:
    for k, v in ((abi.VKRT_OPT_WF_FRAMES_IN_FLIGHT, F), (abi.VKRT_OPT_WF_SUBFRAMES, S)):
        r.set_option(k, v)
    r.reserve(shard)
    img = None
    best = None
    for rep in range(REPS + 1):  # the first repetition warms up
        torch.cuda.synchronize(); r.reset_counters(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        f = 1
        while f < 1 + FRAMES:
            n = 1 if PER == 0 else min(PER, 1 + FRAMES - f)
            pc = make_push_constants(samples=SPP, depth=DEPTH, frame=f, lights_count=8)
            if PER == 0:
                img = r.pathtrace(pc, cam, W, H, seed=f, shard=shard, image=img)
            else:
                img = r.pathtrace_frames(pc, cam, W, H, n, seed=f, shard=shard, image=img)
            f += n
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / FRAMES
        if rep > 0:
            best = ms if best is None else min(best, ms)
    c = r.counters()
    rays = (c["rays_closest"] + c["rays_shadow"]) / FRAMES
    print(json.dumps({"config": f"{F}:{S}:{PER}", "size": [W, H], "shard": [RANK, N], "ms_per_frame": round(best, 3), "Mrays_s": round(rays / best / 1e3, 1),
                      "faults": c["traversal_faults"]}), flush=True)
r.close()
