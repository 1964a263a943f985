"""Strong-scaling rehearsal on one GPU (BASELINE config 4): the 3840x2160 / 16 spp / depth 8 atrium frame rendered whole, then
one rank's share of it (16-row strips dealt to N ranks).  efficiency = (whole-frame time / N) / shard time: what N-GPU strong
scaling can reach before the gather.  Frames go through vkrt_pathtrace_frames, PROBE_FRAMES_PER_CALL at a time (0 =
single-frame vkrt_pathtrace calls); options via environment (VKRT_WF_FRAMES_IN_FLIGHT, VKRT_WF_SUBFRAMES ...)."""
import json, os, sys, time
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
SPP, DEPTH = int(os.environ.get("PROBE_SPP", 16)), 8
flat, info = atrium.build_atrium(262144, seed=1)
cam = host_py.global_uniforms(width=W, height=H, **atrium.DEFAULT_CAMERA)
r = Renderer(flat, device=0, build=os.environ.get("BUILD", "ploc"))


PER_CALL = int(os.environ.get("PROBE_FRAMES_PER_CALL", 6))  # 0: single-frame vkrt_pathtrace calls (round 3's measurement)
NFRAMES = int(os.environ.get("PROBE_FRAMES", 6))


def frames(shard, n=NFRAMES):
    img = None
    r.reserve(shard)
    for f in range(1):
        img = r.pathtrace(make_push_constants(samples=SPP, depth=DEPTH, frame=f, lights_count=8), cam, W, H, seed=f, shard=shard, image=img)
    torch.cuda.synchronize(); r.reset_counters(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    f = 1
    while f < 1 + n:
        pc = make_push_constants(samples=SPP, depth=DEPTH, frame=f, lights_count=8)
        if PER_CALL == 0:
            img = r.pathtrace(pc, cam, W, H, seed=f, shard=shard, image=img); f += 1
        else:
            m = min(PER_CALL, 1 + n - f)
            img = r.pathtrace_frames(pc, cam, W, H, m, seed=f, shard=shard, image=img); f += m
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / n
    c = r.counters()
    return ms, (c["rays_closest"] + c["rays_shadow"]) / n


full_ms, full_rays = frames(make_shard(W, H, 1, 0))
out = {"env": {k: v for k, v in os.environ.items() if k.startswith("VKRT_")}, "entry_point": "vkrt_pathtrace" if PER_CALL == 0 else f"vkrt_pathtrace_frames x{PER_CALL}",
       "frames_timed": NFRAMES, "size": [W, H], "spp": SPP, "full_ms": round(full_ms, 2),
       "full_Mrays_s": round(full_rays / full_ms / 1e3, 1), "shards": {}}
for n in (2, 4, 8):
    for rank in ((0, n // 2 + 1) if n == 8 else (0,)):
        ms, rays = frames(make_shard(W, H, n, rank))
        out["shards"][f"{rank}/{n}"] = {"ms": round(ms, 2), "Mrays_s": round(rays / ms / 1e3, 1), "efficiency": round(full_ms / n / ms, 3),
                                        "rays_share": round(rays / full_rays * n, 3)}
print(json.dumps(out))
