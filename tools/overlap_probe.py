"""How much do independent frames gain from running concurrently?  Two scene handles (two working sets) on two HIP streams
render the same progressive frames; compared with one handle rendering them back to back.  Decides whether the library should
keep more than one frame in flight (frames only meet in the final accumulation, raytrace.rgen:135-141)."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import vkrt_amd
from vkrt_amd import abi
from vkrt_amd.flat_scene import make_push_constants, uniforms_from_matrices
from vkrt_amd.renderer import Renderer
from vkrt_amd.sharding import make_shard
import atrium, camera_np

W, H = int(os.environ.get("PROBE_W", 1920)), int(os.environ.get("PROBE_H", 1080))
SPP, FRAMES = int(os.environ.get("PROBE_SPP", 8)), int(os.environ.get("PROBE_FRAMES", 6))
flat, info = atrium.build_atrium(262144, seed=1)
cam = uniforms_from_matrices(*camera_np.global_uniforms(width=W, height=H, **atrium.DEFAULT_CAMERA))
NSH = int(os.environ.get("PROBE_SHARDS", 1))  # > 1: every handle renders shard 0 of this many (a small rank's share of the frame)
SHARD = make_shard(W, H, NSH, 0)


def run(handles, subframes):
    rs = [Renderer(flat, device=0, build="ploc", options={abi.VKRT_OPT_WF_SUBFRAMES: subframes}) for _ in range(handles)]
    streams = [torch.cuda.Stream() for _ in range(handles)]
    imgs = [torch.zeros((rs[0].shard_rows(SHARD), W, 4), dtype=torch.float32, device="cuda:0") for _ in range(handles)]
    for r in rs:
        r.reserve(SHARD)
    best = None
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f in range(FRAMES):
            k = f % handles
            pc = make_push_constants(samples=SPP, depth=8, frame=f, lights_count=8)
            with torch.cuda.stream(streams[k]):
                rs[k].pathtrace(pc, cam, W, H, seed=f, image=imgs[k], stream=streams[k], shard=SHARD)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / FRAMES
        best = ms if best is None else min(best, ms)
    for r in rs:
        r.close()
    return best


out = {}
configs = [tuple(int(v) for v in c.split(":")) for c in os.environ.get("PROBE_CONFIGS", "1:1,1:2,1:3,2:1,2:2,3:1,3:2,4:1").split(",")]
for handles, sub in configs:
    out[f"handles{handles}_sub{sub}"] = round(run(handles, sub), 3)
    print(json.dumps(out), flush=True)
