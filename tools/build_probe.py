"""Acceleration-structure builders side by side on the bench scene: build time, tree statistics and trace rate (4 spp frame)."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import vkrt_amd
from vkrt_amd.flat_scene import make_push_constants, uniforms_from_matrices
from vkrt_amd.renderer import Renderer
import atrium, camera_np

W, H = 1920, 1080
flat, info = atrium.build_atrium(int(os.environ.get("PROBE_TRIS", 262144)), seed=1)
cam = uniforms_from_matrices(*camera_np.global_uniforms(width=W, height=H, **atrium.DEFAULT_CAMERA))
r = Renderer(flat, device=0, build=None)
img = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
for kind in os.environ.get("PROBE_BUILDS", "lbvh,ploc,sah").split(","):
    best = None
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r.build(kind)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        best = ms if best is None else min(best, ms)
    a = r.accel_info()
    pc = make_push_constants(samples=4, depth=8, frame=0, lights_count=8)
    times = []
    for it in range(4):
        r.reset_counters()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r.pathtrace(pc, cam, W, H, seed=1, image=img)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    c = r.counters()
    rays = c["rays_closest"] + c["rays_shadow"]
    print(json.dumps({"build": kind, "env": {k: v for k, v in os.environ.items() if k.startswith("VKRT_")}, "build_ms": round(best, 2),
                      "nodes": a["node_count"], "depth": a["max_depth"], "sah_cost": round(a["sah_cost"], 3), "frame_ms": round(min(times[1:]), 3),
                      "Mrays_s": round(rays / min(times[1:]) / 1e3, 1), "faults": c["traversal_faults"]}), flush=True)
