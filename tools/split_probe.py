"""Triangle pre-splitting (VKRT_OPT_SPLIT_BUDGET) on the two tessellations of the atrium: ray rate, nodes / triangles per ray, tree
size and build time per budget.  PROBE_BUDGETS="0,10,30", PROBE_VARIANTS="default,nonuniform", BUILD=ploc|lbvh, 1080p 16 spp depth 8."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import vkrt_amd
from vkrt_amd import abi, host_py
from vkrt_amd.flat_scene import make_push_constants
from vkrt_amd.renderer import Renderer
import atrium

W, H = int(os.environ.get("PROBE_W", 1920)), int(os.environ.get("PROBE_H", 1080))
SPP = int(os.environ.get("PROBE_SPP", 16))
FRAMES = int(os.environ.get("PROBE_FRAMES", 3))
kind = os.environ.get("BUILD", "ploc")
extra = {int(k): int(v) for k, v in (kv.split("=") for kv in os.environ.get("PROBE_OPTS", "").split(",") if kv)}
for variant in os.environ.get("PROBE_VARIANTS", "default;nonuniform").split(";"):
    flat, info = atrium.build_atrium(262144, seed=1, **({} if variant == "default" else {"variant": variant}))
    camkw = dict(atrium.DEFAULT_CAMERA)
    if os.environ.get("PROBE_ROTATE"):
        # the whole building turned about y and tilted about x (degrees "ry,rx"): no large triangle is aligned with the axes any more, which
        # is where reference splitting is known to pay (the boxes of room-sized diagonal triangles are mostly empty).  Lights stay put.
        ry, rx = (float(v) for v in (os.environ["PROBE_ROTATE"].split(",") + ["0"])[:2])
        camkw = atrium.rotate_scene(flat, camkw, ry, rx)
    cam = host_py.global_uniforms(width=W, height=H, **camkw)
    r = Renderer(flat, device=0, build=None)
    for budget in [int(b) for b in os.environ.get("PROBE_BUDGETS", "0,10,20,30,50").split(",")]:
        r.set_option(abi.VKRT_OPT_SPLIT_BUDGET, budget)
        for k, v in extra.items():
            r.set_option(k, v)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r.build(kind)
        torch.cuda.synchronize(); build_ms = (time.perf_counter() - t0) * 1e3
        a = r.accel_info()
        img = None
        best = None
        for rep in range(2):
            torch.cuda.synchronize(); r.reset_counters(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for f in range(1, 1 + FRAMES):
                img = r.pathtrace(make_push_constants(samples=SPP, depth=8, frame=f, lights_count=8), cam, W, H, seed=f, image=img)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3 / FRAMES
            best = ms if best is None else min(best, ms)
        c = r.counters()
        rays = (c["rays_closest"] + c["rays_shadow"]) / FRAMES
        r.reset_counters()
        r.pathtrace(make_push_constants(samples=4, depth=8, frame=1, lights_count=8), cam, W, H, seed=1, flags=abi.VKRT_TRACE_COUNT_TRAVERSAL, image=img)
        w = r.counters()
        wr = w["rays_closest"] + w["rays_shadow"]
        print(json.dumps({"variant": variant, "rotate": os.environ.get("PROBE_ROTATE", ""), "build": kind, "budget": budget, "opts": extra, "ms_per_frame": round(best, 3), "Mrays_s": round(rays / best / 1e3, 1),
                          "nodes_per_ray": round(w["nodes_visited"] / wr, 2), "tris_per_ray": round(w["tris_tested"] / wr, 2),
                          "node_lane_eff": round(w["nodes_visited"] / max(64 * w["wave_node_steps"], 1), 3), "tri_lane_eff": round(w["tris_tested"] / max(64 * w["wave_tri_steps"], 1), 3),
                          "wave_node_steps_per_kray": round(1e3 * w["wave_node_steps"] / wr, 2), "wave_tri_steps_per_kray": round(1e3 * w["wave_tri_steps"] / wr, 2),
                          "shadow_frac": round(w["rays_shadow"] / wr, 3), "hits_per_closest": round(w["hits"] / max(w["rays_closest"], 1), 3), "rays_M": round(rays / 1e6, 1),
                          "triangles": a["triangle_count"], "references": a["reference_count"], "nodes": a["node_count"], "depth": a["max_depth"],
                          "sah": round(a["sah_cost"], 2), "build_ms": round(build_ms, 1), "anyhit_order": r.get_option(abi.VKRT_INFO_ANYHIT_ORDER),
                          "faults": c["traversal_faults"]}), flush=True)
    r.close()
