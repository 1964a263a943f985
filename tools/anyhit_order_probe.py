"""Which child order do any-hit (shadow) walks want?  Front to back / farthest first (VKRT_OPT_WF_SHARE_FLAGS bits 1, 2), per scene
tessellation and per light placement (the reference's fallback lights: light 0 inside the building, lights 1-7 far outside)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import vkrt_amd
from vkrt_amd import abi, host_py
from vkrt_amd.flat_scene import make_push_constants
from vkrt_amd.renderer import Renderer
import atrium

W, H = 1920, 1080
cam = host_py.global_uniforms(width=W, height=H, **atrium.DEFAULT_CAMERA)
ROT = os.environ.get("PROBE_ROTATE")  # "ry,rx" degrees: the building turned (tools/atrium.py rotate_scene), pre-splitting automatic
for variant in (None, "nonuniform"):
    flat, _ = atrium.build_atrium(262144, seed=1, variant=variant)
    if ROT:
        camkw = atrium.rotate_scene(flat, dict(atrium.DEFAULT_CAMERA), *[float(v) for v in ROT.split(",")])
        cam = host_py.global_uniforms(width=W, height=H, **camkw)
    all_lights = flat.lights.copy()
    for name, lights in ((("all 8", all_lights),) if ROT else (("all 8", all_lights), ("inside light only", all_lights[:1]), ("outside lights only", all_lights[1:]))):
        flat.lights = lights.copy()
        for flags in ((17, 21, 25) if ROT else (1, 3, 5)):
            r = Renderer(flat, device=0, build="ploc", options={abi.VKRT_OPT_WF_SHARE_FLAGS: flags, abi.VKRT_OPT_SPLIT_BUDGET: -1 if ROT else 0})
            pc = make_push_constants(samples=4, depth=8, frame=0, lights_count=len(lights))
            img = None
            for it in range(3):
                r.reset_counters()
                img = r.pathtrace(pc, cam, W, H, seed=1, flags=abi.VKRT_TRACE_TIME_KERNELS, image=img)
                torch.cuda.synchronize()
                t = r.last_trace_timing()
            r.reset_counters()
            r.pathtrace(pc, cam, W, H, seed=1, flags=abi.VKRT_TRACE_COUNT_TRAVERSAL, image=img)
            c = r.counters()
            rays = c["rays_closest"] + c["rays_shadow"]
            print(json.dumps({"scene": variant or "uniform", "rotate": ROT or "", "resolved_order": r.get_option(abi.VKRT_INFO_ANYHIT_ORDER), "split": r.get_option(abi.VKRT_INFO_SPLIT_BUDGET), "lights": name, "flags": flags, "traverse_ms": round(t["traverse_ms"], 3), "nodes_per_ray": round(c["nodes_visited"] / rays, 2),
                              "tris_per_ray": round(c["tris_tested"] / rays, 2), "shadow_fraction": round(c["rays_shadow"] / rays, 3)}), flush=True)
            r.close()
