"""Micro-probe of the wavefront traversal kernel: times each traversal launch of small frames."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import vkrt_amd
from vkrt_amd import abi
from vkrt_amd.flat_scene import make_push_constants, uniforms_from_matrices
from vkrt_amd.renderer import Renderer
import atrium, camera_np
W, H = int(os.environ.get("PROBE_W", 1920)), int(os.environ.get("PROBE_H", 1080))
flat, info = atrium.build_atrium(262144, seed=1)
cam = uniforms_from_matrices(*camera_np.global_uniforms(width=W, height=H, **atrium.DEFAULT_CAMERA))
r = Renderer(flat, device=0, build=os.environ.get("BUILD", "sah"))
img = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
for spp, depth in (((4, 8),) if os.environ.get("PROBE_QUICK") else ((1, 1), (1, 8), (4, 8))):
    pc = make_push_constants(samples=spp, depth=depth, frame=0, lights_count=8)
    for it in range(3):
        r.reset_counters()
        r.pathtrace(pc, cam, W, H, seed=1, flags=abi.VKRT_TRACE_TIME_KERNELS, image=img)
        torch.cuda.synchronize()
        t = r.last_trace_timing(); c = r.counters()
    rays = c["rays_closest"] + c["rays_shadow"]
    free = []
    for it in range(4):  # untimed kernels: the sub-frame pipeline (VKRT_WF_SUBFRAMES) is active
        r.pathtrace(pc, cam, W, H, seed=1, flags=0, image=img)
        torch.cuda.synchronize()
        free.append(r.last_trace_ms())
    frame_ms = min(free[1:])
    r.reset_counters()
    r.pathtrace(pc, cam, W, H, seed=1, flags=abi.VKRT_TRACE_COUNT_TRAVERSAL, image=img)
    cc = r.counters()
    work = {"nodes_per_ray": round(cc["nodes_visited"] / rays, 2), "tris_per_ray": round(cc["tris_tested"] / rays, 2),
            "shadow_frac": round(cc["rays_shadow"] / rays, 3),
            "node_lane_eff": round(cc["nodes_visited"] / max(64 * cc["wave_node_steps"], 1), 3),
            "tri_lane_eff": round(cc["tris_tested"] / max(64 * cc["wave_tri_steps"], 1), 3),
            "wave_node_steps": cc["wave_node_steps"], "wave_tri_steps": cc["wave_tri_steps"]}
    print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("VKRT_")}, "spp": spp, "depth": depth, "rays": rays,
                      "frame_ms": round(frame_ms, 3), "Mrays_s_frame": round(rays / frame_ms / 1e3, 1), "total_ms": round(t["total_ms"], 3), "traverse_ms": round(t["traverse_ms"], 3), "launches": t["traverse_launches"],
                      "Mrays_s_total": round(rays / t["total_ms"] / 1e3, 1), "Mrays_s_traverse": round(rays / max(t["traverse_ms"], 1e-9) / 1e3, 1), "mode": t["mode"], **work}))
