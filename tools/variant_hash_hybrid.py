"""Hash of the hybrid-mode planes (ray-cast G-buffer + shadows/AO/GI accumulation, 2 frames) of the bench scene under the
current VKRT_* environment; run under different switches and compare (one-off invariance check)."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import vkrt_amd, atrium, camera_np
from vkrt_amd.flat_scene import make_push_constants, uniforms_from_matrices
from vkrt_amd.renderer import Renderer
W, H = 1920, 1080
flat, _ = atrium.build_atrium(262144, seed=1)
cam = uniforms_from_matrices(*camera_np.global_uniforms(width=W, height=H, **atrium.DEFAULT_CAMERA))
r = Renderer(flat, device=0, build=os.environ.get("BUILD", "sah"))
g = r.gbuffer_raycast(cam, W, H, lights_count=len(flat.lights))
acc = None
for f in range(2):
    pc = make_push_constants(samples=1, depth=8, frame=f, lights_count=len(flat.lights))
    pc.useShadows, pc.useAO, pc.useGI = 1, 1, 1
    acc = r.hybrid_trace(pc, cam, W, H, g, seed=f, accum=acc)
h = hashlib.sha256()
for k in ("color", "position", "normal", "roughMetal"):
    h.update(g[k].cpu().numpy().tobytes())
h.update(acc.cpu().numpy().tobytes())
print("HASH", h.hexdigest()[:16], {k: v for k, v in os.environ.items() if k.startswith(("VKRT_", "BUILD"))}, flush=True)
