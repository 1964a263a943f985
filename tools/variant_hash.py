"""Image hash of N progressive frames of the bench scene under the current VKRT_* environment (one-off invariance checks:
run it under different switches and compare).  usage: variant_hash.py [W H spp depth frames triangles]"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import vkrt_amd, atrium, camera_np
from vkrt_amd.flat_scene import make_push_constants, uniforms_from_matrices
from vkrt_amd.renderer import Renderer
a = [int(x) for x in sys.argv[1:]] + [1920, 1080, 4, 8, 2, 262144][len(sys.argv) - 1:]
W, H, spp, depth, frames, tris = a[:6]
flat, _ = atrium.build_atrium(tris, seed=1)
cam = uniforms_from_matrices(*camera_np.global_uniforms(width=W, height=H, **atrium.DEFAULT_CAMERA))
r = Renderer(flat, device=0, build=os.environ.get("BUILD", "sah"))
img = None
for f in range(frames):
    img = r.pathtrace(make_push_constants(samples=spp, depth=depth, frame=f, lights_count=len(flat.lights)), cam, W, H, seed=f, image=img)
h = img.cpu().numpy()
c = r.counters()
print("HASH", hashlib.sha256(h.tobytes()).hexdigest()[:16], "rays", c["rays_closest"] + c["rays_shadow"], {k: v for k, v in os.environ.items() if k.startswith(("VKRT_", "BUILD"))}, flush=True)
np.save(os.path.join(ROOT, "gpurun_out", "vh_%s.npy" % (os.environ.get("TAG", "x"))), h) if os.environ.get("SAVE") else None
