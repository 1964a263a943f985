"""Times the hybrid-mode kernels (BASELINE config 5 stand-in: atrium, 1080p, shadows + AO + GI)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import vkrt_amd
from vkrt_amd.flat_scene import make_push_constants, uniforms_from_matrices
from vkrt_amd.renderer import Renderer
import atrium, camera_np
W, H = 1920, 1080
flat, info = atrium.build_atrium(262144, seed=1)
cam = uniforms_from_matrices(*camera_np.global_uniforms(width=W, height=H, **atrium.DEFAULT_CAMERA))
r = Renderer(flat, device=0, build=os.environ.get("BUILD", "ploc"))
def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): out = fn()
    e1.record(); torch.cuda.synchronize()
    return out, e0.elapsed_time(e1) / n
g, ms_g = timed(lambda: r.gbuffer_raycast(cam, W, H, lights_count=8))
print(json.dumps({"kernel": "k_gbuffer", "texture_lod": "implicit (mips, anisotropy 4)", "ms": round(ms_g, 3), "Mpixels_s": round(W * H / ms_g / 1e3, 1)}))
from vkrt_amd import abi
r.set_option(abi.VKRT_OPT_GBUFFER_MIPS, 0)
_, ms_0 = timed(lambda: r.gbuffer_raycast(cam, W, H, lights_count=8))
r.set_option(abi.VKRT_OPT_GBUFFER_MIPS, 1)
print(json.dumps({"kernel": "k_gbuffer", "texture_lod": "0", "ms": round(ms_0, 3), "Mpixels_s": round(W * H / ms_0 / 1e3, 1)}))
for sh, ao, gi in ((1, 0, 0), (1, 1, 0), (1, 1, 1)):
    pc = make_push_constants(samples=1, depth=8, frame=0, lights_count=8)
    pc.useShadows, pc.useAO, pc.useGI = sh, ao, gi
    r.reset_counters(); r.hybrid_trace(pc, cam, W, H, g, seed=1); torch.cuda.synchronize()
    c = r.counters(); rays = c["rays_closest"] + c["rays_shadow"]
    _, ms = timed(lambda: r.hybrid_trace(pc, cam, W, H, g, seed=1))
    print(json.dumps({"kernel": "k_hybrid", "shadows": sh, "ao": ao, "gi": gi, "ms": round(ms, 3), "rays": rays, "Mrays_s": round(rays / ms / 1e3, 1)}))
_, ms_p = timed(lambda: r.post(g["color"], g["position"], rt_mode=0))
print(json.dumps({"kernel": "k_post", "ms": round(ms_p, 3), "GB_s": round(W * H * 48 / ms_p / 1e6, 1)}))
