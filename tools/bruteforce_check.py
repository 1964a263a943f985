"""GPU images (SAH and LBVH builds, default pipeline) against the oracle's BRUTE-FORCE walk -- no tree on the checking side --
full images of two small atriums, two progressive frames each.  Run on the GPU box (the brute-force leg takes ~30 s on 16
threads).  Last run: 0 differing pixels of 146,944 (profiles/r01_experiments.md #43)."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import vkrt_amd, atrium, camera_np, oracle_py
from vkrt_amd.flat_scene import make_push_constants, uniforms_from_matrices
from vkrt_amd.renderer import Renderer
for tris, seed, (W, H), spp, depth in ((20000, 3, (384, 216), 3, 8), (8000, 11, (320, 200), 4, 6)):
    flat, _ = atrium.build_atrium(tris, seed=seed)
    cam = uniforms_from_matrices(*camera_np.global_uniforms(width=W, height=H, **atrium.DEFAULT_CAMERA))
    orc = oracle_py.OracleScene(flat)
    for build in ("sah", "lbvh"):
        r = Renderer(flat, device=0, build=build)
        img = None
        for f in range(2):
            img = r.pathtrace(make_push_constants(samples=spp, depth=depth, frame=f, lights_count=len(flat.lights)), cam, W, H, seed=5 + f, image=img)
        g = img.cpu().numpy()
        r.close()
        if build == "sah":
            t0 = time.time()
            bf = np.zeros((H, W, 4), np.float32); bv = np.zeros((H, W, 4), np.float32)
            for f in range(2):
                pc = make_push_constants(samples=spp, depth=depth, frame=f, lights_count=len(flat.lights))
                orc.render(pc, cam, W, H, seed=5 + f, image=bf, use_bvh=False, threads=16)
                orc.render(pc, cam, W, H, seed=5 + f, image=bv, threads=16)
            print("oracle brute vs bvh differing pixels:", int(np.any(bf.view(np.uint32) != bv.view(np.uint32), axis=-1).sum()), "cpu s", round(time.time() - t0, 1), flush=True)
        print(tris, build, "GPU vs brute force differing pixels:", int(np.any(g.view(np.uint32) != bf.view(np.uint32), axis=-1).sum()), "of", W * H, flush=True)
