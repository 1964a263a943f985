"""Does the hit shader care where a triangle's three vertex records live?  Same atrium, same pixels, three vertex layouts fed through the
unchanged library: as generated (indexed grids), de-indexed (every triangle's three 48-byte records adjacent: what a per-triangle
shading record would look like), and the indexed vertices in a random order (the worst case).  Prints one JSON line per layout.
PROBE_VARIANTS="default;nonuniform", BUILD=ploc, 1080p 16 spp depth 8, 6 frames per call."""
import dataclasses, hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
import vkrt_amd
from vkrt_amd import host_py
from vkrt_amd.flat_scene import make_push_constants
from vkrt_amd.renderer import Renderer
import atrium

W, H = int(os.environ.get("PROBE_W", 1920)), int(os.environ.get("PROBE_H", 1080))
SPP = int(os.environ.get("PROBE_SPP", 16))
FRAMES = int(os.environ.get("PROBE_FRAMES", 6))
kind = os.environ.get("BUILD", "ploc")


def relayout(flat, how):
    if how == "indexed":
        return flat
    pos, nrm, tan, uv, idx, pms = [], [], [], [], [], flat.prim_meshes.copy()
    vo = io = 0
    rng = np.random.default_rng(7)
    for k, pm in enumerate(flat.prim_meshes):
        i = flat.indices[pm["firstIndex"]: pm["firstIndex"] + pm["indexCount"]].astype(np.int64)
        v0, vn = int(pm["vertexOffset"]), int(pm["vertexCount"])
        if how == "deindexed":
            g = v0 + i
            new_i = np.arange(len(i), dtype=np.uint32)
        else:  # "shuffled": same vertices, random order
            perm = rng.permutation(vn)  # new position p holds old vertex perm[p]
            inv = np.empty(vn, np.int64); inv[perm] = np.arange(vn)
            g = v0 + perm
            new_i = inv[i].astype(np.uint32)
        pos.append(flat.positions[g]); nrm.append(flat.normals[g]); tan.append(flat.tangents[g]); uv.append(flat.texcoords0[g]); idx.append(new_i)
        pms[k]["firstIndex"], pms[k]["vertexOffset"], pms[k]["vertexCount"] = io, vo, len(g)
        vo += len(g); io += len(new_i)
    return dataclasses.replace(flat, positions=np.concatenate(pos), normals=np.concatenate(nrm), tangents=np.concatenate(tan),
                               texcoords0=np.concatenate(uv), indices=np.concatenate(idx), prim_meshes=pms)


for variant in os.environ.get("PROBE_VARIANTS", "default").split(";"):
    base, info = atrium.build_atrium(262144, seed=1, **({} if variant == "default" else {"variant": variant}))
    cam = host_py.global_uniforms(width=W, height=H, **atrium.DEFAULT_CAMERA)
    for how in os.environ.get("PROBE_LAYOUTS", "indexed;deindexed;shuffled").split(";"):
        flat = relayout(base, how)
        r = Renderer(flat, device=0, build=kind)
        best, img = None, None
        for rep in range(3):
            img = None
            torch.cuda.synchronize(); t0 = time.perf_counter()
            img = r.pathtrace_frames(make_push_constants(samples=SPP, depth=8, frame=0, lights_count=8), cam, W, H, FRAMES, seed=1, image=img)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3 / FRAMES
            best = ms if best is None else min(best, ms)
        print(json.dumps({"variant": variant, "layout": how, "vertices": int(flat.positions.shape[0]), "vertex_MB": round(flat.positions.shape[0] * 48 / 1e6, 1),
                          "ms_per_frame": round(best, 3), "image_sha": hashlib.sha256(img.cpu().numpy().tobytes()).hexdigest()[:16],
                          "faults": r.counters()["traversal_faults"]}), flush=True)
        r.close()
