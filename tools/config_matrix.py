"""Every BASELINE.json config on one MI355X box: rate, frame time and parity against the CPU oracle.

    python tools/config_matrix.py [--out gpurun_out/configs.json] [--cpu-rows 24]

  C1  cornell 256x256, 1 spp, depth 1                  CPU oracle only (the ground-truth config), timed
  C2  cornell 1280x720, depth 4, 64 spp                 (a) 64 progressive frames x 1 spp (the reference's way), (b) one launch x 64 spp;
                                                        GPU LBVH build + wavefront pipeline; RMSE vs oracle on evenly spaced rows
  C3  Sponza-class atrium 1920x1080, 16 spp, depth 8    = bench.py's workload; here: rate + RMSE on sampled rows of frame 0 and 1
  C4  atrium 3840x2160, 16 spp, depth 8, 8 shards       this GPU renders shard 0 of 8 (its 1/8 of the rows): per-GPU rate + parity of its rows
  C5  hybrid mode (suntemple stand-in: the atrium)      ray-cast G-buffer + shadows + AO + GI + post, 1920x1080: ms per frame + parity

The cornell scene is the reference asset flattened (tests/golden/cornell_flat.npz); Sponza / suntemple are git-ignored
upstream, the seeded atrium stands in.  RMSE is over pixels and RGB of the linear rgba32f image, same seed.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch

import vkrt_amd  # noqa: F401
from vkrt_amd import abi
from vkrt_amd.flat_scene import FlatScene, make_push_constants, uniforms_from_matrices
from vkrt_amd.renderer import Renderer
from vkrt_amd.sharding import make_shard, shard_row_indices
import atrium
import camera_np
import oracle_py


def cam_for(w, h, **kw):
    return uniforms_from_matrices(*camera_np.global_uniforms(width=w, height=h, **kw))


def parity(gpu_rows, ref_rows):
    """RMSE over pixels and RGB (pixels that are NaN on both sides -- pow of a negative value in post.frag -- are skipped and
    counted), fraction of pixels whose bit patterns differ, largest absolute difference."""
    a, b = gpu_rows[..., :3].astype(np.float64), ref_rows[..., :3].astype(np.float64)
    both_nan = np.isnan(a) & np.isnan(b)
    d = np.where(both_nan, 0.0, a - b)
    return {"rmse": float(np.sqrt(np.mean(d * d))), "max_abs": float(np.abs(d).max()),
            "pixels_differing": float(np.mean(np.any(gpu_rows.view(np.uint32) != ref_rows.view(np.uint32), axis=-1))),
            "nan_on_both_sides": int(both_nan.any(axis=-1).sum()), "rows_compared": int(gpu_rows.shape[0])}


def timed_frames(r, frames, W, H, cam, spp, depth, lights, shard=None, image=None, seed0=0, frame0=0):
    """Runs `frames` progressive frames; returns (image tensor, ms per frame, rays per frame)."""
    torch.cuda.synchronize()
    r.reset_counters()
    t0 = time.perf_counter()
    for f in range(frames):
        pc = make_push_constants(samples=spp, depth=depth, frame=frame0 + f, lights_count=lights)
        image = r.pathtrace(pc, cam, W, H, seed=seed0 + f, shard=shard, image=image)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / frames
    c = r.counters()
    return image, ms, (c["rays_closest"] + c["rays_shadow"]) / frames


def timed_frames_call(r, frames, W, H, cam, spp, depth, lights, per_call, shard=None, image=None, seed0=0, frame0=0):
    """The same through vkrt_pathtrace_frames, `per_call` frames per library call (frames in flight inside the call)."""
    torch.cuda.synchronize()
    r.reset_counters()
    t0 = time.perf_counter()
    f = 0
    while f < frames:
        n = min(per_call, frames - f)
        pc = make_push_constants(samples=spp, depth=depth, frame=frame0 + f, lights_count=lights)
        image = r.pathtrace_frames(pc, cam, W, H, n, seed=seed0 + f, shard=shard, image=image)
        f += n
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / frames
    c = r.counters()
    return image, ms, (c["rays_closest"] + c["rays_shadow"]) / frames


def tessellation_block(a, out, lights, threads):
    # ---- C3 on Sponza-like tessellation: the same atrium with room-sized wall triangles, needle mouldings, strip drapery and dense
    #      small detail (atrium variant "nonuniform"), beside the uniform one, for the three builders -- is the headline number a
    #      property of the uniform tessellation?  (1080p, 16 spp, depth 8; nodes / triangles per ray from an instrumented frame)
    W, H = 1920, 1080
    cam = cam_for(W, H, **atrium.DEFAULT_CAMERA)
    rows = np.unique(np.linspace(0, H - 1, a.cpu_rows).astype(np.uint32))
    tess = {}
    for variant in (None, "nonuniform"):
        fl, inf = atrium.build_atrium(262144, seed=1, variant=variant)
        entry = {"triangles": int(inf["triangles"]), "unique_triangles": int(inf["unique_triangles"]), "builders": {}}
        for kind in ("ploc", "sah", "lbvh"):
            rr = Renderer(fl, device=0, build=kind)
            ai = rr.accel_info()
            timed_frames(rr, 1, W, H, cam, 16, 8, lights)
            _, ms, rpf = timed_frames(rr, 3, W, H, cam, 16, 8, lights)
            rr.reset_counters()
            rr.pathtrace(make_push_constants(samples=16, depth=8, frame=0, lights_count=lights), cam, W, H, seed=0, flags=abi.VKRT_TRACE_COUNT_TRAVERSAL)
            c = rr.counters()
            nr = c["rays_closest"] + c["rays_shadow"]
            entry["builders"][kind] = {"ms_per_frame": ms, "Mrays_s": rpf / ms / 1e3, "rays_per_frame": rpf, "nodes_per_ray": c["nodes_visited"] / nr,
                                       "tris_per_ray": c["tris_tested"] / nr, "sah_cost": ai["sah_cost"], "nodes": ai["node_count"], "depth": ai["max_depth"],
                                       "build_ms": ai["build_ms"], "traversal_faults": c["traversal_faults"]}
            if kind == "ploc" and variant is not None:  # parity of the default tree on the new scene: frame 0 against oracle rows
                img, _, _ = timed_frames(rr, 1, W, H, cam, 16, 8, lights)
                ref, _ = oracle_py.OracleScene(fl).render(make_push_constants(samples=16, depth=8, frame=0, lights_count=lights), cam, W, H, seed=0, rows=rows, threads=threads)
                entry["parity_ploc"] = parity(img.cpu().numpy()[rows], ref)
            rr.close()
        tess["uniform" if variant is None else variant] = entry
        print("C3 tessellation", variant, entry, flush=True)
    u, n = tess["uniform"]["builders"], tess["nonuniform"]["builders"]
    tess["nonuniform_over_uniform_Mrays_s"] = {k: n[k]["Mrays_s"] / u[k]["Mrays_s"] for k in u}
    out["configs"]["C3_tessellation_uniform_vs_sponza_like"] = tess


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "configs.json"))
    ap.add_argument("--cpu-rows", type=int, default=24, help="rows of each image the oracle renders for the parity figure")
    ap.add_argument("--only-tessellation", action="store_true", help="only the uniform vs Sponza-like tessellation block")
    ap.add_argument("--no-tessellation", action="store_true", help="skip the tessellation block")
    ap.add_argument("--only-hybrid-frames", type=int, default=0, help="render this many hybrid frames (C5) and exit: the workload of the rocprofv3 kernel-stats pass")
    a = ap.parse_args()
    if a.only_tessellation:
        out = {"device": torch.cuda.get_device_name(0), "cpu_threads": min(16, os.cpu_count() or 1), "configs": {}}
        tessellation_block(a, out, 8, out["cpu_threads"])
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        json.dump(out, open(a.out, "w"), indent=1)
        print("wrote", a.out)
        return
    if a.only_hybrid_frames:
        flat, info = atrium.build_atrium(262144, seed=1)
        lights = len(flat.lights)
        r = Renderer(flat, device=0, build="ploc")
        W, H = 1920, 1080
        cam = cam_for(W, H, **atrium.DEFAULT_CAMERA)
        pc = make_push_constants(samples=1, depth=8, frame=0, lights_count=lights)
        pc.useShadows, pc.useAO, pc.useGI = 1, 1, 1
        acc = None
        for f in range(a.only_hybrid_frames):
            pc.frame = f
            g = r.gbuffer_raycast(cam, W, H, lights_count=lights)
            acc = r.hybrid_trace(pc, cam, W, H, g, seed=3 + f, accum=acc)
            r.post(g["color"], acc, rt_mode=0, use_gi=1)
        torch.cuda.synchronize()
        r.close()
        print("hybrid frames", a.only_hybrid_frames)
        return
    threads = min(16, os.cpu_count() or 1)
    out = {"device": torch.cuda.get_device_name(0), "cpu_threads": threads, "configs": {}}
    cornell = FlatScene.load_npz(os.path.join(ROOT, "tests", "golden", "cornell_flat.npz"))
    orc_c = oracle_py.OracleScene(cornell)

    # ---- C1: CPU only
    W = H = 256
    cam = cam_for(W, H)
    pc = make_push_constants(samples=1, depth=1, frame=0, lights_count=1)
    t0 = time.perf_counter()
    img, c = orc_c.render(pc, cam, W, H, seed=0, threads=1)
    dt = time.perf_counter() - t0
    rays = c["rays_closest"] + c["rays_shadow"]
    out["configs"]["C1_cornell_256_1spp_d1_cpu"] = {"rays": rays, "seconds_1_thread": dt, "Mrays_s_1_thread": rays / dt / 1e6,
                                                    "mean_radiance": float(img[..., :3].mean())}
    print("C1", out["configs"]["C1_cornell_256_1spp_d1_cpu"], flush=True)

    # ---- C2: cornell 720p, depth 4, 64 spp, LBVH build on the GPU
    W, H = 1280, 720
    cam = cam_for(W, H)
    r = Renderer(cornell, device=0, build="ploc")
    rows = np.unique(np.linspace(0, H - 1, a.cpu_rows).astype(np.uint32))
    res = {"build": r.accel_info() if hasattr(r, "accel_info") else None}
    # (a) 64 frames x 1 spp, seed = frame index
    timed_frames(r, 4, W, H, cam, 1, 4, 1)  # warm-up
    img, ms, rpf = timed_frames(r, 64, W, H, cam, 1, 4, 1)
    ref = np.zeros((len(rows), W, 4), np.float32)
    for f in range(64):
        orc_c.render(make_push_constants(samples=1, depth=4, frame=f, lights_count=1), cam, W, H, seed=f, rows=rows, image=ref, threads=threads)
    res["progressive_64x1spp"] = {"ms_per_frame": ms, "ms_total": ms * 64, "Mrays_s": rpf / ms / 1e3, **parity(img.cpu().numpy()[rows], ref)}
    # (a') the same 64 frames handed to the library in ONE vkrt_pathtrace_frames call (and 8 calls of 8): frames in flight
    for per_call in (64, 8):
        timed_frames_call(r, 8, W, H, cam, 1, 4, 1, per_call)
        img2, ms2, rpf2 = timed_frames_call(r, 64, W, H, cam, 1, 4, 1, per_call)
        res[f"progressive_64x1spp_frames_call_{per_call}"] = {"ms_per_frame": ms2, "ms_total": ms2 * 64, "Mrays_s": rpf2 / ms2 / 1e3,
                                                              "bit_identical_to_single_calls": bool(torch.equal(img, img2)), **parity(img2.cpu().numpy()[rows], ref)}
    # (b) one launch x 64 spp
    timed_frames(r, 1, W, H, cam, 64, 4, 1)
    img, ms, rpf = timed_frames(r, 3, W, H, cam, 64, 4, 1, seed0=7)
    img, _, _ = timed_frames(r, 1, W, H, cam, 64, 4, 1, seed0=7)
    ref, _ = orc_c.render(make_push_constants(samples=64, depth=4, frame=0, lights_count=1), cam, W, H, seed=7, rows=rows, threads=threads)
    res["single_launch_64spp"] = {"ms_per_frame": ms, "Mrays_s": rpf / ms / 1e3, **parity(img.cpu().numpy()[rows], ref)}
    out["configs"]["C2_cornell_720p_64spp_d4_lbvh"] = res
    print("C2", res, flush=True)
    # ---- the reference's own default frame: config.json 1280x720, pcRay.samples 1, pcRay.depth 3 (config.json:10-11, hello_vulkan.cpp:911-912)
    dflt = {}
    timed_frames(r, 4, W, H, cam, 1, 3, 1)
    img, ms, rpf = timed_frames(r, 64, W, H, cam, 1, 3, 1)
    ref = np.zeros((len(rows), W, 4), np.float32)
    for f in range(64):
        orc_c.render(make_push_constants(samples=1, depth=3, frame=f, lights_count=1), cam, W, H, seed=f, rows=rows, image=ref, threads=threads)
    dflt["single_calls"] = {"ms_per_frame": ms, "fps": 1e3 / ms, "Mrays_s": rpf / ms / 1e3, **parity(img.cpu().numpy()[rows], ref)}
    for per_call in (64, 8, 3):
        timed_frames_call(r, 8, W, H, cam, 1, 3, 1, per_call)
        img2, ms2, rpf2 = timed_frames_call(r, 64, W, H, cam, 1, 3, 1, per_call)
        dflt[f"frames_call_{per_call}"] = {"ms_per_frame": ms2, "fps": 1e3 / ms2, "Mrays_s": rpf2 / ms2 / 1e3, "bit_identical_to_single_calls": bool(torch.equal(img, img2))}
    out["configs"]["reference_default_frame_cornell_720p_1spp_d3"] = dflt
    print("default frame", dflt, flush=True)
    r.close()

    # ---- C3 / C4 / C5: atrium
    flat, info = atrium.build_atrium(262144, seed=1)
    orc_a = oracle_py.OracleScene(flat, build_bvh=True, max_leaf=4)
    lights = len(flat.lights)
    r = Renderer(flat, device=0, build="ploc")
    W, H = 1920, 1080
    cam = cam_for(W, H, **atrium.DEFAULT_CAMERA)
    rows = np.unique(np.linspace(0, H - 1, a.cpu_rows).astype(np.uint32))
    timed_frames(r, 1, W, H, cam, 16, 8, lights)
    _, ms, rpf = timed_frames(r, 3, W, H, cam, 16, 8, lights)
    img, _, _ = timed_frames(r, 2, W, H, cam, 16, 8, lights)  # frames 0 and 1 accumulated
    ref = np.zeros((len(rows), W, 4), np.float32)
    for f in range(2):
        orc_a.render(make_push_constants(samples=16, depth=8, frame=f, lights_count=lights), cam, W, H, seed=f, rows=rows, image=ref, threads=threads)
    timed_frames_call(r, 3, W, H, cam, 16, 8, lights, 6)
    img6, ms6, rpf6 = timed_frames_call(r, 12, W, H, cam, 16, 8, lights, 6, frame0=0)
    out["configs"]["C3_atrium_1080p_16spp_d8"] = {"triangles": int(info["triangles"]) if isinstance(info, dict) and "triangles" in info else None,
                                                  "ms_per_frame": ms, "Mrays_s": rpf / ms / 1e3, **parity(img.cpu().numpy()[rows], ref),
                                                  "frames_call_6": {"ms_per_frame": ms6, "Mrays_s": rpf6 / ms6 / 1e3, "note": "bench.py's way: 6 frames per vkrt_pathtrace_frames call, 3 in flight"}}
    print("C3", out["configs"]["C3_atrium_1080p_16spp_d8"], flush=True)

    W, H = 3840, 2160
    cam = cam_for(W, H, **atrium.DEFAULT_CAMERA)
    for rank in (0, 5):
        shard = make_shard(W, H, 8, rank)
        timed_frames(r, 1, W, H, cam, 16, 8, lights, shard=shard)
        _, ms, rpf = timed_frames(r, 3, W, H, cam, 16, 8, lights, shard=shard)
        timed_frames_call(r, 3, W, H, cam, 16, 8, lights, 6, shard=shard)
        _, ms6, rpf6 = timed_frames_call(r, 6, W, H, cam, 16, 8, lights, 6, shard=shard)
        img, _, _ = timed_frames(r, 1, W, H, cam, 16, 8, lights, shard=shard)
        grow = shard_row_indices(H, 8, rank)
        pick = np.unique(np.linspace(0, len(grow) - 1, max(6, a.cpu_rows // 2)).astype(np.int64))
        ref, _ = orc_a.render(make_push_constants(samples=16, depth=8, frame=0, lights_count=lights), cam, W, H, seed=0,
                              rows=grow[pick].astype(np.uint32), threads=threads)
        out["configs"][f"C4_atrium_4k_16spp_d8_shard{rank}of8"] = {"local_rows": int(len(grow)), "ms_per_frame": ms, "Mrays_s_this_gpu": rpf / ms / 1e3,
                                                                  "frames_call_6": {"ms_per_frame": ms6, "Mrays_s_this_gpu": rpf6 / ms6 / 1e3},
                                                                  "note": "one of eight shards; the N-GPU run is bench.py --gpus N (driver)",
                                                                  **parity(img.cpu().numpy()[pick], ref)}
        print("C4", rank, out["configs"][f"C4_atrium_4k_16spp_d8_shard{rank}of8"], flush=True)

    W, H = 1920, 1080
    cam = cam_for(W, H, **atrium.DEFAULT_CAMERA)
    pc = make_push_constants(samples=1, depth=8, frame=0, lights_count=lights)
    pc.useShadows, pc.useAO, pc.useGI = 1, 1, 1

    def hybrid_frame():
        g = r.gbuffer_raycast(cam, W, H, lights_count=lights)
        acc = r.hybrid_trace(pc, cam, W, H, g, seed=3)
        return g, acc, r.post(g["color"], acc, rt_mode=0, use_gi=1)

    hybrid_frame()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g, acc, disp = hybrid_frame()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / 5
    rows = np.unique(np.linspace(0, H - 1, a.cpu_rows).astype(np.uint32))
    go = orc_a.gbuffer(cam, W, H, lights_count=lights, rows=rows, threads=threads)
    ao, _ = orc_a.hybrid(pc, cam, W, H, go, seed=3, rows=rows, threads=threads)
    want = oracle_py.post(go["color"], ao, rt_mode=0, use_gi=1)
    out["configs"]["C5_hybrid_atrium_1080p_shadow_ao_gi"] = {"ms_per_frame": ms, "fps": 1e3 / ms,
                                                             "gbuffer_color": parity(g["color"].cpu().numpy()[rows], go["color"]),
                                                             "accumulated": parity(acc.cpu().numpy()[rows], ao),
                                                             "display": parity(disp.cpu().numpy()[rows], want),
                                                             "display_note": "post.frag's pow() is outside the bit-exact arithmetic profile: GPU and CPU pow differ in the last bits"}
    print("C5", out["configs"]["C5_hybrid_atrium_1080p_shadow_ao_gi"], flush=True)
    r.close()

    if not a.no_tessellation:
        tessellation_block(a, out, lights, threads)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
