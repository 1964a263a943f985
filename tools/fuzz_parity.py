"""Randomised differential test: random small scenes (triangle soups + quads, instanced with arbitrary affine transforms, random
PBR materials with random NPOT textures, point / directional / spot lights), random cameras, builders, options, sample counts and
depths -- the HIP path tracer and the hybrid passes against the CPU oracle, bit for bit.

    python tools/fuzz_parity.py [--seconds 300] [--seed 1] [--out gpurun_out/fuzz_parity.json]

Bit-identity is the expectation for the path tracer (both sides follow one arithmetic profile and the image is a function of the
triangle set); the hybrid planes may differ in isolated pixels where libm's exp2f / the G-buffer's float rounding feed a
quantisation boundary, so those are compared by mismatch fraction.  Every failing case prints its seed for replay (--only SEED)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np

import vkrt_amd  # noqa: F401
from vkrt_amd import abi
from vkrt_amd.flat_scene import LIGHT_DTYPE, MAT_DTYPE, NODE_DTYPE, PRIM_DTYPE, FlatScene, make_push_constants, uniforms_from_matrices
from vkrt_amd.renderer import Renderer
import camera_np
import oracle_py


def random_scene(rng):
    meshes = []
    n_mesh = int(rng.integers(1, 4))
    for _ in range(n_mesh):
        kind = rng.integers(0, 3)
        if kind == 0:  # soup
            n = int(rng.integers(1, 400))
            if rng.random() < 0.04:  # now and then large enough for the SAH top of the clustered build (> 4096 triangles) and its binned ranges
                n = int(rng.integers(5000, 60000))
            c = rng.uniform(-2, 2, (n, 1, 3))
            P = (c + rng.normal(0, rng.choice([0.05, 0.3, 1.0]), (n, 3, 3))).reshape(-1, 3)
            idx = np.arange(3 * n)
        elif kind == 1:  # grid (shared vertices, coplanar neighbours: edge and tie cases)
            g = int(rng.integers(1, 12))
            xs = np.linspace(-2, 2, g + 1)
            gx, gz = np.meshgrid(xs, xs)
            P = np.stack([gx.ravel(), rng.normal(0, 0.02, gx.size) * rng.integers(0, 2), gz.ravel()], -1)
            idx = []
            for j in range(g):
                for i in range(g):
                    a = j * (g + 1) + i
                    idx += [a, a + g + 1, a + 1, a + 1, a + g + 1, a + g + 2]
            idx = np.array(idx)
        else:  # box-ish shell of big triangles
            c = rng.uniform(-1, 1, (8, 3)) * 0.2 + np.array([[x, y, z] for x in (-3, 3) for y in (-2, 3) for z in (-3, 3)])
            P = c
            idx = np.array([0, 1, 2, 1, 3, 2, 4, 6, 5, 5, 6, 7, 0, 4, 1, 1, 4, 5, 2, 3, 6, 3, 7, 6, 0, 2, 4, 2, 6, 4, 1, 5, 3, 3, 5, 7])
        if rng.random() < 0.15 and idx.size >= 6:
            idx = idx[: idx.size - int(rng.integers(1, 3))]  # ragged index count (hello_vulkan.cpp:960-969)
        hostile = rng.random()
        if hostile < 0.05:      # degenerate triangles: two corners coincide / all three on a line
            P = P.copy()
            t = idx[: 3 * (idx.size // 3)].reshape(-1, 3)
            pick = rng.random(t.shape[0]) < 0.3
            P[t[pick, 1]] = P[t[pick, 0]]
        elif hostile < 0.10:    # a pile: every triangle the same three points
            P = np.tile(P[:3], (P.shape[0] // 3 + 1, 1))[: P.shape[0]]
        elif hostile < 0.15:    # far from the origin: coordinates ~1e3..1e4 with unit-size detail
            P = P + rng.choice([1.0e3, 1.0e4]) * rng.choice([-1.0, 1.0], 3)
        elif hostile < 0.20:    # flat: zero thickness along one axis, axis-aligned
            P = P.copy(); P[:, int(rng.integers(0, 3))] = float(rng.integers(-2, 3))
        elif hostile < 0.25:    # one huge triangle among the small ones
            P = P.copy(); P[idx[:3]] = rng.uniform(-1, 1, (3, 3)) * 500.0
        nrm = rng.normal(size=P.shape)
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        tan = np.concatenate([np.roll(nrm, 1, axis=1), np.where(rng.random((P.shape[0], 1)) < 0.5, 1.0, -1.0)], 1)
        uv = rng.uniform(-1, 3, (P.shape[0], 2))
        meshes.append((P.astype(np.float32), nrm.astype(np.float32), tan.astype(np.float32), uv.astype(np.float32), idx.astype(np.uint32)))
    n_tex = int(rng.integers(0, 5))
    textures = []
    for _ in range(n_tex):
        w, h = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        textures.append({"rgba8": rng.integers(0, 256, (h, w, 4), dtype=np.uint8), "is_srgb": bool(rng.random() < 0.5)})
    n_mat = int(rng.integers(1, 5))
    mats = np.zeros(n_mat, MAT_DTYPE)
    for m in mats:
        m["pbrBaseColorFactor"] = [*rng.uniform(0.05, 1.0, 3), 1.0]
        for key in ("pbrBaseColorTexture", "metallicRoughnessTexture", "normalTexture", "emissiveTexture"):
            m[key] = int(rng.integers(0, n_tex)) if (n_tex and rng.random() < 0.5) else -1
        m["metallicFactor"] = rng.choice([0.0, 1.0, rng.random()])
        m["roughnessFactor"] = rng.choice([0.0, 1.0, rng.random()])
        m["emissiveFactor"] = rng.uniform(0, 2, 3) * (rng.random() < 0.4)
    pos = np.concatenate([m[0] for m in meshes]); nrm = np.concatenate([m[1] for m in meshes])
    tan = np.concatenate([m[2] for m in meshes]); uv = np.concatenate([m[3] for m in meshes])
    idx = np.concatenate([m[4] for m in meshes])
    pm = np.zeros(n_mesh, PRIM_DTYPE)
    vo = io = 0
    for k, m in enumerate(meshes):
        pm[k] = (io, m[4].size, vo, m[0].shape[0], int(rng.integers(-1, n_mat)))
        vo += m[0].shape[0]; io += m[4].size
    n_nodes = int(rng.integers(1, 6))
    nodes = np.zeros(n_nodes, NODE_DTYPE)
    for nd in nodes:
        M = np.eye(4)
        if rng.random() < 0.7:
            A = rng.normal(size=(3, 3))
            Q, _ = np.linalg.qr(A)
            S = np.diag(rng.uniform(0.3, 2.0, 3) * np.where(rng.random(3) < 0.15, -1.0, 1.0))
            M[:3, :3] = Q @ S
            M[:3, 3] = rng.uniform(-2, 2, 3)
        nd["worldMatrix"] = M.T.astype(np.float32).ravel()  # column-major storage
        nd["primMesh"] = int(rng.integers(0, n_mesh))
    if rng.random() < 0.08:  # extreme instance scales (1e-3 .. 1e3, different per axis)
        for nd in nodes:
            M = np.asarray(nd["worldMatrix"], np.float64).reshape(4, 4).T
            M[:3, :3] = M[:3, :3] @ np.diag(10.0 ** rng.uniform(-3, 3, 3))
            nd["worldMatrix"] = M.T.astype(np.float32).ravel()
    if n_nodes > 1 and rng.random() < 0.1:  # the same instance twice: every triangle coincides with one of another gid (tie rule)
        nodes[1] = nodes[0]
    n_l = int(rng.integers(1, 5))
    lights = np.zeros(n_l, LIGHT_DTYPE)
    for l in lights:
        l["position"] = rng.uniform(-4, 4, 3)
        l["color"] = rng.uniform(0, 1, 3)
        l["intensity"] = rng.uniform(0.5, 60)
        l["type"] = int(rng.integers(0, 3))
    return FlatScene(pos, nrm, tan, uv, idx, pm, mats, lights, nodes, textures)


def needle_measure(flat):
    """smallest height / longest-edge ratio over the instanced triangles (world space)"""
    worst = 1.0
    P = flat.positions.astype(np.float64)
    for nd in flat.nodes:
        M = np.asarray(nd["worldMatrix"], np.float64).reshape(4, 4).T
        pm = flat.prim_meshes[int(nd["primMesh"])]
        n = int(pm["indexCount"]) // 3
        if n == 0:
            continue
        t = flat.indices[int(pm["firstIndex"]): int(pm["firstIndex"]) + 3 * n].reshape(n, 3).astype(np.int64) + int(pm["vertexOffset"])
        W = P[t] @ M[:3, :3].T + M[:3, 3]
        e1, e2 = W[:, 1] - W[:, 0], W[:, 2] - W[:, 0]
        longest = np.maximum(np.maximum(np.linalg.norm(e1, axis=1), np.linalg.norm(e2, axis=1)), np.linalg.norm(e2 - e1, axis=1))
        area2 = np.linalg.norm(np.cross(e1, e2), axis=1)
        ok = longest > 0
        if ok.any():
            worst = min(worst, float((area2[ok] / longest[ok] ** 2).min()))
    return worst


def run_case(seed, verbose=False, hook=None, force_opts=None):
    """force_opts: {option: value} applied on top of the case's own draw (e.g. the watertight test for a recorded seed)."""
    rng = np.random.default_rng(seed)
    rng3 = np.random.default_rng([seed, 3])  # round-3 options come from their own stream: the cases of earlier campaigns keep their draws
    rng4 = np.random.default_rng([seed, 4])  # round 4: triangle pre-splitting, frames in one call
    flat = random_scene(rng)
    W, H = int(rng.integers(8, 64)), int(rng.integers(8, 48))
    if rng.random() < 0.08:  # now and then an image large enough for the sub-frame pipeline (>= 512 tiles)
        W, H = int(rng.integers(200, 330)), int(rng.integers(130, 210))
    eye = rng.uniform(-5, 5, 3); center = rng.uniform(-1, 1, 3)
    far = np.abs(flat.positions).max() > 100.0
    if rng.random() < 0.05:  # a view from far away / from inside the geometry's bounding box
        eye = center + rng.uniform(-1, 1, 3) * float(rng.choice([0.01, 300.0]))
    if far:  # look at the geometry wherever it is
        c0 = flat.positions.mean(0) if flat.nodes.shape[0] == 0 else np.asarray(flat.positions.mean(0), np.float64)
        center = c0 + rng.uniform(-1, 1, 3); eye = c0 + rng.uniform(-6, 6, 3)
    up = (0, 1, 0)
    if rng.random() < 0.1:  # axis-parallel view from a lattice point (rays along cell walls, origins on planes)
        ax = int(rng.integers(0, 3))
        center = np.round(center); eye = center.copy(); eye[ax] += float(rng.integers(2, 6))
        if ax == 1:
            up = (0, 0, 1)
    cam = uniforms_from_matrices(*camera_np.global_uniforms(width=W, height=H, eye=tuple(eye), center=tuple(center), up=up, fov=float(rng.uniform(20, 100))))
    kind = str(rng.choice(["sah", "lbvh", "ploc"]))
    opts = {}
    if rng.random() < 0.2: opts[abi.VKRT_OPT_BVH_LAYOUT] = 0
    if rng.random() < 0.2: opts[abi.VKRT_OPT_WF_SHARE] = int(rng.choice([0, 4, 40]))
    if rng.random() < 0.15: opts[abi.VKRT_OPT_MODE] = 0
    if rng.random() < 0.2: opts[abi.VKRT_OPT_WF_SUBFRAMES] = int(rng.integers(1, 5))
    if rng3.random() < 0.2: opts[abi.VKRT_OPT_WATERTIGHT] = 1               # the other triangle test, on both sides
    if rng3.random() < 0.2: opts[abi.VKRT_OPT_SKIP_DEAD_SHADOW_RAYS] = 1    # must not change a pixel
    if rng3.random() < 0.3: opts[abi.VKRT_OPT_WF_SHARE_FLAGS] = int(rng3.integers(0, 16))  # child donation and the order rules of any-hit walks on and off
    if rng3.random() < 0.15:                                                 # the any-hit alpha / dissolve stage, with non-opaque materials
        opts[abi.VKRT_OPT_ANYHIT_DISSOLVE] = 1
        for m in flat.materials:
            if rng3.random() < 0.6:
                m["pbrBaseColorFactor"][3] = float(rng3.choice([0.0, 0.05, 0.5, 0.95, float(rng3.random())]))
    if abi.VKRT_OPT_WF_SHARE_FLAGS in opts and rng4.random() < 0.5: opts[abi.VKRT_OPT_WF_SHARE_FLAGS] |= 16  # triangle-group donation with the drawn rules (the default has it on)
    if rng4.random() < 0.35: opts[abi.VKRT_OPT_SPLIT_BUDGET] = int(rng4.choice([10, 30, 100]))  # several references per large triangle (device builders)
    rng5 = np.random.default_rng([seed, 5])  # round 5 (a stream of its own: every earlier seed keeps its case): the automatic budget
    if rng5.random() < 0.15: opts[abi.VKRT_OPT_SPLIT_BUDGET] = -1
    frames_call = rng4.random() < 0.3  # the frames of the sequence in ONE vkrt_pathtrace_frames call, with whatever lanes the draw gives
    if frames_call:
        opts[abi.VKRT_OPT_WF_FRAMES_IN_FLIGHT] = int(rng4.integers(1, 5))
    opts.update(force_opts or {})
    spp, depth, frames = int(rng.integers(1, 4)), int(rng.integers(1, 7)), int(rng.integers(1, 3))
    if frames_call and rng4.random() < 0.5:
        frames = int(rng4.integers(2, 6))
    L = int(rng.integers(1, len(flat.lights) + 1))
    if rng.random() < 0.05:  # long sample sequences and deep paths (pixels far out of step with each other in the paired rounds)
        spp, depth = int(rng.integers(4, 12)), int(rng.integers(7, 16))
    edge = rng.random()
    if edge < 0.02: spp = 0          # degenerate launches: the pixel is resolved without a ray
    elif edge < 0.04: depth = 0
    elif edge < 0.06: L = 0          # the light pick then always reads light 0 (int(rnd * 0))
    elif edge < 0.08: W, H = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    if frames_call and W * H * max(spp, 1) * max(depth, 1) * frames > 3_000_000:  # (the scalar oracle renders every frame: keep a long sequence of a large, deep case short)
        frames = 2
    first_frame = int(rng.integers(0, 3)) if rng.random() < 0.2 else 0  # start a sequence at frame > 0 (jittered from the first launch on)
    flags = abi.VKRT_TRACE_SEED_INDEX_ROW_MAJOR if rng.random() < 0.5 else 0
    info = dict(seed=seed, tris=flat.instanced_triangle_count, size=(W, H), kind=kind, opts={int(k): int(v) for k, v in opts.items()}, spp=spp, depth=depth, frames=frames)
    orc = oracle_py.OracleScene(flat)
    orc.set_watertight(opts.get(abi.VKRT_OPT_WATERTIGHT, 0) == 1)
    orc.set_dissolve(opts.get(abi.VKRT_OPT_ANYHIT_DISSOLVE, 0) == 1)
    r = Renderer(flat, device=0, build=kind, options=opts)
    problems = []
    brute = bool(rng.random() < 0.3) and W * H * spp * frames < 6000  # the oracle's loop over all triangles as the referee
    try:
        img = ref = None
        if frames_call:  # (a call that starts at frame > 0 blends into the zero image, like the single-frame calls below do)
            img = r.pathtrace_frames(make_push_constants(samples=spp, depth=depth, frame=first_frame, lights_count=L), cam, W, H, frames,
                                     seed=seed + first_frame, flags=flags, image=img)
        for f in range(first_frame, first_frame + frames):
            pc = make_push_constants(samples=spp, depth=depth, frame=f, lights_count=L)
            if not frames_call:
                img = r.pathtrace(pc, cam, W, H, seed=seed + f, flags=flags, image=img)
            ref, _ = orc.render(pc, cam, W, H, seed=seed + f, flags=flags, image=ref, use_bvh=not brute)
        got = img.cpu().numpy()
        same = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
        if not same.all() and not brute:
            # referee: the oracle's loop over all triangles.  (Needle triangles hundreds of units long make the triangle test accept
            # points centimetres outside the triangle; the oracle's own tree walk, with per-leaf boxes, can prune such a "hit".)
            ref = None
            for f in range(first_frame, first_frame + frames):
                pc = make_push_constants(samples=spp, depth=depth, frame=f, lights_count=L)
                ref, _ = orc.render(pc, cam, W, H, seed=seed + f, flags=flags, image=ref, use_bvh=False)
            same = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
            info["oracle_tree_differs_from_brute_force"] = True
        if not same.all():
            frac = float(1 - same.all(-1).mean())
            if frac < 1e-3 and needle_measure(flat) < 1e-4:
                # Needles (height below 1e-4 of the length, e.g. an instance flattened by a 1 : 1e5 scale): the triangle test accepts
                # points far outside such a triangle, further than any box pad reaches, so whether the "hit" is found depends on
                # the tree.  Known limit, not counted as a finding.
                info["needle_limit_pixels"] = int((~same.all(-1)).sum())
            else:
                problems.append(("pathtrace", frac, float(np.nanmax(np.abs(got - ref)))))
        if hook is not None and not same.all():  # tools/debug_case.py: localise a path-tracer mismatch
            hook(dict(stage="pathtrace", flat=flat, renderer=r, oracle=orc, cam=cam, W=W, H=H, got=got, ref=ref, seed=seed, flags=flags, kind=kind, opts=opts,
                      spp=spp, depth=depth, L=L, first_frame=first_frame, frames=frames))
        c = r.counters()
        info["rays"] = int(c["rays_closest"] + c["rays_shadow"])
        info["lit"] = float((np.nan_to_num(got[..., :3]).sum(-1) > 0).mean())
        if c["traversal_faults"]:
            problems.append(("faults", c["traversal_faults"], 0))
        chk = r.check_accel()
        if chk["triangles_missing"] or chk["triangles_repeated"] or chk["box_violations"] or chk["bad_references"] or chk["triangles_uncovered"]:
            problems.append(("tree", chk, 0))
        if rng.random() < 0.35 and abi.VKRT_OPT_MODE not in opts:  # a random image-strip shard equals the rows of the whole frame
            count = int(rng.integers(2, 6)); index = int(rng.integers(0, count)); strip = int(rng.choice([1, 3, 16]))
            sh = abi.Shard(W, H, strip, count, index)
            rows = [y for y in range(H) if (y // strip) % count == index]
            part = None
            for f in range(first_frame, first_frame + frames):
                pc = make_push_constants(samples=spp, depth=depth, frame=f, lights_count=L)
                part = r.pathtrace(pc, cam, W, H, seed=seed + f, flags=flags, image=part, shard=sh)
            part = part.cpu().numpy()
            if part.shape[0] != len(rows) or not np.array_equal(part.view(np.uint32), got[rows].view(np.uint32)):
                problems.append(("shard", (count, index, strip), 0))
        if rng.random() < 0.1 and first_frame == 0 and spp > 0 and depth > 0 and L == len(flat.lights):  # the C++ host end to end (its own scene handle and oracle, default options): file -> loader -> HelloVkrt -> image, against the oracle on the numpy ingest
            import tempfile
            import gltf_export
            import gltf_flatten
            from vkrt_amd import host_py
            with tempfile.TemporaryDirectory() as tmp:
                mode = int(rng.integers(0, 3))
                path = os.path.join(tmp, "s.glb" if mode == 1 else "s.gltf")
                gltf_export.export_gltf(flat, path, glb=(mode == 1), embed=(mode == 2), write_lights=True)
                cfg = dict(eye=tuple(float(v) for v in eye), center=tuple(float(v) for v in center), up=up, fov=float(rng.uniform(20, 100)))
                bf = {"sah": abi.VKRT_BUILD_SAH_HOST, "lbvh": abi.VKRT_BUILD_LBVH_GPU, "ploc": abi.VKRT_BUILD_PLOC_GPU}[kind]
                himg = host_py.render_gltf(path, W, H, samples=spp, depth=depth, frames=frames, seed0=seed, build=bf, **cfg)
                f2 = gltf_flatten.load_gltf(path)
                o2 = oracle_py.OracleScene(f2)
                u2 = host_py.global_uniforms(width=W, height=H, **cfg)
                href = None
                for f in range(frames):
                    pc = make_push_constants(samples=spp, depth=depth, frame=f, lights_count=len(f2.lights))
                    href, _ = o2.render(pc, u2, W, H, seed=seed + f, image=href)
                ok = (himg.view(np.uint32) == href.view(np.uint32)) | (np.isnan(himg) & np.isnan(href))
                if not ok.all():
                    problems.append(("cpp_host", float(1 - ok.all(-1).mean()), float(np.nanmax(np.abs(himg - href)))))
        # hybrid passes (with the NRD front-end planes in a third of the cases)
        vm = None
        if rng.random() < 0.33:
            vi = np.asarray(cam.viewInverse.m[:], np.float64).reshape(4, 4).T  # column-major storage -> matrix
            vm = np.linalg.inv(vi).T.astype(np.float32).ravel()
        g = r.gbuffer_raycast(cam, W, H, lights_count=max(L, 1), view_matrix=vm)
        gref = orc.gbuffer(cam, W, H, lights_count=max(L, 1)) if vm is None else orc.gbuffer_nrd(cam, vm, W, H, lights_count=max(L, 1))
        def gbuffer_findings(refplanes):
            out = []
            for k in refplanes:
                a, b = g[k].cpu().numpy(), refplanes[k]
                diff = (a.view(np.uint32) != b.view(np.uint32)) & ~(np.isnan(a) & np.isnan(b))
                bad = diff.reshape(diff.shape[0], diff.shape[1], -1).any(-1).mean()
                if bad > (0.02 if k.startswith("nrd") else 0.0):  # (the NRD planes are quantised: a boundary now and then; the G-buffer itself must be exact)
                    out.append(("gbuffer_" + k, float(bad), float(np.nanmax(np.abs(a - b)))))
            return out
        gf = gbuffer_findings(gref)
        if gf:  # referee: the primary rays by the oracle's loop over all triangles
            gref = orc.gbuffer(cam, W, H, lights_count=max(L, 1), use_bvh=False) if vm is None else orc.gbuffer_nrd(cam, vm, W, H, lights_count=max(L, 1), use_bvh=False)
            gf = gbuffer_findings(gref)
            info["oracle_tree_differs_from_brute_force"] = True
        problems += gf
        pc = make_push_constants(samples=1, depth=max(depth, 2), frame=0, lights_count=max(L, 1))
        pc.useShadows, pc.useAO, pc.useGI = int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.integers(0, 2))
        gnp = {k: v.cpu().numpy() for k, v in g.items()}
        acc = r.hybrid_trace(pc, cam, W, H, g, seed=seed, flags=flags).cpu().numpy()
        if rng.random() < 0.25:  # the hybrid passes of a random image-strip shard are the rows of the whole frame
            count = int(rng.integers(2, 5)); index = int(rng.integers(0, count)); strip = int(rng.choice([2, 16]))
            sh = abi.Shard(W, H, strip, count, index)
            rows = [y for y in range(H) if (y // strip) % count == index]
            gp = r.gbuffer_raycast(cam, W, H, lights_count=max(L, 1), view_matrix=vm, shard=sh)
            wrong = [k for k in gp if not np.array_equal(gp[k].cpu().numpy().view(np.uint32), gnp[k][rows].view(np.uint32))]  # (before the trace packs the radiance plane)
            accp = r.hybrid_trace(pc, cam, W, H, gp, seed=seed, flags=flags, shard=sh).cpu().numpy()
            if accp.shape[0] != len(rows) or not np.array_equal(accp.view(np.uint32), acc[rows].view(np.uint32)):
                wrong.append("accum")
            if vm is not None and not np.array_equal(gp["nrdRadianceHitDist"].cpu().numpy().view(np.uint32), g["nrdRadianceHitDist"].cpu().numpy()[rows].view(np.uint32)):
                wrong.append("nrdRadianceHitDist")
            if wrong:
                problems.append(("hybrid_shard", (count, index, strip), wrong))
        if vm is None:
            accr, _ = orc.hybrid(pc, cam, W, H, gnp, seed=seed, flags=flags)
        else:
            accr, radr = orc.hybrid_nrd(pc, cam, W, H, gnp, seed=seed, flags=flags)
            rad = g["nrdRadianceHitDist"].cpu().numpy()
            badr = ((rad.view(np.uint32) != radr.view(np.uint32)) & ~(np.isnan(rad) & np.isnan(radr))).any(-1).mean()
            if badr > 0.02:  # (exp2f differs in the last bits between libm and the GPU: a quantisation boundary now and then)
                problems.append(("nrd_radiance", float(badr), float(np.nanmax(np.abs(rad - radr)))))
        bad = ((acc.view(np.uint32) != accr.view(np.uint32)) & ~(np.isnan(acc) & np.isnan(accr))).any(-1).mean()
        if bad > 0.0:
            # referee: the oracle's loop over all triangles (which of the two tree walks pruned a hit the triangle test accepts?)
            accb, _ = orc.hybrid(pc, cam, W, H, gnp, seed=seed, flags=flags, use_bvh=False)
            gb = int(((acc.view(np.uint32) != accb.view(np.uint32)) & ~(np.isnan(acc) & np.isnan(accb))).any(-1).sum())
            tb = int(((accr.view(np.uint32) != accb.view(np.uint32)) & ~(np.isnan(accr) & np.isnan(accb))).any(-1).sum())
            if gb:
                problems.append(("hybrid", float(bad), float(np.nanmax(np.abs(acc - accb))), {"gpu_vs_brute_pixels": gb, "oracle_tree_vs_brute_pixels": tb}))
            else:
                info["oracle_tree_differs_from_brute_force"] = True
        if hook is not None:  # tools/debug_case.py: everything a replay needs to look at single pixels / rays
            hook(dict(stage="hybrid", flat=flat, renderer=r, oracle=orc, cam=cam, pc=pc, W=W, H=H, gbuffer=gnp, acc=acc, accr=accr, seed=seed, flags=flags, kind=kind, opts=opts))
    finally:
        r.close()
    if verbose or problems:
        print(json.dumps({**info, "problems": problems}, default=str), flush=True)
    return info, problems


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", type=int, default=None, help="replay one case")
    ap.add_argument("--start", type=int, default=0, help="first case number of the campaign (seed = --seed * 1000003 + number)")
    ap.add_argument("--count", type=int, default=0, help="stop after this many cases (0 = run for --seconds)")
    ap.add_argument("--echo", action="store_true", help="print every seed before its case runs, and the Python stack of a case that takes longer than 60 s")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "fuzz_parity.json"))
    ap.add_argument("--force-opt", action="append", default=[], metavar="ID=VALUE", help="apply this execution option to every case (e.g. 10=1: the watertight test)")
    a = ap.parse_args()
    force = {int(k): int(v) for k, v in (x.split("=") for x in a.force_opt)} or None
    if a.only is not None:
        run_case(a.only, verbose=True, force_opts=force)
        return 0
    t0 = time.time()
    n = bad = 0
    first = a.start
    failures = []
    tris = rays = tree_notes = needle_notes = needle_notes_wt = n_wt = n_dissolve = n_skip = n_split = n_frames_call = 0
    lit = 0.0
    while time.time() - t0 < a.seconds and (a.count == 0 or n < a.count):
        seed = a.seed * 1000003 + first + n
        if a.echo:
            import faulthandler
            print(f"case {first + n} seed {seed}", flush=True)
            faulthandler.dump_traceback_later(60, exit=False)
        try:
            info, problems = run_case(seed, force_opts=force)
        except Exception as e:  # an API error is a finding too
            info, problems = dict(seed=seed), [("exception", repr(e), 0)]
            print(json.dumps({"seed": seed, "exception": repr(e)}), flush=True)
        if a.echo:
            faulthandler.cancel_dump_traceback_later()
        n += 1
        tris += info.get("tris", 0)
        tree_notes += 1 if info.get("oracle_tree_differs_from_brute_force") else 0
        needle_notes += 1 if info.get("needle_limit_pixels") else 0
        o = info.get("opts", {})
        n_wt += 1 if o.get(abi.VKRT_OPT_WATERTIGHT) else 0
        n_dissolve += 1 if o.get(abi.VKRT_OPT_ANYHIT_DISSOLVE) else 0
        n_skip += 1 if o.get(abi.VKRT_OPT_SKIP_DEAD_SHADOW_RAYS) else 0
        n_split += 1 if o.get(abi.VKRT_OPT_SPLIT_BUDGET) else 0
        n_frames_call += 1 if abi.VKRT_OPT_WF_FRAMES_IN_FLIGHT in o else 0
        needle_notes_wt += 1 if (info.get("needle_limit_pixels") and o.get(abi.VKRT_OPT_WATERTIGHT)) else 0
        rays += info.get("rays", 0)
        lit += info.get("lit", 0.0)
        if problems:
            bad += 1
            failures.append({**info, "problems": problems})
        if n % 50 == 0:
            print(f"[{time.time() - t0:6.0f} s] {n} cases, {bad} with findings", flush=True)
    out = {"forced_options": force, "cases": n, "with_findings": bad, "seconds": round(time.time() - t0, 1), "triangles_total": tris, "rays_total": rays, "mean_lit_pixel_fraction": round(lit / max(n, 1), 3),
           "cases_where_only_the_oracle_tree_walk_differed_from_brute_force": tree_notes,
           "cases_at_the_needle_limit": needle_notes, "cases_at_the_needle_limit_with_the_watertight_test": needle_notes_wt,
           "cases_with_the_watertight_test": n_wt, "cases_with_the_anyhit_dissolve_stage": n_dissolve, "cases_skipping_dead_shadow_rays": n_skip,
           "cases_with_triangle_presplitting": n_split, "cases_through_vkrt_pathtrace_frames": n_frames_call, "first_seed": a.seed * 1000003, "failures": failures[:50]}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1, default=str)
    print(json.dumps({k: v for k, v in out.items() if k != "failures"}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
