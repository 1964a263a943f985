"""Hybrid mode (reference rtMode == 0, SURVEY.md 8f row 1 / BASELINE config 5): G-buffer ray cast, raytraceHybrid.rgen,
post composite.  CPU tests pin the oracle restatement; GPU tests compare the HIP path with it."""
import numpy as np
import pytest

import oracle_py
from conftest import default_camera
from vkrt_amd.flat_scene import make_push_constants


def test_half_quantisation_matches_ieee_half():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(0, 1, 4000), rng.uniform(-70000, 70000, 500), 2.0 ** rng.uniform(-30, -13, 500),
                        [0.0, 1.0, 0.5, 65504.0, 65519.9, 65520.0, 6.1e-5, 5.96e-8, 2.98e-8, 2.9802325e-8, 1e-9]]).astype(np.float32)
    got = oracle_py.quantize_half(x)
    with np.errstate(over="ignore"):
        want = x.astype(np.float16).astype(np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_oracle_gbuffer_structure(cornell_oracle, cornell_flat):
    W, H = 96, 64
    cam = default_camera(W, H)
    g = cornell_oracle.gbuffer(cam, W, H, lights_count=1, clear_color=(0.2, 0.3, 0.4, 1.0))
    hit = np.any(g["position"][..., :3] != 0, axis=-1)
    assert 0.5 < hit.mean() < 1.0
    # cleared pixels (main.cpp:482-487): colour = clear colour, position/normal = (0,0,0,1), rough = 0
    assert np.all(g["color"][~hit] == np.array([0.2, 0.3, 0.4, 1.0], np.float32))
    assert np.all(g["position"][~hit] == np.array([0, 0, 0, 1], np.float32)) and np.all(g["normal"][~hit] == np.array([0, 0, 0, 1], np.float32))
    assert np.all(g["roughMetal"][~hit] == 0)
    assert np.allclose(np.linalg.norm(g["normal"][hit][:, :3], axis=1), 1.0, atol=1e-5)
    # albedo = (1 - metalness) * baseColor is carried in the three w channels; rough/metal survive an rg16f round trip
    rm = g["roughMetal"][hit]
    assert np.array_equal(rm, rm.astype(np.float16).astype(np.float32)) and rm.min() >= 0 and rm.max() <= 1
    albedo = np.stack([g["color"][..., 3], g["position"][..., 3], g["normal"][..., 3]], -1)[hit]
    assert albedo.min() >= 0 and albedo.max() <= 1.0
    # brute force == BVH
    rows = np.arange(0, H, 8, dtype=np.uint32)
    gb = cornell_oracle.gbuffer(cam, W, H, lights_count=1, clear_color=(0.2, 0.3, 0.4, 1.0), rows=rows, use_bvh=False)
    for k in g:
        assert np.array_equal(g[k][rows].view(np.uint32), gb[k].view(np.uint32)), k


def test_oracle_hybrid_semantics(cornell_oracle):
    W, H = 64, 48
    cam = default_camera(W, H)
    g = cornell_oracle.gbuffer(cam, W, H, lights_count=1)
    hit = np.any(g["position"][..., :3] != 0, axis=-1)
    pc = make_push_constants(samples=1, depth=3, frame=0, lights_count=1)
    pc.useShadows, pc.useAO, pc.useGI = 0, 0, 0
    a, c = cornell_oracle.hybrid(pc, cam, W, H, g, seed=3)
    assert np.all(a == np.array([0, 0, 0, 1], np.float32)) and c["rays_shadow"] == 0 and c["rays_closest"] == 0  # nothing enabled
    pc.useShadows = 1
    a, c = cornell_oracle.hybrid(pc, cam, W, H, g, seed=3)
    assert set(np.unique(a[hit][:, 3])) <= {np.float32(0.01), np.float32(1.0)} and np.all(a[~hit] == np.array([0, 0, 0, 1], np.float32))
    assert 0 < c["rays_shadow"] <= hit.sum() and np.all(a[..., :3] == 0)
    pc.useAO = 1
    a2, c2 = cornell_oracle.hybrid(pc, cam, W, H, g, seed=3)
    assert c2["rays_shadow"] == c["rays_shadow"] + 4 * hit.sum()  # 4 AO rays per shaded pixel (rgen:31,140)
    ratios = np.unique(np.round(a2[hit][:, 3] / np.maximum(a[hit][:, 3], 1e-9), 4))
    assert set(ratios) <= {0.0, 0.25, 0.5, 0.75, 1.0}
    pc.useGI = 1
    a3, c3 = cornell_oracle.hybrid(pc, cam, W, H, g, seed=3)
    assert c3["rays_closest"] > 0 and np.any(a3[..., :3] > 0) and np.array_equal(a3[..., 3], a2[..., 3])
    # frame accumulation blends all four channels (rgen:36-48)
    pc.frame = 1
    old = np.full((H, W, 4), 0.5, np.float32)
    b, _ = cornell_oracle.hybrid(pc, cam, W, H, g, seed=3, accum=old.copy())
    assert np.allclose(b, 0.5 * 0.5 + a3 * 0.5, atol=1e-6)
    # brute force == BVH
    pc.frame = 0
    bb, _ = cornell_oracle.hybrid(pc, cam, W, H, g, seed=3, use_bvh=False)
    assert np.array_equal(a3.view(np.uint32), bb.view(np.uint32))


def test_oracle_post_composite():
    rng = np.random.default_rng(2)
    m, r = rng.uniform(0, 1, (5, 7, 4)).astype(np.float32), rng.uniform(0, 1, (5, 7, 4)).astype(np.float32)
    out = oracle_py.post(m, r, rt_mode=0)
    want = np.concatenate([m[..., :3] * r[..., 3:4] + r[..., :3], np.ones((5, 7, 1), np.float32)], -1) ** (1 / 2.2)
    assert np.allclose(out, want, rtol=1e-6)
    assert np.allclose(oracle_py.post(m, None, rt_mode=1), m ** (1 / 2.2), rtol=1e-6)
    assert np.allclose(oracle_py.post(m, r, 0, 1, 0)[..., :3], (r[..., 3:4] ** (1 / 2.2)).repeat(3, -1), rtol=1e-6)


# ---- GPU parity -----------------------------------------------------------------------------------------------------
def _mismatch(a, b):
    return float(np.mean(np.any(np.ascontiguousarray(a).view(np.uint32) != np.ascontiguousarray(b).view(np.uint32), axis=-1)))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["sah", "lbvh"])
def test_gpu_gbuffer_and_hybrid_match_oracle_cornell(cornell_flat, cornell_oracle, kind):
    import torch
    from vkrt_amd.renderer import Renderer

    W, H = 320, 200
    cam = default_camera(W, H)
    r = Renderer(cornell_flat, device=0, build=kind)
    g = r.gbuffer_raycast(cam, W, H, lights_count=1, clear_color=(0.1, 0.2, 0.3, 1.0))
    gref = cornell_oracle.gbuffer(cam, W, H, lights_count=1, clear_color=(0.1, 0.2, 0.3, 1.0))
    for k in gref:
        assert _mismatch(g[k].cpu().numpy(), gref[k]) < 1e-4, k
    accum = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    ref = np.zeros((H, W, 4), np.float32)
    gnp = {k: v.cpu().numpy() for k, v in g.items()}
    for f in range(3):
        pc = make_push_constants(samples=1, depth=4, frame=f, lights_count=1)
        pc.useShadows, pc.useAO, pc.useGI = 1, 1, 1
        r.reset_counters()
        r.hybrid_trace(pc, cam, W, H, g, seed=40 + f, accum=accum)
        c = r.counters()
        _, cref = cornell_oracle.hybrid(pc, cam, W, H, gnp, seed=40 + f, accum=ref)
        assert abs(c["rays_shadow"] - cref["rays_shadow"]) <= 8 and abs(c["rays_closest"] - cref["rays_closest"]) <= 8
    out = accum.cpu().numpy()
    assert _mismatch(out, ref) < 1e-4
    assert float(np.sqrt(np.mean((out - ref) ** 2))) < 1e-3
    # post composite (gamma uses powf: library vs device differ by rounding, not bit-exact)
    comp = r.post(g["color"], accum, rt_mode=0).cpu().numpy()
    want = oracle_py.post(gnp["color"], ref, rt_mode=0)
    bad = np.abs(comp - want) > (2e-5 * np.abs(want) + 1e-6)
    assert bad.any(axis=-1).mean() < 2e-4  # only the few path-divergent pixels allowed above
    r.close()


@pytest.mark.gpu
def test_gpu_hybrid_textured_atrium_and_toggles():
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import atrium
    from vkrt_amd.renderer import Renderer

    flat, info = atrium.build_atrium(20000, seed=3, with_textures=True)
    W, H = 256, 144
    cam = default_camera(W, H, **atrium.DEFAULT_CAMERA)
    orc = oracle_py.OracleScene(flat)
    r = Renderer(flat, device=0, build="sah")
    g = r.gbuffer_raycast(cam, W, H)  # all 8 fallback lights, clear colour (1,1,1,1)
    gref = orc.gbuffer(cam, W, H, lights_count=len(flat.lights))
    for k in gref:
        assert _mismatch(g[k].cpu().numpy(), gref[k]) < 2e-4, k
    gnp = {k: v.cpu().numpy() for k, v in g.items()}
    for sh, ao, gi in ((1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 1)):
        pc = make_push_constants(samples=1, depth=3, frame=0, lights_count=len(flat.lights))
        pc.useShadows, pc.useAO, pc.useGI = sh, ao, gi
        out = r.hybrid_trace(pc, cam, W, H, g, seed=5).cpu().numpy()
        ref, _ = orc.hybrid(pc, cam, W, H, gnp, seed=5)
        assert _mismatch(out, ref) < 2e-4, (sh, ao, gi)
    # the wave-synchronous kernel with work sharing (default) and the one-lane-one-walk kernel are the same function of the pixel
    from vkrt_amd import abi
    pc = make_push_constants(samples=1, depth=6, frame=0, lights_count=len(flat.lights))
    pc.useShadows, pc.useAO, pc.useGI = 1, 1, 1
    shared = r.hybrid_trace(pc, cam, W, H, g, seed=11).cpu().numpy()
    r.close()
    for opts in ({abi.VKRT_OPT_WF_SHARE: 0}, {abi.VKRT_OPT_BVH_LAYOUT: 0}, {abi.VKRT_OPT_MODE: 0}):  # MODE 0: the whole rgen in one kernel (no streams)
        r2 = Renderer(flat, device=0, build="sah", options=opts)
        alone = r2.hybrid_trace(pc, cam, W, H, g, seed=11).cpu().numpy()
        r2.close()
        assert np.array_equal(shared.view(np.uint32), alone.view(np.uint32)), opts


@pytest.mark.gpu
def test_gpu_hybrid_sharded_equals_whole_frame(small_scene):
    """The hybrid passes under image-strip sharding (vkrt_shard): each shard's G-buffer, accumulated plane (two progressive frames,
    GI paths on the wavefront streams) and NRD planes are the rows of the unsharded result, bit for bit."""
    from vkrt_amd import abi
    from vkrt_amd.flat_scene import make_push_constants
    from vkrt_amd.renderer import Renderer

    flat, camkw = small_scene
    W, H = 200, 150  # not a multiple of the 8x8 tiles or of the 16-row strips
    cam = default_camera(W, H, **camkw)
    vm = _view_matrix(camkw)
    L = len(flat.lights)
    r = Renderer(flat, device=0)

    def frames(shard):
        g = r.gbuffer_raycast(cam, W, H, lights_count=L, view_matrix=vm, shard=shard)
        acc = None
        for f in range(2):
            pc = make_push_constants(samples=1, depth=5, frame=f, lights_count=L)
            pc.useShadows, pc.useAO, pc.useGI = 1, 1, 1
            acc = r.hybrid_trace(pc, cam, W, H, g, seed=20 + f, accum=acc, shard=shard)
        return {**{k: v.cpu().numpy() for k, v in g.items()}, "accum": acc.cpu().numpy()}

    whole = frames(None)
    for count, index in ((3, 0), (3, 2), (4, 1)):
        part = frames(abi.Shard(W, H, 16, count, index))
        rows = [y for y in range(H) if (y // 16) % count == index]
        for k, v in part.items():
            assert v.shape[0] == len(rows), k
            assert np.array_equal(v.view(np.uint32), whole[k][rows].view(np.uint32)), (count, index, k)
    r.close()


# ---- NRD / REBLUR front-end planes (gltf.glsl:156-273) -------------------------------------------------
@pytest.fixture(scope="module")
def small_scene():
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import atrium

    flat, _ = atrium.build_atrium(6000, seed=3, with_textures=True, variant="emissive_mixed_lights")
    return flat, atrium.DEFAULT_CAMERA


def _view_matrix(camkw):
    import camera_np

    V = camera_np.look_at(camkw.get("eye", (0, 0, 15)), camkw.get("center", (0, 0, 0)), camkw.get("up", (0, 1, 0)))
    return np.asarray(V, np.float32).T.reshape(-1).copy()  # column-major


def _decode_oct(p):
    """_NRD_DecodeUnitVector(p, false, true), gltf.glsl:179-190"""
    q = p * 2.0 - 1.0
    n = np.stack([q[..., 0], q[..., 1], 1.0 - np.abs(q[..., 0]) - np.abs(q[..., 1])], -1)
    t = np.clip(-n[..., 2], 0.0, 1.0)
    n[..., 0] -= t * np.where(n[..., 0] >= 0, 1.0, -1.0)
    n[..., 1] -= t * np.where(n[..., 1] >= 0, 1.0, -1.0)
    return n / np.linalg.norm(n, axis=-1, keepdims=True)


def test_oracle_nrd_planes_decode_back_to_the_gbuffer(small_scene):
    """Property test of the packing on the CPU oracle: the oct-encoded normal decodes to the G-buffer normal within the
    rgb10 quantisation, roughness / viewZ match their sources, the packed radiance unpacks (YCoCg -> linear) to the GI term."""
    import oracle_py
    from vkrt_amd.flat_scene import make_push_constants

    flat, camkw = small_scene
    W, H = 160, 90
    cam = default_camera(W, H, **camkw)
    orc = oracle_py.OracleScene(flat)
    L = len(flat.lights)
    g = orc.gbuffer_nrd(cam, _view_matrix(camkw), W, H, lights_count=L)
    g0 = orc.gbuffer(cam, W, H, lights_count=L)
    for k in g0:
        assert np.array_equal(g[k].view(np.uint32), g0[k].view(np.uint32)), k
    hit = np.any(g["position"][..., :3] != 0, axis=-1)
    assert 0.5 < hit.mean() <= 1.0
    n = _decode_oct(g["nrdNormalRoughness"][..., :2].astype(np.float64))
    assert np.abs(n[hit] - g["normal"][..., :3][hit]).max() < 4e-3  # 10-bit oct encoding
    assert np.abs(g["nrdNormalRoughness"][..., 2][hit] - g["roughMetal"][..., 0][hit]).max() <= 0.5 / 1023 + 1e-3
    assert np.all(g["nrdNormalRoughness"][~hit] == 0) and np.all(g["nrdViewZ"][~hit] == 0)
    eye = np.asarray(camkw.get("eye", (0, 0, 15)), np.float64)
    fwd = np.asarray(camkw.get("center", (0, 0, 0)), np.float64) - eye
    fwd /= np.linalg.norm(fwd)
    depth = -((g["position"][..., :3].astype(np.float64) - eye) @ fwd)  # right-handed view space looks down -z
    assert np.abs(g["nrdViewZ"][hit] - depth[hit]).max() < 0.02 * np.abs(depth[hit]).max()  # r16f
    pc = make_push_constants(samples=1, depth=5, frame=0, lights_count=L)
    pc.useShadows, pc.useAO, pc.useGI = 1, 1, 1
    acc, rad = orc.hybrid_nrd(pc, cam, W, H, g, seed=9)
    acc0, _ = orc.hybrid(pc, cam, W, H, g0, seed=9)
    assert np.array_equal(acc.view(np.uint32), acc0.view(np.uint32))
    Y, Co, Cg = rad[..., 0].astype(np.float64), rad[..., 1].astype(np.float64), rad[..., 2].astype(np.float64)
    t = Y - Cg
    lin = np.maximum(np.stack([t + Co, Y + Cg, t - Co], -1), 0.0)  # _NRD_YCoCgToLinear, gltf.glsl:215-225
    want = np.clip(acc[..., :3].astype(np.float64), 0, 65504.0)
    assert np.abs(lin[hit] - want[hit]).max() <= 2e-3 * max(1.0, want[hit].max())  # three half-precision stores
    assert np.all((rad[..., 3] >= 0) & (rad[..., 3] <= 1)) and (rad[..., 3][hit] > 0).mean() > 0.3 and np.all(rad[~hit] == 0)


@pytest.mark.gpu
def test_gpu_nrd_planes_match_oracle(small_scene):
    import oracle_py
    from vkrt_amd.flat_scene import make_push_constants
    from vkrt_amd.renderer import Renderer

    flat, camkw = small_scene
    W, H = 256, 144
    cam = default_camera(W, H, **camkw)
    vm = _view_matrix(camkw)
    L = len(flat.lights)
    orc = oracle_py.OracleScene(flat)
    r = Renderer(flat, device=0, build="sah")
    g = r.gbuffer_raycast(cam, W, H, lights_count=L, view_matrix=vm)
    go = orc.gbuffer_nrd(cam, vm, W, H, lights_count=L)
    for k in ("nrdNormalRoughness", "nrdViewZ"):
        got = g[k].cpu().numpy()
        assert np.mean(got.view(np.uint32) != go[k].view(np.uint32)) < 2e-3, k   # same arithmetic; a rare quantisation-boundary flip
        assert np.abs(got - go[k]).max() <= (1.5 / 1023 if k == "nrdNormalRoughness" else 0.05)
    pc = make_push_constants(samples=1, depth=5, frame=0, lights_count=L)
    pc.useShadows, pc.useAO, pc.useGI = 1, 1, 1
    acc = r.hybrid_trace(pc, cam, W, H, g, seed=9)
    gnp = {k: v.cpu().numpy() for k, v in g.items()}
    acc_o, rad_o = orc.hybrid_nrd(pc, cam, W, H, gnp, seed=9)
    rad = g["nrdRadianceHitDist"].cpu().numpy()
    r.close()
    assert np.mean(np.any(acc.cpu().numpy().view(np.uint32) != acc_o.view(np.uint32), axis=-1)) < 1e-3
    assert np.mean(np.any(rad.view(np.uint32) != rad_o.view(np.uint32), axis=-1)) < 5e-3   # exp2f differs in the last bits between CPU and GPU
    assert np.abs(rad - rad_o).max() < 5e-3 * max(1.0, np.abs(rad_o).max())
