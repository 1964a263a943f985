"""Implicit-LOD texture() of the hybrid mode's G-buffer (frag_shader.frag runs as a fragment shader in the reference, so its
texture() calls see the sampler of hello_vulkan.cpp:448-454: trilinear over the mip chain of hello_vulkan.cpp:499, anisotropy 4).

CPU: the mip chain (vkCmdBlitImage LINEAR per level) against hand-computed values and against the numpy restatement; the
sampler against explicit single-level samples where the Vulkan formulas make the answer obvious; oracle.cpp against the
independent numpy implementation on random coordinates and derivatives.
GPU: the HIP G-buffer against oracle.cpp bit for bit on a scene with non-power-of-two, non-square, sRGB and UNORM textures seen at
a grazing angle (anisotropy) -- with VKRT_OPT_GBUFFER_MIPS on (default) and off."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from conftest import default_camera  # noqa: E402


def _textures():
    rng = np.random.default_rng(11)
    sizes = [(37, 21, True), (64, 64, False), (8, 2, True), (5, 3, False), (1, 1, True), (128, 32, True)]  # (w, h, srgb)
    tex = []
    for w, h, srgb in sizes:
        img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        if w >= 32:  # some structure so minification visibly changes the colour: 4-texel checker on top of the noise
            yy, xx = np.mgrid[0:h, 0:w]
            img[..., :3] = np.where((((xx // 4) + (yy // 4)) & 1)[..., None] == 1, img[..., :3] // 4, 255 - img[..., :3] // 4)
        tex.append({"rgba8": np.ascontiguousarray(img), "is_srgb": srgb})
    return tex


def lod_scene():
    """A long floor strip (grazing view: anisotropic footprints, every LOD) and an upright wall, four materials using all six textures."""
    from vkrt_amd.flat_scene import LIGHT_DTYPE, MAT_DTYPE, NODE_DTYPE, PRIM_DTYPE, FlatScene

    def quad(p0, du, dv, n, t, uvscale):
        P = np.array([p0, p0 + du, p0 + du + dv, p0 + dv], np.float32)
        UV = np.array([[0, 0], [uvscale[0], 0], [uvscale[0], uvscale[1]], [0, uvscale[1]]], np.float32)
        return P, np.tile(np.asarray(n, np.float32), (4, 1)), np.tile(np.asarray(t, np.float32), (4, 1)), UV

    quads = [quad(np.array([-3, 0, 2], np.float32), np.array([3, 0, 0], np.float32), np.array([0, 0, -60], np.float32), [0, 1, 0], [1, 0, 0, 1], (3, 40)),
             quad(np.array([0, 0, 2], np.float32), np.array([3, 0, 0], np.float32), np.array([0, 0, -60], np.float32), [0, 1, 0], [1, 0, 0, 1], (1.5, 9)),
             quad(np.array([-3, 0, -12], np.float32), np.array([6, 0, 0], np.float32), np.array([0, 5, 0], np.float32), [0, 0, 1], [1, 0, 0, 1], (7, 5)),
             quad(np.array([-3, 0, 1], np.float32), np.array([0, 0, -13], np.float32), np.array([0, 4, 0], np.float32), [1, 0, 0], [0, 0, -1, 1], (20, 2))]
    pos = np.concatenate([q[0] for q in quads]); nrm = np.concatenate([q[1] for q in quads])
    tan = np.concatenate([q[2] for q in quads]); uv = np.concatenate([q[3] for q in quads])
    idx = np.array([0, 1, 2, 0, 2, 3], np.uint32)
    pm = np.zeros(4, PRIM_DTYPE)
    for k in range(4):
        pm[k] = (0, 6, 4 * k, 4, k)
    mats = np.zeros(4, MAT_DTYPE)
    slots = [(0, 1, 5, 2), (5, -1, -1, 3), (1, 0, -1, -1), (3, 2, 1, 0)]  # base, metallicRoughness, normal, emissive
    for k, (b, mr, nt, em) in enumerate(slots):
        mats[k]["pbrBaseColorFactor"] = [1.0, 0.9, 0.8, 1.0]
        mats[k]["pbrBaseColorTexture"], mats[k]["metallicRoughnessTexture"], mats[k]["normalTexture"], mats[k]["emissiveTexture"] = b, mr, nt, em
        mats[k]["metallicFactor"], mats[k]["roughnessFactor"] = 0.8, 0.9
        mats[k]["emissiveFactor"] = [0.3, 0.2, 0.1]
    nodes = np.zeros(4, NODE_DTYPE)
    for k in range(4):
        nodes[k]["worldMatrix"] = np.eye(4, dtype=np.float32).ravel()
        nodes[k]["primMesh"] = k
    lights = np.zeros(2, LIGHT_DTYPE)
    lights[0] = ((0.5, 4.0, -3.0), (1, 1, 1), 40.0, 0)
    lights[1] = ((0.3, 1.0, 0.2), (1, 0.9, 0.8), 1.5, 1)
    return FlatScene(pos, nrm, tan, uv, idx, pm, mats, lights, nodes, _textures())


CAMERA = dict(eye=(0.4, 1.1, 1.5), center=(0.2, 0.6, -10.0), up=(0, 1, 0), fov=55.0)


@pytest.fixture(scope="module")
def scene():
    import oracle_py

    flat = lod_scene()
    return flat, oracle_py.OracleScene(flat)


# ---- mip chain ------------------------------------------------------------------------------------------------------------
def test_mip_chain_sizes_and_box_filter(scene):
    flat, orc = scene
    for ti, tx in enumerate(flat.textures):
        h, w = tx["rgba8"].shape[:2]
        chain = orc.texture_levels(ti)
        assert len(chain) == int(np.floor(np.log2(max(w, h)))) + 1                     # nvvk mipLevels: floor(log2(max(w, h))) + 1
        for L, img in enumerate(chain):
            assert img.shape[:2] == (max(1, h >> L), max(1, w >> L))
        assert np.array_equal(chain[0], tx["rgba8"])
    # UNORM, even sizes: level 1 is the rounded mean of each 2x2 block
    img = flat.textures[1]["rgba8"].astype(np.float64)
    want = np.floor((img[0::2, 0::2] + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2]) / 4.0 + 0.5)
    got = orc.texture_levels(1)[1].astype(np.float64)
    assert np.abs(got - want).max() <= 1 and np.mean(got != want) < 0.01               # exact-tie roundings may land either side
    # sRGB: colour is averaged on decoded values, alpha on UNORM values
    tx = flat.textures[5]["rgba8"]
    c = tx.astype(np.float64) / 255.0
    lin = np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)
    lin[..., 3] = c[..., 3]
    avg = (lin[0::2, 0::2] + lin[0::2, 1::2] + lin[1::2, 0::2] + lin[1::2, 1::2]) / 4.0
    enc = np.where(avg <= 0.0031308, 12.92 * avg, 1.055 * avg ** (1 / 2.4) - 0.055)
    enc[..., 3] = avg[..., 3]
    want = np.floor(enc * 255.0 + 0.5)
    got = orc.texture_levels(5)[1].astype(np.float64)
    assert np.abs(got - want).max() <= 1 and np.mean(got != want) < 0.01


def test_mip_chain_odd_sizes_blit_linear(scene):
    """5x3 UNORM -> 2x1: destination centres map to source x = 1.25 - 0.5, 3.75 - 0.5 and y = 1.5 - 0.5 (vkCmdBlitImage, LINEAR)."""
    flat, orc = scene
    src = flat.textures[3]["rgba8"].astype(np.float64)
    chain = orc.texture_levels(3)
    assert [c.shape[:2] for c in chain] == [(3, 5), (1, 2), (1, 1)]
    for x, sx in enumerate((0.75, 3.25)):
        x0, ax = int(np.floor(sx)), sx - np.floor(sx)
        want = src[1, x0] * (1 - ax) + src[1, x0 + 1] * ax                              # y = 1.0 exactly: row 1 only
        assert np.abs(chain[1][0, x].astype(np.float64) - np.floor(want + 0.5)).max() <= 1
    want = chain[1][0].astype(np.float64).mean(0)                                       # 2x1 -> 1x1: x = 0.5 between the two texels
    assert np.abs(chain[2][0, 0].astype(np.float64) - np.floor(want + 0.5)).max() <= 1


def test_mip_chain_matches_numpy_restatement(scene):
    import np_pathtrace as npt

    flat, orc = scene
    sc = npt.NpScene(flat)
    for ti in range(len(flat.textures)):
        a, b = orc.texture_levels(ti), sc.mip_chain(ti)
        assert len(a) == len(b)
        for x, y in zip(a, b):
            d = np.abs(x.astype(int) - y.astype(int))
            assert d.max() <= 1 and np.mean(d > 0) < 0.005                              # float32 vs float64 filtering: a rare rounding tie


# ---- the sampler ----------------------------------------------------------------------------------------------------------
def _level_sample(orc, flat, ti, level, uv):
    """bilinear REPEAT sample of one mip level through an auxiliary one-texture scene whose level 0 is that level"""
    import oracle_py
    from vkrt_amd.flat_scene import FlatScene

    lv = orc.texture_levels(ti)[level]
    aux = FlatScene(flat.positions, flat.normals, flat.tangents, flat.texcoords0, flat.indices, flat.prim_meshes, flat.materials, flat.lights,
                    flat.nodes, [{"rgba8": np.ascontiguousarray(lv), "is_srgb": flat.textures[ti]["is_srgb"]}])
    return oracle_py.OracleScene(aux).sample_texture(0, uv)


def test_sampler_known_lods(scene):
    flat, orc = scene
    rng = np.random.default_rng(2)
    uv = rng.uniform(-1.5, 2.5, (200, 2)).astype(np.float32)
    ti, (h, w) = 1, flat.textures[1]["rgba8"].shape[:2]                                 # 64x64 UNORM
    zero = np.zeros((200, 4), np.float32)
    # magnification and rho <= 1: level 0, one tap
    assert np.array_equal(orc.sample_texture_grad(ti, uv, zero), orc.sample_texture(ti, uv))
    g = np.tile(np.array([0.7 / w, 0, 0, 0.7 / h], np.float32), (200, 1))
    assert np.allclose(orc.sample_texture_grad(ti, uv, g), orc.sample_texture(ti, uv), atol=1e-6)
    # isotropic footprint of exactly 2^k texels: lambda = k, a single level
    for k in (1, 2, 3):
        g = np.tile(np.array([2.0 ** k / w, 0, 0, 2.0 ** k / h], np.float32), (200, 1))
        assert np.allclose(orc.sample_texture_grad(ti, uv, g), _level_sample(orc, flat, ti, k, uv), atol=2e-6), k
    # lambda = 1.5: halfway between levels 1 and 2
    g = np.tile(np.array([2.0 ** 1.5 / w, 0, 0, 2.0 ** 1.5 / h], np.float32), (200, 1))
    want = 0.5 * (_level_sample(orc, flat, ti, 1, uv) + _level_sample(orc, flat, ti, 2, uv))
    assert np.allclose(orc.sample_texture_grad(ti, uv, g), want, atol=2e-5)
    # anisotropy 4 along x with an 8-texel major axis: eta = 4, lambda = log2(8 / 4) = 1, four taps at u + (i / 5 - 1/2) * dudx
    g = np.tile(np.array([8.0 / w, 0, 0, 2.0 / h], np.float32), (200, 1))
    taps = [_level_sample(orc, flat, ti, 1, uv + np.array([(i / 5.0 - 0.5) * 8.0 / w, 0], np.float32)) for i in (1, 2, 3, 4)]
    assert np.allclose(orc.sample_texture_grad(ti, uv, g), np.mean(taps, 0), atol=2e-5)
    # ratio above maxAnisotropy: eta clamps to 4, lambda = log2(32 / 4) = 3; major axis y this time
    g = np.tile(np.array([1.0 / w, 0, 0, 32.0 / h], np.float32), (200, 1))
    taps = [_level_sample(orc, flat, ti, 3, uv + np.array([0, (i / 5.0 - 0.5) * 32.0 / h], np.float32)) for i in (1, 2, 3, 4)]
    assert np.allclose(orc.sample_texture_grad(ti, uv, g), np.mean(taps, 0), atol=2e-5)
    # a footprint larger than the texture, infinite and NaN derivatives: the 1x1 level
    last = orc.texture_levels(ti)[-1][0, 0].astype(np.float32) / 255.0
    for val in (1.0e4, np.inf, np.nan):
        g = np.tile(np.array([val, 0, 0, val], np.float32), (200, 1))
        assert np.allclose(orc.sample_texture_grad(ti, uv, g), last[None, :], atol=1e-6), val
    # a 1x1 texture and an invalid index
    assert np.allclose(orc.sample_texture_grad(4, uv, g), orc.sample_texture(4, uv))
    assert np.array_equal(orc.sample_texture_grad(99, uv, g), np.ones((200, 4), np.float32))


def test_sampler_matches_numpy_restatement(scene):
    import np_pathtrace as npt

    flat, orc = scene
    sc = npt.NpScene(flat)
    rng = np.random.default_rng(3)
    n = 4000
    uv = rng.uniform(-2, 3, (n, 2)).astype(np.float32)
    mag = (2.0 ** rng.uniform(-9, 1, (n, 1))).astype(np.float32)
    grad = (rng.normal(size=(n, 4)) * mag * np.array([1, 1, 0.3, 0.3])).astype(np.float32)
    grad[:50] = 0
    grad[50:100, 2:] = 0                                                                # rho_min = 0
    for ti in range(len(flat.textures)):
        a = orc.sample_texture_grad(ti, uv, grad)
        b = sc.texture_grad(np.full(n, ti), uv, grad)
        # the level choice flips when lambda or eta sit on an integer within float rounding: allow a handful
        assert np.mean(np.abs(a - b).max(1) > 2e-5) < 2e-3, (ti, float(np.abs(a - b).max()))


def test_oracle_gbuffer_lod_agrees_with_numpy_restatement(scene):
    import np_pathtrace as npt

    flat, orc = scene
    W, H = 160, 90
    cam = default_camera(W, H, **CAMERA)
    sc = npt.NpScene(flat)
    vi = np.asarray(cam.viewInverse.m[:], np.float32)
    pi = np.asarray(cam.projInverse.m[:], np.float32)
    rows = np.array([20, 47, 62, 80], np.uint32)
    xs, ys = np.tile(np.arange(W), len(rows)), np.repeat(rows.astype(np.int64), W)
    planes = {}
    for mips in (True, False):
        orc.set_gbuffer_mips(mips)
        go = orc.gbuffer(cam, W, H, lights_count=2, rows=rows)
        gn = npt.gbuffer_pixels(sc, (1.0, 1.0, 1.0, 1.0), 2, vi, pi, W, H, xs, ys, mips=mips)
        for k in go:
            d = np.abs(go[k] - gn[k].reshape(len(rows), W, -1))
            lim = 2e-3 * np.maximum(1.0, np.abs(go[k]))
            assert np.mean((d > lim).any(-1)) < 0.01, (mips, k, float(d.max()))
        planes[mips] = go
    orc.set_gbuffer_mips(True)
    hit = planes[True]["position"][..., :3].any(-1)
    assert hit.mean() > 0.5
    changed = np.abs(planes[True]["color"] - planes[False]["color"])[hit].max(-1) > 1e-3
    assert changed.mean() > 0.3                                                         # the floor is minified almost everywhere
    assert np.array_equal(planes[True]["position"][..., :3], planes[False]["position"][..., :3])


# ---- the HIP G-buffer -----------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["sah", "lbvh"])
def test_gpu_gbuffer_lod_matches_oracle(scene, kind):
    from vkrt_amd import abi
    from vkrt_amd.renderer import Renderer

    flat, orc = scene
    r = Renderer(flat, device=0, build=kind)
    assert r.get_option(abi.VKRT_OPT_GBUFFER_MIPS) == 1
    try:
        for W, H in ((320, 180), (161, 91)):                                            # odd sizes: the quad partner of the last column / row lies outside
            cam = default_camera(W, H, **CAMERA)
            out = {}
            for mips in (1, 0, 1):
                r.set_option(abi.VKRT_OPT_GBUFFER_MIPS, mips)
                orc.set_gbuffer_mips(mips)
                g = {k: v.cpu().numpy() for k, v in r.gbuffer_raycast(cam, W, H, lights_count=2).items()}
                want = orc.gbuffer(cam, W, H, lights_count=2)
                for k in want:
                    bad = (g[k].view(np.uint32) != want[k].view(np.uint32)).any(-1)
                    assert bad.mean() < 2e-4, (W, mips, k, float(bad.mean()), float(np.abs(g[k] - want[k]).max()))
                out[mips] = g
            assert np.abs(out[1]["color"] - out[0]["color"]).max() > 0.05
        # sharded launches see the same derivatives (global pixel coordinates)
        cam = default_camera(320, 180, **CAMERA)
        full = r.gbuffer_raycast(cam, 320, 180, lights_count=2)["color"].cpu().numpy()
        part = r.gbuffer_raycast(cam, 320, 180, lights_count=2, shard=abi.Shard(320, 180, 16, 3, 1))["color"].cpu().numpy()
        mine = [y for y in range(180) if (y // 16) % 3 == 1]                                # interleaved 16-row strips, shard 1 of 3
        assert part.shape[0] == len(mine) and np.array_equal(part, full[mine])
    finally:
        orc.set_gbuffer_mips(True)
        r.close()
