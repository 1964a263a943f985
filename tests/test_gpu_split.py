"""Triangle pre-splitting in the device builders (VKRT_OPT_SPLIT_BUDGET; the reference asks its driver for PREFER_FAST_TRACE,
hello_vulkan.cpp:1010, :1046).  Only references multiply: rays, pixels, counters of rays and the tie rule are those of the
unsplit tree; the tree itself must cover every triangle through the boxes above its slots (vkrt_debug_check_accel)."""
import hashlib
import os
import sys

import numpy as np
import pytest

from conftest import default_camera

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def sound(chk, info):
    assert chk["triangles_missing"] == 0 and chk["triangles_repeated"] == 0 and chk["box_violations"] == 0 and chk["bad_references"] == 0, chk
    assert chk["triangles_uncovered"] == 0, chk
    assert chk["triangles_referenced"] == info["reference_count"] and 0 < chk["nodes_reached"] <= info["node_count"]  # (BVH2 from the radix tree: nodes inside collapsed leaves are not part of the tree)


@pytest.fixture(scope="module")
def sponza_like():
    import atrium

    flat, info = atrium.build_atrium(60000, seed=2, with_textures=True, variant="nonuniform")
    return flat, atrium.DEFAULT_CAMERA


@pytest.mark.parametrize("kind", ["ploc", "lbvh"])
def test_split_references_keep_every_pixel_and_cover_every_triangle(sponza_like, kind):
    """Budgets 0 / 10 / 30 / 100 % on Sponza-like tessellation (room-sized walls, 15-m needles, drapery strips): image hash, ray
    counts and closest-hit (t, u, v, triangle id) of 60 k rays are those of the unsplit tree and of the oracle; the reference count
    stays within the budget; the validator finds every triangle covered."""
    import oracle_py
    from vkrt_amd import abi
    from vkrt_amd.flat_scene import make_push_constants
    from vkrt_amd.renderer import Renderer

    flat, camkw = sponza_like
    W, H = 320, 180
    cam = default_camera(W, H, **camkw)
    lights = len(flat.lights)
    rng = np.random.default_rng(5)
    lo, hi = flat.positions.min(0), flat.positions.max(0)
    o = rng.uniform(lo, hi, (60000, 3)).astype(np.float32)
    d = rng.standard_normal((60000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    orc = oracle_py.OracleScene(flat)
    tr, ur, vr, gr, _ = orc.trace_rays(o, d)
    ar = orc.trace_rays(o, d, tmax=3.0, any_hit=True)[3] >= 0
    results = {}
    for budget in (0, 10, 30, 100):
        r = Renderer(flat, device=0, build=kind, options={abi.VKRT_OPT_SPLIT_BUDGET: budget})
        info = r.accel_info()
        chk = r.check_accel()
        sound(chk, info)
        T = info["triangle_count"]
        assert T == flat.instanced_triangle_count
        assert T <= info["reference_count"] <= T + T * budget // 100
        if budget == 0:
            assert info["reference_count"] == T and chk["triangles_split"] == 0
        else:
            assert chk["triangles_split"] > 0 and info["reference_count"] > T
        img = None
        for f in range(2):
            img = r.pathtrace(make_push_constants(samples=2, depth=6, frame=f, lights_count=lights), cam, W, H, seed=3 + f, image=img)
        c = r.counters()
        assert c["traversal_faults"] == 0
        t, u, v, g = r.trace_rays(o, d)
        a = r.trace_rays(o, d, tmax=3.0, any_hit=True)[3] >= 0
        assert np.array_equal(g, gr) and np.array_equal(t[g >= 0].view(np.uint32), tr[gr >= 0].view(np.uint32))
        assert np.array_equal(u[g >= 0].view(np.uint32), ur[gr >= 0].view(np.uint32)) and np.array_equal(v[g >= 0].view(np.uint32), vr[gr >= 0].view(np.uint32))
        assert np.array_equal(a, ar)
        results[budget] = (hashlib.sha256(img.cpu().numpy().tobytes()).hexdigest(), c["rays_closest"], c["rays_shadow"], info["reference_count"], chk["triangles_split"])
        r.close()
    assert len({v[:3] for v in results.values()}) == 1, results
    assert results[10][3] < results[30][3] < results[100][3]


def test_split_cornell_config1_bit_identical_with_every_budget(cornell_flat, cornell_oracle):
    """BASELINE config 1 (Cornell 256^2, 1 spp, depth 1): two triangles per wall, each split into as many pieces as the budget gives;
    with the watertight test and the any-hit stage compiled in as well."""
    from vkrt_amd import abi
    from vkrt_amd.flat_scene import make_push_constants
    from vkrt_amd.renderer import Renderer

    W = H = 256
    cam = default_camera(W, H)
    pc = make_push_constants(samples=1, depth=1, frame=0, lights_count=1)
    ref, _ = cornell_oracle.render(pc, cam, W, H, seed=11)
    for opts in ({abi.VKRT_OPT_SPLIT_BUDGET: 100}, {abi.VKRT_OPT_SPLIT_BUDGET: 50, abi.VKRT_OPT_BVH_LAYOUT: 0},
                 {abi.VKRT_OPT_SPLIT_BUDGET: 100, abi.VKRT_OPT_MODE: 0}):
        for kind in ("ploc", "lbvh"):
            r = Renderer(cornell_flat, device=0, build=kind, options=opts)
            info = r.accel_info()
            sound(r.check_accel(), info)
            assert info["reference_count"] > info["triangle_count"]
            img = r.pathtrace(pc, cam, W, H, seed=11).cpu().numpy()
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (opts, kind)
            r.close()
    # the host builder ignores the option
    r = Renderer(cornell_flat, device=0, build="sah", options={abi.VKRT_OPT_SPLIT_BUDGET: 100})
    info = r.accel_info()
    assert info["reference_count"] == info["triangle_count"]
    r.close()


def test_split_with_dissolve_and_watertight_and_hybrid(sponza_like):
    """The options that change the records (exact vertices, the any-hit flag in the id word) and the hybrid passes on a split tree:
    identical to the unsplit tree of the same options."""
    from vkrt_amd import abi
    from vkrt_amd.flat_scene import make_push_constants
    from vkrt_amd.renderer import Renderer
    import copy

    flat, camkw = sponza_like
    flat = copy.deepcopy(flat)
    for k, m in enumerate(flat.materials):
        if k % 3 == 0:
            m["pbrBaseColorFactor"][3] = 0.4
    W, H = 256, 144
    cam = default_camera(W, H, **camkw)
    lights = len(flat.lights)
    out = {}
    for budget in (0, 40):
        r = Renderer(flat, device=0, build="ploc", options={abi.VKRT_OPT_SPLIT_BUDGET: budget, abi.VKRT_OPT_WATERTIGHT: 1, abi.VKRT_OPT_ANYHIT_DISSOLVE: 1})
        sound(r.check_accel(), r.accel_info())
        img = r.pathtrace(make_push_constants(samples=2, depth=5, frame=0, lights_count=lights), cam, W, H, seed=8)
        g = r.gbuffer_raycast(cam, W, H)
        pc = make_push_constants(samples=1, depth=4, frame=0, lights_count=lights)
        pc.useShadows, pc.useAO, pc.useGI = 1, 1, 1
        acc = r.hybrid_trace(pc, cam, W, H, g, seed=8)
        out[budget] = [hashlib.sha256(x.cpu().numpy().tobytes()).hexdigest() for x in (img, g["color"], g["position"], g["normal"], acc)]
        assert r.counters()["traversal_faults"] == 0
        r.close()
    assert out[0] == out[40]


def test_automatic_budget_splits_a_rotated_building_and_leaves_an_aligned_one_alone():
    """VKRT_OPT_SPLIT_BUDGET = -1 (round 5): the device builders build with a 30 % budget and without and keep the split tree only when
    its SAH cost is below 0.9 of the unsplit one.  On the Sponza-like tessellation turned 35 / 20 degrees about y / x (room-sized
    diagonal triangles: profiles/r05_split_rotated.jsonl, +97 % ray rate at 30 %) that resolves to 30; on the same building aligned with
    the axes to 0.  Pixels are the unsplit tree's either way."""
    import atrium
    from vkrt_amd import abi
    from vkrt_amd.flat_scene import make_push_constants
    from vkrt_amd.renderer import Renderer

    W, H = 320, 200
    for rot, want in ((None, 0), ((35.0, 20.0), 30)):
        flat, info = atrium.build_atrium(60000, seed=3, variant="nonuniform")
        camkw = dict(atrium.DEFAULT_CAMERA)
        if rot:
            camkw = atrium.rotate_scene(flat, camkw, *rot)
        cam = default_camera(W, H, **camkw)
        lights = len(flat.lights)
        hashes, refs = {}, {}
        for budget in (0, -1):
            for kind in ("ploc", "lbvh"):
                r = Renderer(flat, device=0, build=None, options={abi.VKRT_OPT_SPLIT_BUDGET: budget})
                r.build(kind)
                assert r.get_option(abi.VKRT_OPT_SPLIT_BUDGET) == budget  # the option keeps what the caller set
                got = r.get_option(abi.VKRT_INFO_SPLIT_BUDGET)
                a = r.accel_info()
                if budget == -1:
                    assert got == want, (rot, kind, got, a)
                    assert (a["reference_count"] > a["triangle_count"]) == (want > 0)
                    # any-hit child order (automatic): room-sized triangles that enter the tree whole switch far-first off; split into
                    # references they do not (profiles/r05_experiments.md #144)
                    assert r.get_option(abi.VKRT_INFO_ANYHIT_ORDER) == (4 if want else 0), (rot, kind)
                else:
                    assert got == 0 and a["reference_count"] == a["triangle_count"]
                img = r.pathtrace(make_push_constants(samples=2, depth=5, frame=0, lights_count=lights), cam, W, H, seed=11)
                c = r.counters()
                assert c["traversal_faults"] == 0
                hashes[(budget, kind)] = (hashlib.sha256(img.cpu().numpy().tobytes()).hexdigest(), c["rays_closest"], c["rays_shadow"])
                refs[(budget, kind)] = a["sah_cost"]
                r.close()
        assert len(set(hashes.values())) == 1, hashes
        if want:
            assert refs[(-1, "ploc")] < 0.9 * refs[(0, "ploc")]
    # the host builder does not split: automatic means 0 there
    r = Renderer(flat, device=0, build=None, options={abi.VKRT_OPT_SPLIT_BUDGET: -1})
    r.build("sah")
    assert r.get_option(abi.VKRT_INFO_SPLIT_BUDGET) == 0
    r.close()
