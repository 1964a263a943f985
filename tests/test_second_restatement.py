"""oracle.cpp (CPU tests) and the HIP kernels (GPU tests) against the SECOND restatement of the path.

tests/golden/np_pathtrace_fixtures.npz holds outputs of oracle/np_pathtrace.py -- the rgen / rchit / rmiss / hybrid
control flow written a second time from the GLSL, in numpy, with brute-force float64 ray queries (generator:
tests/golden/make_np_pathtrace_fixtures.py).  The reference ships no golden images (parity unpinned), so
this is the pin available: two independent readings of the shaders must produce the same pictures.

The two sides differ in operation order, in sin / cos / pow and in the ray-triangle arithmetic, so agreement is to
rounding, not bit for bit; and path tracing amplifies rounding (a 1e-7 difference in a hit point grows ~100x per glossy
bounce), so deep paths agree per pixel for most pixels only.  Bars per case (`BARS`): shallow cases (depth <= 2,
Cornell) pin every pixel; deep cases pin the ray counts, the image mean and the bulk of the pixels.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
FIXTURES = os.path.join(ROOT, "tests", "golden", "np_pathtrace_fixtures.npz")

# case -> (fraction of pixels that must agree to REL, REL, max |mean difference| / mean, max relative ray-count difference, rmse bound)
BARS = {
    "cornell_c1": (1.0, 1e-4, 1e-5, 0.0, 1e-6),
    "cornell_c2": (0.999, 1e-4, 1e-4, 0.0, 1e-4),
    "atrium_d2": (0.98, 2e-3, 2e-3, 1e-3, 1e-3),           # rmse bar = the north star's 1e-3
    "atrium_emissive_d2": (0.92, 2e-3, 5e-3, 1e-3, 1e-3),  # striped emissive texture: a 2e-5 difference in uv moves the texel blend by ~1e-3
    "atrium_d8": (0.80, 2e-3, 2e-2, 5e-3, 2e-3),
    "atrium_emissive": (0.70, 2e-3, 3e-2, 5e-3, 3e-2),
    "hybrid_d2": (0.995, 1e-4, 1e-4, 0.0, 1e-5),
    "hybrid": (0.80, 2e-3, 2e-2, 2e-3, 1e-2),
}


def agreement(got, want):
    g = np.asarray(got, np.float64).reshape(-1, got.shape[-1])
    w = np.asarray(want, np.float64).reshape(-1, want.shape[-1])
    d = np.abs(g - w)
    rel = (d / np.maximum(np.abs(w), 1e-2)).max(1)
    return {"rel": rel, "rmse": float(np.sqrt((d[:, :3] ** 2).mean())), "mean_rel": float(abs(g[:, :3].mean() - w[:, :3].mean()) / max(w[:, :3].mean(), 1e-6))}


def check(name, got, want, rays_got=None, rays_want=None):
    frac, rel, mean_tol, ray_tol, rmse_tol = BARS[name]
    a = agreement(got, want)
    ok = float((a["rel"] <= rel).mean())
    assert ok >= frac, f"{name}: only {ok:.4f} of the pixels within {rel} (need {frac})"
    assert a["mean_rel"] <= mean_tol, f"{name}: image mean differs by {a['mean_rel']:.2e}"
    assert a["rmse"] <= rmse_tol, f"{name}: rmse {a['rmse']:.2e}"
    if rays_got is not None:
        for g, w in zip(rays_got, rays_want):
            assert abs(int(g) - int(w)) <= ray_tol * int(w), f"{name}: ray counts {list(rays_got)} vs {list(rays_want)}"


@pytest.fixture(scope="module")
def fx():
    return np.load(FIXTURES)


@pytest.fixture(scope="module")
def cases():
    import make_np_pathtrace_fixtures as mk

    return mk


def _scene(mk, cache, name):
    if name not in cache:
        cache[name] = mk.load_scene(name)
    return cache[name]


def _camera(W, H, camkw):
    import camera_np
    from vkrt_amd.flat_scene import uniforms_from_matrices

    return uniforms_from_matrices(*camera_np.global_uniforms(width=W, height=H, **camkw))


# ---- CPU: the oracle against the second restatement ----------------------------------------------------------------
@pytest.mark.parametrize("name", ["cornell_c1", "cornell_c2", "atrium_d2", "atrium_emissive_d2", "atrium_d8", "atrium_emissive"])
def test_oracle_path_tracer_matches_second_restatement(fx, cases, name):
    import oracle_py
    from vkrt_amd.flat_scene import make_push_constants

    c = cases.CASES[name]
    flat, camkw = cases.load_scene(c["scene"])
    orc = oracle_py.OracleScene(flat)
    W, H = c["W"], c["H"]
    cam = _camera(W, H, camkw)
    rows = np.asarray(c["rows"], np.uint32)
    img = np.zeros((len(rows), W, 4), np.float32)
    rays = [0, 0]
    for f in range(c["frames"]):
        pc = make_push_constants(samples=c["samples"], depth=c["depth"], frame=f, lights_count=len(flat.lights))
        _, cnt = orc.render(pc, cam, W, H, seed=c["seed0"] + f, rows=rows, image=img)
        rays[0] += cnt["rays_closest"]
        rays[1] += cnt["rays_shadow"]
    check(name, img, fx[name + "/image"], rays, fx[name + "/rays"])


@pytest.mark.parametrize("name", ["hybrid_d2", "hybrid"])
def test_oracle_hybrid_matches_second_restatement(fx, cases, name):
    """frag_shader.frag (G-buffer planes) and raytraceHybrid.rgen; the rgen is driven from the FIXTURE's G-buffer so the
    two halves are compared separately."""
    import oracle_py
    from vkrt_amd.flat_scene import make_push_constants

    c = cases.HYBRID[name]
    flat, camkw = cases.load_scene(c["scene"])
    orc = oracle_py.OracleScene(flat)
    W, H = c["W"], c["H"]
    cam = _camera(W, H, camkw)
    rows = np.asarray(c["rows"], np.uint32)
    L = len(flat.lights)
    go = orc.gbuffer(cam, W, H, lights_count=L, rows=rows)
    for k, tol in (("color", 5e-3), ("position", 1e-3), ("normal", 5e-3), ("roughMetal", 1e-3)):
        a = agreement(go[k], fx[f"{name}/gbuffer_{k}"])
        assert (a["rel"] <= tol).mean() >= 0.995, (k, float((a["rel"] <= tol).mean()))
    g = {k: np.ascontiguousarray(fx[f"{name}/gbuffer_{k}"]) for k in go}
    acc, rays = None, [0, 0]
    for f in range(c["frames"]):
        pc = make_push_constants(samples=1, depth=c["depth"], frame=f, lights_count=L)
        pc.useShadows, pc.useAO, pc.useGI = 1, 1, 1
        acc, cnt = orc.hybrid(pc, cam, W, H, g, seed=c["seed0"] + f, rows=rows, accum=acc)
        rays[0] += cnt["rays_closest"]
        rays[1] += cnt["rays_shadow"]
    check(name, acc, fx[name + "/accum"], rays, fx[name + "/rays"])


def test_second_restatement_prng_matches_kat():
    """np_pathtrace's own tea / lcg against the committed integer KATs."""
    import json

    import np_pathtrace as npt

    kat = json.load(open(os.path.join(ROOT, "tests", "golden", "prng_kat.json")))
    for a, b, want in kat["tea"]:
        assert int(npt.tea(np.array([a], np.uint64), b)[0]) == want
    for chain in kat["lcg"]:
        s = np.array([chain["start"]], np.uint32)
        for state, bits in chain["seq"]:
            s, f = npt.rnd(s)
            assert int(s[0]) == state and int(round(float(f[0]) * 16777216.0)) == bits


# ---- GPU: the HIP path against the same fixtures ------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cornell_c1", "cornell_c2", "atrium_d2", "atrium_emissive_d2", "atrium_d8", "atrium_emissive"])
@pytest.mark.parametrize("kind", ["sah", "lbvh"])
def test_gpu_path_tracer_matches_second_restatement(fx, cases, name, kind):
    from vkrt_amd.flat_scene import make_push_constants
    from vkrt_amd.renderer import Renderer

    c = cases.CASES[name]
    flat, camkw = cases.load_scene(c["scene"])
    W, H = c["W"], c["H"]
    cam = _camera(W, H, camkw)
    r = Renderer(flat, device=0, build=kind)
    img = None
    for f in range(c["frames"]):
        pc = make_push_constants(samples=c["samples"], depth=c["depth"], frame=f, lights_count=len(flat.lights))
        img = r.pathtrace(pc, cam, W, H, seed=c["seed0"] + f, image=img)
    got = img.cpu().numpy()[np.asarray(c["rows"])]
    faults = r.counters()["traversal_faults"]
    r.close()
    assert faults == 0
    check(name, got, fx[name + "/image"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["hybrid_d2", "hybrid"])
def test_gpu_hybrid_matches_second_restatement(fx, cases, name):
    import torch
    from vkrt_amd.flat_scene import make_push_constants
    from vkrt_amd.renderer import Renderer

    c = cases.HYBRID[name]
    flat, camkw = cases.load_scene(c["scene"])
    W, H = c["W"], c["H"]
    cam = _camera(W, H, camkw)
    rows = np.asarray(c["rows"])
    L = len(flat.lights)
    r = Renderer(flat, device=0, build="sah")
    g = r.gbuffer_raycast(cam, W, H, lights_count=L)
    for k, tol in (("color", 5e-3), ("position", 1e-3), ("normal", 5e-3), ("roughMetal", 1e-3)):
        a = agreement(g[k].cpu().numpy()[rows], fx[f"{name}/gbuffer_{k}"])
        assert (a["rel"] <= tol).mean() >= 0.995, (k, float((a["rel"] <= tol).mean()))
    # drive the HIP rgen from the fixture's G-buffer rows (other rows: the GPU's own planes)
    for k in g:
        g[k][torch.from_numpy(rows).cuda()] = torch.from_numpy(np.ascontiguousarray(fx[f"{name}/gbuffer_{k}"])).cuda()
    acc = None
    for f in range(c["frames"]):
        pc = make_push_constants(samples=1, depth=c["depth"], frame=f, lights_count=L)
        pc.useShadows, pc.useAO, pc.useGI = 1, 1, 1
        acc = r.hybrid_trace(pc, cam, W, H, g, seed=c["seed0"] + f, accum=acc)
    got = acc.cpu().numpy()[rows]
    r.close()
    check(name, got, fx[name + "/accum"])
