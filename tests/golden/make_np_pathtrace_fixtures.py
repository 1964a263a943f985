"""Writes tests/golden/np_pathtrace_fixtures.npz: outputs of the SECOND restatement of the path (oracle/np_pathtrace.py,
numpy, brute-force ray queries, written from the GLSL) on fixed pixel sets.  tests/test_second_restatement.py compares
oracle.cpp (CPU) and the HIP kernels (GPU) with these numbers; the file is regenerated only by running this script
(a few minutes of numpy):

    python tests/golden/make_np_pathtrace_fixtures.py

Cases (inputs are all seeded / shipped: tests/golden/cornell_flat.npz, tools/atrium.py):
  cornell_c1      cornell 256x256, 1 spp, depth 1, frame 0, seed 0 (BASELINE config 1), 8 rows = 2048 pixels
  cornell_c2      cornell 1280x720, depth 4, frames 0..2 x 1 spp, seed = frame (BASELINE config 2's progressive definition), 2 rows
  atrium_d2       8k-triangle textured atrium 320x180, 2 spp, depth 2, frames 0..1, 7 rows = 2240 pixels (per-pixel pin of
                  instancing, textures, normal maps, NEE, shadow rays and one bounce)
  atrium_d8       same, 1 spp, depth 8: the full loop.  Paths are chaotic (a 1e-7 difference grows ~100x per glossy bounce), so
                  deep paths agree per pixel only for most pixels; ray counts and image statistics pin the rest
  atrium_emissive_d2 / atrium_emissive: same geometry with emissive textures + point/directional/spot lights; depth 2 / depth 5
  hybrid_d2 / hybrid: G-buffer (frag_shader.frag) + raytraceHybrid.rgen shadows / AO / GI, depth 2 / depth 6, frames 0..1, 3 rows
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import vkrt_amd  # noqa: E402,F401
from vkrt_amd.flat_scene import FlatScene, make_push_constants  # noqa: E402
import atrium  # noqa: E402
import camera_np  # noqa: E402
import np_pathtrace as npt  # noqa: E402


def camera(W, H, **kw):
    """column-major viewInverse / projInverse (16 floats each) as GlobalUniforms stores them"""
    _, vi, pi = camera_np.global_uniforms(width=W, height=H, **kw)
    return np.asarray(vi, np.float32).T.reshape(-1).copy(), np.asarray(pi, np.float32).T.reshape(-1).copy()


def pixel_rows(W, rows):
    rows = np.asarray(rows, np.int64)
    return np.tile(np.arange(W), len(rows)), np.repeat(rows, W)


CASES = {
    "cornell_c1": dict(scene="cornell", W=256, H=256, samples=1, depth=1, frames=1, seed0=0, rows=[16, 48, 80, 112, 144, 176, 208, 240]),
    "cornell_c2": dict(scene="cornell", W=1280, H=720, samples=1, depth=4, frames=3, seed0=0, rows=[250, 470]),
    "atrium_d2": dict(scene="atrium", W=320, H=180, samples=2, depth=2, frames=2, seed0=40, rows=[12, 38, 64, 90, 116, 142, 168]),
    "atrium_d8": dict(scene="atrium", W=320, H=180, samples=1, depth=8, frames=2, seed0=20, rows=[12, 38, 64, 90, 116, 142, 168]),
    "atrium_emissive_d2": dict(scene="atrium_emissive", W=320, H=180, samples=2, depth=2, frames=1, seed0=50, rows=[30, 90, 150]),
    "atrium_emissive": dict(scene="atrium_emissive", W=320, H=180, samples=2, depth=5, frames=1, seed0=30, rows=[30, 90, 150]),
}
HYBRID = {"hybrid_d2": dict(scene="atrium_emissive", W=320, H=180, depth=2, frames=2, seed0=7, rows=[40, 95, 150]),
          "hybrid": dict(scene="atrium_emissive", W=320, H=180, depth=6, frames=2, seed0=3, rows=[40, 95, 150])}


def load_scene(name):
    if name == "cornell":
        return FlatScene.load_npz(os.path.join(ROOT, "tests", "golden", "cornell_flat.npz")), {}
    variant = "emissive_mixed_lights" if name == "atrium_emissive" else None
    flat, _ = atrium.build_atrium(8000, seed=3, with_textures=True, variant=variant)
    return flat, atrium.DEFAULT_CAMERA


def main():
    out = {}
    scenes = {}
    for name, c in list(CASES.items()) + list(HYBRID.items()):
        if c["scene"] not in scenes:
            flat, camkw = load_scene(c["scene"])
            scenes[c["scene"]] = (flat, camkw, npt.NpScene(flat))
        flat, camkw, sc = scenes[c["scene"]]
        W, H = c["W"], c["H"]
        vi, pi = camera(W, H, **camkw)
        xs, ys = pixel_rows(W, c["rows"])
        lights = len(flat.lights)
        t0 = time.time()
        sc.rays_closest = sc.rays_shadow = 0
        if name not in HYBRID:
            img = None
            for f in range(c["frames"]):
                pc = make_push_constants(samples=c["samples"], depth=c["depth"], frame=f, lights_count=lights)
                img = npt.pathtrace_pixels(sc, pc, vi, pi, W, H, c["seed0"] + f, xs, ys, old=img)
            out[name + "/image"] = img.reshape(len(c["rows"]), W, 4)
        else:
            g = npt.gbuffer_pixels(sc, (1.0, 1.0, 1.0, 1.0), lights, vi, pi, W, H, xs, ys)
            for k, v in g.items():
                out[name + "/gbuffer_" + k] = v.reshape(len(c["rows"]), W, -1)
            gb_rays = sc.rays_closest
            sc.rays_closest = 0
            acc = None
            for f in range(c["frames"]):
                pc = make_push_constants(samples=1, depth=c["depth"], frame=f, lights_count=lights)
                pc.useShadows, pc.useAO, pc.useGI = 1, 1, 1
                acc = npt.hybrid_pixels(sc, pc, vi, W, H, c["seed0"] + f, xs, ys, g, accum_old=acc)
            out[name + "/accum"] = acc.reshape(len(c["rows"]), W, 4)
            out[name + "/gbuffer_rays"] = np.array([gb_rays], np.int64)
        out[name + "/rays"] = np.array([sc.rays_closest, sc.rays_shadow], np.int64)
        out[name + "/rows"] = np.asarray(c["rows"], np.int64)
        print(f"{name}: {len(xs)} pixels, rays closest {sc.rays_closest} shadow {sc.rays_shadow}, {time.time() - t0:.1f} s", flush=True)
    path = os.path.join(ROOT, "tests", "golden", "np_pathtrace_fixtures.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
