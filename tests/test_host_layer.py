"""Host layer above the C ABI (vk-raytracing-engine_amd/host): glTF ingest, camera, config.json.
The C++ loader is checked against the numpy restatement of the ingest (oracle/gltf_flatten.py) and
against the original arrays of generated scenes after a glTF round trip."""
import io
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import vkrt_amd  # noqa: E402,F401
from vkrt_amd import host_py  # noqa: E402

REF_CORNELL = "/root/reference/media/scenes/cornell.gltf"


def _same_geometry(a, b, exact_tangents=False):
    assert a.positions.shape == b.positions.shape and np.array_equal(a.positions, b.positions)
    assert np.array_equal(a.indices, b.indices)
    assert np.allclose(a.normals, b.normals, atol=1e-6)
    assert np.array_equal(a.texcoords0, b.texcoords0)
    if exact_tangents:
        assert np.array_equal(a.tangents, b.tangents)
    else:
        assert np.allclose(a.tangents, b.tangents, atol=2e-5)
    for k in ("firstIndex", "indexCount", "vertexOffset", "vertexCount", "materialIndex"):
        assert np.array_equal(a.prim_meshes[k], b.prim_meshes[k]), k
    assert np.array_equal(a.nodes["primMesh"], b.nodes["primMesh"])
    assert np.allclose(a.nodes["worldMatrix"], b.nodes["worldMatrix"], atol=1e-6)
    for k in a.materials.dtype.names:
        assert np.allclose(a.materials[k], b.materials[k], atol=1e-7), k
    for k in a.lights.dtype.names:
        assert np.allclose(a.lights[k], b.lights[k], atol=1e-6), k


@pytest.mark.skipif(not os.path.exists(REF_CORNELL), reason="reference tree not mounted")
def test_cpp_loader_matches_numpy_restatement_on_cornell(cornell_flat):
    import gltf_flatten

    a = host_py.load_gltf(REF_CORNELL)
    b = gltf_flatten.load_gltf(REF_CORNELL)
    _same_geometry(a, b)
    _same_geometry(a, cornell_flat)  # and the committed fixture
    assert a.positions.shape[0] == 16808 and a.indices.shape[0] == 50160 and len(a.prim_meshes) == 9 and len(a.nodes) == 10
    assert a.instanced_triangle_count == 16732 and len(a.lights) == 1 and len(a.textures) == 0
    assert np.allclose(a.lights["position"][0], [0, 4.5, 0]) and a.lights["intensity"][0] == 100.0
    assert np.all(np.isfinite(a.tangents))


@pytest.fixture(scope="module")
def small_atrium():
    import atrium

    return atrium.build_atrium(2000, seed=5, with_textures=True)[0]


@pytest.mark.parametrize("mode", ["gltf", "glb", "embedded"])
def test_gltf_round_trip_of_generated_scene(tmp_path, small_atrium, mode):
    import gltf_export
    import gltf_flatten

    path = str(tmp_path / ("a.glb" if mode == "glb" else "a.gltf"))
    gltf_export.export_gltf(small_atrium, path, glb=(mode == "glb"), embed=(mode == "embedded"))
    a = host_py.load_gltf(path)
    _same_geometry(a, small_atrium, exact_tangents=True)  # TANGENT present -> taken verbatim
    assert len(a.lights) == 8 and np.allclose(a.lights["position"][0], [1.0, 5.0, -1.33])  # fallback lights
    assert len(a.textures) == len(small_atrium.textures)
    for ta, tb in zip(a.textures, small_atrium.textures):
        assert np.array_equal(ta["rgba8"], tb["rgba8"]) and ta["is_srgb"] == tb["is_srgb"]
    b = gltf_flatten.load_gltf(path)
    _same_geometry(a, b, exact_tangents=True)


def test_random_scenes_round_trip_through_both_loaders(tmp_path):
    """Random scenes of the GPU campaign's generator (tools/fuzz_parity.py: soups, grids, ragged index counts, arbitrary node
    matrices with negative scales, random materials / NPOT textures / light types) exported as .gltf / .glb / embedded and read
    back by the C++ loader and by the numpy restatement of the reference's flattening: same arrays (750 seeds checked by hand,
    40 kept here)."""
    import gltf_export
    import gltf_flatten

    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py")).read()
    ns = {}
    exec("import numpy as np\nfrom vkrt_amd.flat_scene import LIGHT_DTYPE, MAT_DTYPE, NODE_DTYPE, PRIM_DTYPE, FlatScene\n"
         + src[src.index("def random_scene(rng):"): src.index("def run_case(")], ns)  # (the generator only: the tool itself imports torch)
    for seed in range(5000, 5040):
        flat = ns["random_scene"](np.random.default_rng(seed))
        mode = ("gltf", "glb", "embedded")[seed % 3]
        path = str(tmp_path / ("a.glb" if mode == "glb" else "a.gltf"))
        gltf_export.export_gltf(flat, path, glb=(mode == "glb"), embed=(mode == "embedded"), write_lights=True)
        a, b = host_py.load_gltf(path), gltf_flatten.load_gltf(path)
        _same_geometry(a, b, exact_tangents=True)
        assert len(a.textures) == len(b.textures) and all(np.array_equal(x["rgba8"], y["rgba8"]) and x["is_srgb"] == y["is_srgb"]
                                                          for x, y in zip(a.textures, b.textures)), seed
        assert len(a.lights) == len(b.lights) and np.allclose(a.lights["position"], b.lights["position"]), seed
        assert np.array_equal(a.lights["type"], b.lights["type"]), seed
        for k in a.materials.dtype.names:
            assert np.allclose(np.asarray(a.materials[k], np.float64), np.asarray(b.materials[k], np.float64)), (seed, k)


def _write_handmade(tmp_path, with_normals, index_dtype):
    """Two triangles as one mesh used by two nodes under a parent with TRS; no tangents, optional normals/uvs."""
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0.5]], np.float32)
    nrm = np.tile(np.array([[0, 0, 1]], np.float32), (4, 1))
    uv = np.array([[0, 0], [1, 0], [0, 1], [1, 1]], np.float32)
    idx = np.array([0, 1, 2, 2, 1, 3], index_dtype)
    blob = bytearray()
    views, acc = [], []
    def add(arr, ctype, atype):
        while len(blob) % 4:
            blob.append(0)
        views.append({"buffer": 0, "byteOffset": len(blob), "byteLength": arr.nbytes})
        blob.extend(arr.tobytes())
        acc.append({"bufferView": len(views) - 1, "componentType": ctype, "count": len(arr), "type": atype})
        return len(acc) - 1
    attrs = {"POSITION": add(pos, 5126, "VEC3")}
    if with_normals:
        attrs["NORMAL"] = add(nrm, 5126, "VEC3")
        attrs["TEXCOORD_0"] = add(uv, 5126, "VEC2")
    ia = add(idx, {np.uint8: 5121, np.uint16: 5123, np.uint32: 5125}[index_dtype], "SCALAR")
    g = {
        "asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}],
        "nodes": [
            {"translation": [1, 2, 3], "rotation": [0, 0.7071068, 0, 0.7071068], "scale": [2, 2, 2], "children": [1, 2]},
            {"mesh": 0, "translation": [0, 0, -1]},
            {"mesh": 0, "matrix": [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 5, 0, 0, 1]},
        ],
        "meshes": [{"primitives": [{"attributes": attrs, "indices": ia}, {"attributes": attrs, "indices": ia, "material": 0}]}],
        "materials": [{"pbrMetallicRoughness": {"metallicFactor": 0.25}, "emissiveFactor": [1, 2, 3]}],
        "accessors": acc, "bufferViews": views, "buffers": [{"byteLength": len(blob), "uri": "h.bin"}],
    }
    (tmp_path / "h.bin").write_bytes(bytes(blob))
    (tmp_path / "h.gltf").write_text(json.dumps(g))
    return str(tmp_path / "h.gltf")


@pytest.mark.parametrize("with_normals", [True, False])
@pytest.mark.parametrize("index_dtype", [np.uint8, np.uint16, np.uint32])
def test_handmade_gltf_hierarchy_sharing_and_defaults(tmp_path, with_normals, index_dtype):
    import gltf_flatten

    path = _write_handmade(tmp_path, with_normals, index_dtype)
    a = host_py.load_gltf(path)
    b = gltf_flatten.load_gltf(path)
    _same_geometry(a, b)
    # two primitives sharing one attribute set share vertices (cache); 2 nodes x 2 primMeshes
    assert len(a.prim_meshes) == 2 and a.positions.shape[0] == 4 and len(a.nodes) == 4
    assert np.array_equal(a.prim_meshes["vertexOffset"], [0, 0]) and np.array_equal(a.prim_meshes["materialIndex"], [-1, 0])
    assert a.indices.dtype == np.uint32 and a.indices.shape[0] == 12
    # parent TRS: scale 2, rotate +90 deg about y, translate (1,2,3); child translation (0,0,-1) -> world (1-2, 2, 3)
    w = a.nodes["worldMatrix"][0].reshape(4, 4).T
    assert np.allclose(w[:3, 3], [-1, 2, 3], atol=1e-5) and np.allclose(np.linalg.det(w[:3, :3]), 8, atol=1e-4)
    assert a.materials["metallicFactor"][0] == 0.25 and a.materials["roughnessFactor"][0] == 1.0
    assert np.array_equal(a.materials["emissiveFactor"][0], [1, 2, 3]) and a.materials["pbrBaseColorTexture"][0] == -1
    assert len(a.lights) == 8 and a.lights["intensity"][3] == 50.0
    if not with_normals:
        assert np.allclose(np.linalg.norm(a.normals, axis=1), 1, atol=1e-6) and np.all(a.texcoords0 == 0)
    assert np.all(np.isfinite(a.tangents)) and np.allclose(np.linalg.norm(a.tangents[:, :3], axis=1), 1, atol=1e-5)
    assert set(np.unique(a.tangents[:, 3])) <= {-1.0, 1.0}


def test_loader_errors(tmp_path):
    (tmp_path / "bad.gltf").write_text("{ not json")
    with pytest.raises(RuntimeError):
        host_py.load_gltf(str(tmp_path / "bad.gltf"))
    with pytest.raises(RuntimeError):
        host_py.load_gltf(str(tmp_path / "missing.gltf"))


@pytest.mark.parametrize("mode", ["RGBA", "RGB", "L", "LA", "P", "I;16"])
def test_png_decoder(mode):
    from PIL import Image

    rng = np.random.default_rng(1)
    if mode == "I;16":
        arr = rng.integers(0, 65536, (9, 13), dtype=np.uint16)
        im = Image.fromarray(arr)  # uint16 -> mode I;16
        want = np.stack([(arr >> 8).astype(np.uint8)] * 3 + [np.full_like(arr, 255, np.uint8)], -1)
    else:
        base = Image.fromarray(rng.integers(0, 256, (9, 13, 4), dtype=np.uint8), "RGBA")
        im = base.convert(mode) if mode != "P" else base.convert("RGB").quantize(16)
        want = np.array(im.convert("RGBA"), np.uint8)
    buf = io.BytesIO()
    im.save(buf, format="PNG")
    got = host_py.decode_png(buf.getvalue())
    assert np.array_equal(got, want)


def test_camera_matches_numpy_restatement():
    import camera_np

    for kw in (dict(), dict(eye=(-12.5, 4.2, 0.6), center=(6.0, 3.6, -0.4), fov=45.0, width=1920, height=1080)):
        u = host_py.global_uniforms(**kw)
        vp, vi, pi = camera_np.global_uniforms(**kw)
        for name, M in (("viewProj", vp), ("viewInverse", vi), ("projInverse", pi)):
            got = np.array(getattr(u, name).m, np.float32).reshape(4, 4).T
            assert np.allclose(got, M, rtol=2e-5, atol=2e-5), name
    # perspectiveVK flips Y and maps depth to 0..1: m11 < 0, projInverse exists
    u = host_py.global_uniforms()
    proj_inv = np.array(u.projInverse.m).reshape(4, 4).T
    assert proj_inv[1, 1] < 0


def test_config_json_schema():
    text = json.dumps({"scenes": ["media/scenes/Sponza.gltf", "media/scenes/fireplace/fireplace.gltf", "media/scenes/cornell.gltf",
                                  "media/scenes/suntemple/suntemple.gltf"], "scene": 2, "vsync": False, "width": 1280, "height": 720})
    c = host_py.parse_config(text)
    assert c["scene_path"] == "media/scenes/cornell.gltf" and (c["width"], c["height"], c["vsync"]) == (1280, 720, 0)
    assert (c["samples"], c["depth"], c["frames"]) == (1, 3, 1)  # reference defaults hello_vulkan.cpp:911-912
    c = host_py.parse_config(json.dumps({"scenes": ["a.gltf"], "scene": 0, "vsync": True, "width": 64, "height": 32, "samples": 16, "depth": 8, "frames": 4}))
    assert (c["samples"], c["depth"], c["frames"], c["vsync"]) == (16, 8, 4, 1)
    # round 4: "framesPerCall" (frames handed to the library per call) and the tri-state library options: a key config.json does not
    # name stays -1 = "leave what vkrt_scene_create read from the environment", a named one is 0 / 1
    assert (c["framesPerCall"], c["watertight"], c["anyHitDissolve"], c["skipDeadShadowRays"]) == (1, -1, -1, -1)
    c = host_py.parse_config(json.dumps({"scenes": ["a.gltf"], "scene": 0, "vsync": True, "width": 64, "height": 32, "frames": 12, "framesPerCall": 6,
                                         "watertight": False, "skipDeadShadowRays": True}))
    assert (c["framesPerCall"], c["watertight"], c["anyHitDissolve"], c["skipDeadShadowRays"]) == (6, 0, -1, 1)
    with pytest.raises(ValueError):
        host_py.parse_config(json.dumps({"scenes": ["a.gltf"], "scene": 0, "vsync": True, "width": 64, "height": 32, "framesPerCall": 0}))
    with pytest.raises(ValueError):
        host_py.parse_config(json.dumps({"scenes": ["a.gltf"], "scene": 0, "vsync": True, "width": 64}))  # missing key
    with pytest.raises(ValueError):
        host_py.parse_config(json.dumps({"scenes": ["a.gltf"], "scene": 3, "vsync": True, "width": 64, "height": 2}))


@pytest.mark.gpu
def test_end_to_end_cpp_host_render_matches_oracle(tmp_path, small_atrium):
    """glTF file -> C++ loader -> HelloVkrt (same call order as main.cpp) -> image, against the oracle
    fed by the numpy ingest of the same file and the numpy camera."""
    import atrium
    import camera_np
    import gltf_export
    import gltf_flatten
    import oracle_py
    from vkrt_amd import abi
    from vkrt_amd.flat_scene import make_push_constants

    path = str(tmp_path / "atrium.gltf")
    gltf_export.export_gltf(small_atrium, path)
    W, H = 160, 90
    cam = atrium.DEFAULT_CAMERA
    for build in (abi.VKRT_BUILD_SAH_HOST, abi.VKRT_BUILD_LBVH_GPU):
        img = host_py.render_gltf(path, W, H, samples=2, depth=4, frames=3, seed0=10, build=build, **cam)
        flat = gltf_flatten.load_gltf(path)
        orc = oracle_py.OracleScene(flat)
        u = host_py.global_uniforms(width=W, height=H, **cam)  # same matrices the C++ class computes
        ref = np.zeros((H, W, 4), np.float32)
        for f in range(3):
            pc = make_push_constants(samples=2, depth=4, frame=f, lights_count=len(flat.lights))
            orc.render(pc, u, W, H, seed=10 + f, image=ref)
        assert np.mean(np.any(img.view(np.uint32) != ref.view(np.uint32), axis=-1)) < 1e-4
        assert float(np.sqrt(np.mean((img[..., :3] - ref[..., :3]) ** 2))) < 1e-3


@pytest.mark.gpu
def test_bench_runs_a_supplied_gltf(tmp_path, small_atrium):
    """"if the user drops the real Khronos Sponza next to the config, the same harness runs it" (config.json:2-7).
    bench.py --gltf on an exported scene file: one JSON line with the contract's fields, rays counted, roofline present."""
    import subprocess

    import atrium
    import gltf_export

    path = str(tmp_path / "supplied.gltf")
    gltf_export.export_gltf(small_atrium, path)
    cam = atrium.DEFAULT_CAMERA
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gltf", path, "--width", "320", "--height", "180", "--spp", "2", "--depth", "4", "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline", "--no-other-builder", "--eye", *map(str, cam["eye"]), "--center", *map(str, cam["center"]), "--fov", str(cam.get("fov", 60.0))]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["metric"] == "Mrays/s" and d["value"] > 0 and d["n_gpus"] == 1 and "supplied.gltf" in d["config"]["workload"]
    assert d["config"]["triangles"] == small_atrium.instanced_triangle_count and d["config"]["rays_per_step"] > 320 * 180 * 2
    assert d["roofline"]["kernel"] == "k_wf_traverse" and d["roofline"]["pmc_stale"] is True  # the committed PMC pass is of the bench workload, not of this file


@pytest.mark.gpu
def test_cli_renders_config_json(tmp_path, small_atrium):
    import subprocess

    import gltf_export

    gltf_export.export_gltf(small_atrium, str(tmp_path / "scene.gltf"))
    cfg = {"scenes": ["scene.gltf"], "scene": 0, "vsync": False, "width": 96, "height": 54, "samples": 2, "depth": 3, "frames": 2,
           "camera": {"eye": [-12.5, 4.2, 0.6], "center": [6.0, 3.6, -0.4], "up": [0, 1, 0], "fov": 60}, "output": str(tmp_path / "out")}
    (tmp_path / "config.json").write_text(json.dumps(cfg))
    exe = os.path.join(ROOT, "vk-raytracing-engine_amd", "vkrt_render")
    p = subprocess.run([exe, "--config", str(tmp_path / "config.json")], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    assert "Mrays/s" in p.stdout
    pfm = (tmp_path / "out.pfm").read_bytes()
    assert pfm.startswith(b"PF\n96 54\n-1.0\n") and len(pfm) == len(b"PF\n96 54\n-1.0\n") + 96 * 54 * 12
    assert (tmp_path / "out.ppm").read_bytes().startswith(b"P6\n96 54\n255\n")
    # display image = post.frag (pass-through + gamma 1/2.2) of the radiance image, as 8-bit PNG
    from PIL import Image

    import imgdiff

    lin, _ = imgdiff.read_image(str(tmp_path / "out.pfm"))
    png = np.asarray(Image.open(tmp_path / "out.png"), np.float32)
    assert png.shape == (54, 96, 4) and np.all(png[..., 3] == 255)
    want = np.clip(np.clip(lin, 0, None) ** (1 / 2.2), 0, 1) * 255
    assert np.abs(png[..., :3] - want).max() <= 1.0
    assert imgdiff.main([str(tmp_path / "out.pfm"), str(tmp_path / "out.pfm"), "--out", str(tmp_path / "d.png")])["rmse"] == 0.0


@pytest.mark.gpu
def test_cli_hybrid_mode_matches_oracle(tmp_path, small_atrium):
    """config.json "mode": "hybrid" -> rasterizeGltf -> raytraceRasterizedScene -> drawPost (reference main.cpp:510-561)."""
    import subprocess

    import atrium
    import gltf_export
    import gltf_flatten
    import oracle_py
    from vkrt_amd.flat_scene import make_push_constants

    path = str(tmp_path / "scene.gltf")
    gltf_export.export_gltf(small_atrium, path)
    W, H = 96, 54
    cam = atrium.DEFAULT_CAMERA
    cfg = {"scenes": ["scene.gltf"], "scene": 0, "vsync": False, "width": W, "height": H, "depth": 3, "frames": 2, "mode": "hybrid", "useGI": True,
           "seed": 5, "camera": {"eye": list(cam["eye"]), "center": list(cam["center"]), "up": list(cam["up"]), "fov": cam["fov"]},
           "output": str(tmp_path / "hy")}
    (tmp_path / "config.json").write_text(json.dumps(cfg))
    exe = os.path.join(ROOT, "vk-raytracing-engine_amd", "vkrt_render")
    p = subprocess.run([exe, "--config", str(tmp_path / "config.json")], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    head = b"PF\n%d %d\n-1.0\n" % (W, H)
    pfm = (tmp_path / "hy.pfm").read_bytes()
    img = np.frombuffer(pfm[len(head):], np.float32).reshape(H, W, 3)[::-1]  # PFM rows run bottom-up

    flat = gltf_flatten.load_gltf(path)
    orc = oracle_py.OracleScene(flat)
    u = host_py.global_uniforms(width=W, height=H, **cam)
    g = orc.gbuffer(u, W, H, lights_count=len(flat.lights), clear_color=(1, 1, 1, 1))
    acc = np.zeros((H, W, 4), np.float32)
    for f in range(2):
        pc = make_push_constants(samples=1, depth=3, frame=f, lights_count=len(flat.lights))
        pc.useShadows, pc.useAO, pc.useGI = 1, 1, 1
        acc, _ = orc.hybrid(pc, u, W, H, g, seed=5 + f, accum=acc)  # seedPerFrame default
    want = g["color"][..., :3] * acc[..., 3:4] + acc[..., :3]
    bad = np.abs(img - want) > 1e-4 * (1 + np.abs(want))
    assert bad.any(axis=-1).mean() < 2e-3


def _jpeg_bytes(img, **kw):
    import io

    from PIL import Image

    b = io.BytesIO()
    Image.fromarray(img).save(b, "JPEG", **kw)
    return b.getvalue()


def _picture(w, h, seed=1):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(xx / 7.0) * np.cos(yy / 11.0), 128 + 90 * np.cos(xx / 5.0 + yy / 9.0), (xx * 3 + yy * 5) % 256], -1)
    img = img + rng.normal(0, 6, img.shape)
    img[h // 3: h // 2 + 1, w // 4: w // 2 + 1] = [250, 10, 10]  # a saturated edge: chroma upsampling and clamping
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("size", [(64, 48), (37, 21), (1, 1), (8, 8), (17, 33), (200, 120)])
def test_jpeg_decoder_against_pillow(size):
    """host/jpeg_decode.cpp (the stb_image arithmetic the reference's texels come from, restated) against Pillow / libjpeg on
    generated files: same Huffman / progressive decoding, so the two differ only where the standard leaves rounding open
    (IDCT precision, the 3/4 - 1/4 upsampling filter's rounding, the fixed-point colour matrix): at most 3 levels."""
    import io

    from PIL import Image

    w, h = size
    src = _picture(w, h)
    variants = {"444": dict(quality=90, subsampling=0), "420": dict(quality=90, subsampling=2), "422": dict(quality=90, subsampling=1),
                "progressive420": dict(quality=90, subsampling=2, progressive=True), "progressive444": dict(quality=85, subsampling=0, progressive=True),
                "q30": dict(quality=30, subsampling=2), "optimised": dict(quality=90, subsampling=2, optimize=True),
                "restart": dict(quality=90, subsampling=2, restart_marker_blocks=3), "restart_rows_prog": dict(quality=80, subsampling=0, progressive=True, restart_marker_rows=1),
                "q100": dict(quality=100, subsampling=0)}
    for name, kw in variants.items():
        data = _jpeg_bytes(src, **kw)
        want = np.asarray(Image.open(io.BytesIO(data)).convert("RGB")).astype(int)
        got = host_py.decode_image(data)
        assert got.shape == (h, w, 4) and (got[..., 3] == 255).all(), name
        d = np.abs(got[..., :3].astype(int) - want)
        if name == "422" and w > 2:
            # stb_image's 2x horizontal filter weighs the second-to-last output column (3 * in[w-2] + in[w-1]) / 4 where the
            # symmetric filter has (in[w-2] + 3 * in[w-1]) / 4; restated as published, so that column is not Pillow's
            d[:, 2 * ((w + 1) // 2 - 1)] = 0
        assert d.max() <= 3 and d.mean() < 0.5, (name, int(d.max()), float(d.mean()))
    grey = _jpeg_bytes(src[..., 1], quality=92)
    got = host_py.decode_image(grey)
    want = np.asarray(Image.open(io.BytesIO(grey)).convert("L")).astype(int)
    assert np.abs(got[..., 0].astype(int) - want).max() <= 1 and np.array_equal(got[..., 0], got[..., 1]) and np.array_equal(got[..., 0], got[..., 2])


def test_jpeg_decoder_refuses_what_it_does_not_read():
    import io

    from PIL import Image

    cmyk = io.BytesIO()
    Image.fromarray(_picture(16, 16)).convert("CMYK").save(cmyk, "JPEG")
    with pytest.raises(ValueError, match="CMYK"):
        host_py.decode_image(cmyk.getvalue())
    good = _jpeg_bytes(_picture(32, 24), quality=90)
    with pytest.raises(ValueError):
        host_py.decode_image(good[:2] + b"\xff\xc9\x00\x0b\x08\x00\x10\x00\x10\x01\x01\x11\x00" + good[2:])  # SOF9: arithmetic coding
    with pytest.raises(ValueError):
        host_py.decode_image(good[:40])  # truncated before any scan
    big = _jpeg_bytes(_picture(200, 120), quality=90)
    cut = host_py.decode_image(big[: len(big) // 2])  # a truncated scan still yields an image (missing data reads as zeros), as stb_image does
    assert cut.shape == (120, 200, 4)
    with pytest.raises(ValueError):
        host_py.decode_image(b"\x00" * 64)


def test_image_decoders_survive_corrupt_input():
    """Mutated PNG / JPEG files either decode or are refused with a message (the decoders were also run under ASan + UBSan on
    100 k such files): no crash, no allocation driven by a forged header."""
    import io
    import struct

    from PIL import Image

    rng = np.random.default_rng(9)
    seeds = [_jpeg_bytes(_picture(40, 24), quality=80, subsampling=2), _jpeg_bytes(_picture(40, 24), quality=80, subsampling=0, progressive=True)]
    b = io.BytesIO()
    Image.fromarray(_picture(31, 17)).save(b, "PNG")
    seeds.append(b.getvalue())
    decoded = refused = 0
    for data in seeds:
        for _ in range(150):
            m = bytearray(data)
            for _k in range(int(rng.integers(1, 6))):
                pos = int(rng.integers(0, len(m)))
                kind = int(rng.integers(0, 4))
                if kind == 0:
                    m[pos] = int(rng.integers(0, 256))
                elif kind == 1:
                    m[pos] ^= 1 << int(rng.integers(0, 8))
                elif kind == 2:
                    del m[pos + 1:]
                else:
                    m[pos] = 0xFF
            try:
                img = host_py.decode_image(bytes(m))
                assert img.ndim == 3 and img.shape[2] == 4
                decoded += 1
            except ValueError:
                refused += 1
    assert decoded > 20 and refused > 20
    # a PNG header promising 60000 x 60000 texels over a few bytes of data
    png = bytearray(seeds[2])
    png[16:24] = struct.pack(">II", 60000, 60000)
    with pytest.raises(ValueError):
        host_py.decode_image(bytes(png))


def test_jpeg_textures_load_natively_and_sidecar_wins(tmp_path):
    """SURVEY 8f row 2 / VERDICT r01 missing 6: JPEG images (external file, GLB-embedded bufferView) decode in the C++ loader;
    a .rgba8 sidecar from tools/decode_textures.py still takes precedence; an image the decoder refuses becomes the reference's
    1x1 white dummy (hello_vulkan.cpp:487-491)."""
    import io

    from PIL import Image

    import decode_textures

    rng = np.random.default_rng(5)
    img = np.kron(rng.integers(0, 256, (4, 6, 3), dtype=np.uint8), np.ones((8, 8, 1), np.uint8))  # blocky: JPEG-friendly
    Image.fromarray(img, "RGB").save(tmp_path / "albedo.jpg", quality=95)
    Image.fromarray(img, "RGB").convert("CMYK").save(tmp_path / "print.jpg")
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    (tmp_path / "tri.bin").write_bytes(pos.tobytes() + np.array([0, 1, 2], np.uint32).tobytes())
    doc = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}], "nodes": [{"mesh": 0}],
           "meshes": [{"primitives": [{"attributes": {"POSITION": 0}, "indices": 1, "material": 0}]}],
           "materials": [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}, "metallicRoughnessTexture": {"index": 1}}}],
           "textures": [{"source": 0}, {"source": 1}], "images": [{"uri": "albedo.jpg"}, {"uri": "print.jpg"}],
           "buffers": [{"uri": "tri.bin", "byteLength": 48}],
           "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 36}, {"buffer": 0, "byteOffset": 36, "byteLength": 12}],
           "accessors": [{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3", "min": [0, 0, 0], "max": [1, 1, 0]},
                         {"bufferView": 1, "componentType": 5125, "count": 3, "type": "SCALAR"}]}
    path = tmp_path / "tri.gltf"
    path.write_text(json.dumps(doc))
    want = np.array(Image.open(tmp_path / "albedo.jpg").convert("RGBA"), np.uint8)
    flat = host_py.load_gltf(str(path))
    assert len(flat.textures) == 2 and flat.textures[0]["is_srgb"] and not flat.textures[1]["is_srgb"]  # hello_vulkan.cpp:417-443
    native = flat.textures[0]["rgba8"]
    assert native.shape == want.shape and np.abs(native.astype(int) - want.astype(int)).max() <= 3
    assert np.abs(native[..., :3].astype(int) - img.astype(int)).mean() < 16  # it really is the picture
    assert flat.textures[1]["rgba8"].shape == (1, 1, 4) and np.all(flat.textures[1]["rgba8"] == 255)  # CMYK: refused -> white dummy
    assert decode_textures.main([str(path)]) == 0 and not (tmp_path / "albedo.jpg.rgba8").exists()  # nothing the loader cannot read... but CMYK is by extension a .jpg
    assert decode_textures.main([str(path), "--all"]) == 0
    flat = host_py.load_gltf(str(path))
    assert np.array_equal(flat.textures[0]["rgba8"], want)  # the sidecar (Pillow's decode) wins
    assert flat.textures[1]["rgba8"].shape == (32, 48, 4)   # and rescues the CMYK file

    # the same JPEG embedded in a .glb (bufferView image)
    import struct

    jpg = (tmp_path / "albedo.jpg").read_bytes()
    geo = pos.tobytes() + np.array([0, 1, 2], np.uint32).tobytes()
    blob = geo + jpg + b"\x00" * (-len(jpg) % 4)
    doc["images"] = [{"bufferView": 2, "mimeType": "image/jpeg"}]
    doc["textures"] = [{"source": 0}]
    doc["materials"] = [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}}}]
    doc["buffers"] = [{"byteLength": len(blob)}]
    doc["bufferViews"].append({"buffer": 0, "byteOffset": 48, "byteLength": len(jpg)})
    js = json.dumps(doc).encode()
    js += b" " * (-len(js) % 4)
    glb = struct.pack("<III", 0x46546C67, 2, 12 + 8 + len(js) + 8 + len(blob)) + struct.pack("<II", len(js), 0x4E4F534A) + js + \
        struct.pack("<II", len(blob), 0x004E4942) + blob
    (tmp_path / "tri.glb").write_bytes(glb)
    flat = host_py.load_gltf(str(tmp_path / "tri.glb"))
    assert np.array_equal(flat.textures[0]["rgba8"], native)


def test_png_writer_round_trip(tmp_path):
    """the display image (post.frag output) as an 8-bit PNG; read back with PIL and with the native decoder."""
    from PIL import Image

    rng = np.random.default_rng(9)
    disp = rng.uniform(-0.1, 1.1, (37, 53, 4)).astype(np.float32)
    disp[0, 0] = [0.0, 1.0, 0.5, 1.0]
    path = str(tmp_path / "d.png")
    host_py.write_png(path, disp)
    want = (np.clip(disp, 0.0, 1.0) * np.float32(255.0) + np.float32(0.5)).astype(np.uint8)
    assert np.array_equal(np.array(Image.open(path)), want)
    assert np.array_equal(host_py.decode_png(open(path, "rb").read()), want)


def test_imgdiff_tool(tmp_path):
    import imgdiff

    rng = np.random.default_rng(3)
    a = rng.uniform(0, 1, (20, 30, 3)).astype(np.float32)
    b = a.copy()
    b[5, 7] += 0.5
    np.save(tmp_path / "a.npy", a)
    np.save(tmp_path / "b.npy", b)
    st = imgdiff.main([str(tmp_path / "a.npy"), str(tmp_path / "b.npy"), "--out", str(tmp_path / "d.png")])
    assert abs(st["max_abs"] - 0.5) < 1e-6 and abs(st["pixels_differing"] - 1 / 600) < 1e-9
    from PIL import Image

    assert Image.open(tmp_path / "d.png").size == (90, 20)


def test_rank_launcher_reaps_failed_ranks_and_cleans_up(tmp_path):
    """`vkrt_render --ranks N` forks one process per GPU before touching HIP.  When a rank fails (here: every rank, on a missing
    config file, before any GPU work) the launcher must come back with that rank's code instead of waiting for ever, and leave
    nothing behind in /tmp (the RCCL id file lives in a private mkdtemp directory)."""
    import glob
    import subprocess

    exe = os.path.join(ROOT, "vk-raytracing-engine_amd", "vkrt_render")
    if not os.path.exists(exe):
        pytest.skip("vkrt_render not built")
    before = set(glob.glob("/tmp/vkrt_rccl_*"))
    p = subprocess.run([exe, "--ranks", "3", "--config", str(tmp_path / "missing.json")], capture_output=True, text=True, timeout=60)
    assert p.returncode == 1, (p.returncode, p.stderr[-500:])
    assert "stopping the other" in p.stderr
    assert set(glob.glob("/tmp/vkrt_rccl_*")) == before
    src = open(os.path.join(ROOT, "vk-raytracing-engine_amd", "host", "main.cpp")).read()
    assert "waitpid(-1" in src and "mkdtemp" in src and "SIGKILL" in src
    gsrc = open(os.path.join(ROOT, "vk-raytracing-engine_amd", "host", "strip_gather.cpp")).read()
    assert "O_EXCL | O_NOFOLLOW" in gsrc


@pytest.mark.gpu
def test_cpp_host_renders_hybrid_frames_in_strips(tmp_path, small_atrium):
    """HelloVkrt::setShard with the hybrid sequence (rasterizeGltf -> raytraceRasterizedScene -> drawPost): the display strips of the
    three ranks of a 3-rank job, stacked by the strip deal, are the whole-frame display image, bit for bit (NaN texels included)."""
    import atrium
    import gltf_export
    from vkrt_amd.sharding import shard_row_indices

    path = str(tmp_path / "scene.gltf")
    gltf_export.export_gltf(small_atrium, path)
    W, H = 128, 70
    cam = atrium.DEFAULT_CAMERA
    whole = host_py.render_gltf_hybrid(path, W, H, depth=4, frames=2, seed0=3, **cam)
    assert whole.shape == (H, W, 4) and np.isfinite(whole[..., :3]).mean() > 0.9
    out = np.zeros_like(whole)
    for rank in range(3):
        part = host_py.render_gltf_hybrid(path, W, H, depth=4, frames=2, seed0=3, rank=rank, world=3, **cam)
        rows = shard_row_indices(H, 3, rank)
        assert part.shape[0] == len(rows)
        out[rows] = part
    assert np.array_equal(out.view(np.uint32), whole.view(np.uint32))


@pytest.mark.gpu
def test_vkrt_render_hybrid_through_the_rank_launcher(tmp_path, small_atrium):
    """`vkrt_render --ranks 1` with "mode": "hybrid": the multi-process path (fork, RCCL id in a private directory, all-gather of the
    display strips, un-interleave) gives the image of the plain single-process run."""
    import subprocess

    import gltf_export
    import imgdiff

    gltf_export.export_gltf(small_atrium, str(tmp_path / "scene.gltf"))
    for name in ("a", "b"):
        cfg = {"scenes": ["scene.gltf"], "scene": 0, "vsync": False, "width": 128, "height": 72, "depth": 3, "frames": 2, "mode": "hybrid", "useGI": True,
               "camera": {"eye": [-12.5, 4.2, 0.6], "center": [6.0, 3.6, -0.4], "up": [0, 1, 0], "fov": 60}, "output": str(tmp_path / name)}
        (tmp_path / f"{name}.json").write_text(json.dumps(cfg))
    exe = os.path.join(ROOT, "vk-raytracing-engine_amd", "vkrt_render")
    p = subprocess.run([exe, "--config", str(tmp_path / "a.json")], capture_output=True, text=True, timeout=180)
    assert p.returncode == 0, p.stderr
    q = subprocess.run([exe, "--config", str(tmp_path / "b.json"), "--ranks", "1"], capture_output=True, text=True, timeout=180)
    assert q.returncode == 0, q.stderr + q.stdout
    a, _ = imgdiff.read_image(str(tmp_path / "a.pfm"))
    b, _ = imgdiff.read_image(str(tmp_path / "b.pfm"))
    assert np.array_equal(a, b, equal_nan=True)


@pytest.mark.gpu
def test_cli_library_options_from_config_json(tmp_path, small_atrium):
    """config.json keys "watertight" / "anyHitDissolve" / "skipDeadShadowRays" reach the library through HelloVkrt (applied before the
    acceleration-structure build): the CLI's image equals the oracle's under the same switches, and the dissolve stage visibly changes
    a scene that has translucent materials."""
    import copy
    import subprocess

    import atrium
    import gltf_export
    import gltf_flatten
    import imgdiff
    import oracle_py
    from vkrt_amd.flat_scene import make_push_constants

    flat = copy.deepcopy(small_atrium)
    flat.materials["pbrBaseColorFactor"][0:24:3, 3] = 0.35  # every third material translucent
    flat.materials["pbrBaseColorFactor"][1, 3] = 0.0        # one invisible
    path = str(tmp_path / "scene.gltf")
    gltf_export.export_gltf(flat, path)
    W, H = 128, 72
    cam = atrium.DEFAULT_CAMERA
    base = {"scenes": ["scene.gltf"], "scene": 0, "vsync": False, "width": W, "height": H, "samples": 2, "depth": 4, "frames": 2, "seed": 9,
            "camera": {"eye": list(cam["eye"]), "center": list(cam["center"]), "up": list(cam["up"]), "fov": cam["fov"]}}
    exe = os.path.join(ROOT, "vk-raytracing-engine_amd", "vkrt_render")
    images = {}
    for name, extra in (("plain", {}), ("options", {"watertight": True, "anyHitDissolve": True, "skipDeadShadowRays": True})):
        (tmp_path / f"{name}.json").write_text(json.dumps({**base, **extra, "output": str(tmp_path / name)}))
        p = subprocess.run([exe, "--config", str(tmp_path / f"{name}.json")], capture_output=True, text=True, timeout=180)
        assert p.returncode == 0, p.stderr
        images[name], _ = imgdiff.read_image(str(tmp_path / f"{name}.pfm"))
    assert np.mean(np.abs(images["plain"] - images["options"]).max(-1) > 1e-3) > 0.02  # translucent materials let light and rays through
    f2 = gltf_flatten.load_gltf(path)
    assert abs(float(f2.materials["pbrBaseColorFactor"][0, 3]) - 0.35) < 1e-6  # alpha survives the export / ingest
    orc = oracle_py.OracleScene(f2)
    orc.set_watertight(True); orc.set_dissolve(True)
    u = host_py.global_uniforms(width=W, height=H, **cam)
    ref = np.zeros((H, W, 4), np.float32)
    for f in range(2):
        pc = make_push_constants(samples=2, depth=4, frame=f, lights_count=len(f2.lights))
        orc.render(pc, u, W, H, seed=9 + f, image=ref)
    got = images["options"]
    assert np.mean(np.any(got[..., :3].view(np.uint32) != ref[..., :3].view(np.uint32), axis=-1)) < 1e-4


@pytest.mark.gpu
def test_cli_frames_per_call_and_environment_precedence(tmp_path, small_atrium):
    """`"framesPerCall": n` hands n iterations of the frame loop (main.cpp:503-508) to the library as ONE vkrt_pathtrace_frames call
    through HelloVkrt::pathtraceFrames: five frames rendered 3 + 2 equal five single-frame iterations bit for bit, with and without a
    per-frame seed.  And the precedence of the library options (round 3 advisor): a key that config.json does not name leaves the
    value vkrt_scene_create read from the environment (VKRT_WATERTIGHT=1 changes the image), a key it names wins over the environment."""
    import subprocess

    import atrium
    import gltf_export
    import imgdiff

    gltf_export.export_gltf(small_atrium, str(tmp_path / "scene.gltf"))
    W, H = 160, 96
    cam = atrium.DEFAULT_CAMERA
    base = {"scenes": ["scene.gltf"], "scene": 0, "vsync": False, "width": W, "height": H, "samples": 2, "depth": 4, "frames": 5, "seed": 3,
            "camera": {"eye": list(cam["eye"]), "center": list(cam["center"]), "up": list(cam["up"]), "fov": cam["fov"]}}
    exe = os.path.join(ROOT, "vk-raytracing-engine_amd", "vkrt_render")

    def render(name, extra, env=None):
        (tmp_path / f"{name}.json").write_text(json.dumps({**base, **extra, "output": str(tmp_path / name)}))
        p = subprocess.run([exe, "--config", str(tmp_path / f"{name}.json")], capture_output=True, text=True, timeout=180, env={**os.environ, **(env or {})})
        assert p.returncode == 0, p.stderr
        return imgdiff.read_image(str(tmp_path / f"{name}.pfm"))[0]

    for per_frame in (True, False):
        one = render(f"one{per_frame}", {"seedPerFrame": per_frame})
        batched = render(f"batched{per_frame}", {"seedPerFrame": per_frame, "framesPerCall": 3})
        assert np.array_equal(one.view(np.uint32), batched.view(np.uint32)), per_frame
    plain = render("plain", {})
    env_wt = render("env_wt", {}, env={"VKRT_WATERTIGHT": "1"})
    assert not np.array_equal(plain.view(np.uint32), env_wt.view(np.uint32))          # the environment reaches the C++ host now
    cfg_off = render("cfg_off", {"watertight": False}, env={"VKRT_WATERTIGHT": "1"})
    assert np.array_equal(plain.view(np.uint32), cfg_off.view(np.uint32))             # an explicit key wins over it
    cfg_on = render("cfg_on", {"watertight": True})
    assert np.array_equal(env_wt.view(np.uint32), cfg_on.view(np.uint32))
