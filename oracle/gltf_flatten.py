"""numpy restatement of the reference's scene ingest (TEST INFRASTRUCTURE, see oracle.cpp).

Restates what `HelloVulkan::loadGltfScene` (reference hello_vulkan.cpp:327-394) obtains from
nvpro_core's `nvh::GltfScene::importMaterials / importDrawableNodes` (call site :344-346) and
what `loadGltfMaterials` (:207-224) / `loadGltfLights` (:226-325) derive from it.  nvpro_core
is an unpinned third-party dependency that is NOT in the reference tree, so its behaviour is
restated from its published algorithm as recorded in SURVEY.md Appendix D (flatten the default
scene's node hierarchy; one primMesh per mesh primitive in mesh order; shared attribute sets
are cached; u8/u16 indices widened; missing NORMAL -> per-face normals; missing TEXCOORD_0 ->
0; missing TANGENT -> per-vertex UV-derivative tangents, Gram-Schmidt against the normal,
w = handedness).  PARITY UNPINNED at this boundary: no reference test pins these arrays.

Used to (a) produce tests/golden/cornell_flat.npz from the reference's shipped asset and
(b) cross-check the product's C++ loader (vk-raytracing-engine_amd/host/gltf_loader.cpp).
"""
import base64
import json
import os
import struct
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
import vkrt_amd  # noqa: E402
from vkrt_amd.flat_scene import FlatScene, LIGHT_DTYPE, MAT_DTYPE, NODE_DTYPE, PRIM_DTYPE  # noqa: E402  (containers only)


def reference_fallback_lights():
    """The oracle's OWN transcription of the lights loadGltfLights appends when a file has none (hello_vulkan.cpp:255-321:
    one light near the origin of the nave and seven along +z; the commented-out entries at :249-254 and :261-284 are not part
    of it).  Deliberately not shared with the product's table (vkrt_amd.flat_scene.fallback_lights, host/gltf_loader.cpp): a
    test compares the three copies."""
    rows = [((1.0, 5.0, -1.33), (1.0, 1.0, 1.0)),        # :255-260
            ((0.0, 3.0, 67.0), (1.0, 0.01, 0.1)),        # :285-290
            ((-1.3, 7.62, 59.0), (1.0, 1.0, 1.0)),       # :291-296
            ((2.4, 2.05, 40.6), (1.0, 1.0, 1.0)),        # :297-302
            ((-0.33, 6.85, 30.0), (1.0, 1.0, 1.0)),      # :303-308
            ((-6.2, 9.6, 20.18), (1.0, 1.0, 1.0)),       # :309-314
            ((-0.23, 6.93, 12.21), (1.0, 1.0, 0.0)),     # :315-320 (position), colour (1, 1, 0)
            ((0.24, 3.03, 49.94), (0.0, 0.0, 1.0))]      # :321-326
    L = np.zeros(len(rows), LIGHT_DTYPE)
    for i, (pos, col) in enumerate(rows):
        L[i]["position"] = pos
        L[i]["color"] = col
        L[i]["intensity"] = 50.0  # every fallback light: intensity 50, type 0 (point)
        L[i]["type"] = 0
    return L


_COMP = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


def _load_json_and_buffers(path):
    with open(path, "rb") as f:
        head = f.read(12)
        if head[:4] == b"glTF":  # .glb container (hello_vulkan.cpp:338-342)
            f.seek(12)
            g, bin_chunk = None, None
            while True:
                h = f.read(8)
                if len(h) < 8:
                    break
                clen, ctype = struct.unpack("<II", h)
                data = f.read(clen)
                if ctype == 0x4E4F534A:
                    g = json.loads(data.decode("utf-8"))
                elif ctype == 0x004E4942:
                    bin_chunk = data
        else:
            f.seek(0)
            g, bin_chunk = json.loads(f.read().decode("utf-8")), None
    base = os.path.dirname(os.path.abspath(path))
    bufs = []
    for b in g.get("buffers", []):
        uri = b.get("uri")
        if uri is None:
            bufs.append(bin_chunk)
        elif uri.startswith("data:"):
            bufs.append(base64.b64decode(uri.split(",", 1)[1]))
        else:
            with open(os.path.join(base, uri), "rb") as f:
                bufs.append(f.read())
    return g, bufs, base


def _accessor(g, bufs, idx):
    a = g["accessors"][idx]
    ct, n = _COMP[a["componentType"]], _NCOMP[a["type"]]
    count = a["count"]
    if "bufferView" not in a:
        return np.zeros((count, n), ct)
    bv = g["bufferViews"][a["bufferView"]]
    off = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
    item = np.dtype(ct).itemsize * n
    stride = bv.get("byteStride", 0) or item
    raw = np.frombuffer(bufs[bv["buffer"]], np.uint8, count=(count - 1) * stride + item, offset=off)
    if stride == item:
        out = raw.view(ct).reshape(count, n)
    else:
        rows = np.lib.stride_tricks.as_strided(raw, (count, item), (stride, 1))
        out = np.ascontiguousarray(rows).view(ct).reshape(count, n)
    if a.get("normalized") and ct != np.float32:
        info = np.iinfo(ct)
        out = np.maximum(out.astype(np.float32) / info.max, -1.0)
    return out


def _quat_to_mat(q):
    x, y, z, w = [np.float32(v) for v in q]
    M = np.eye(4, dtype=np.float32)
    M[0, 0] = 1 - 2 * (y * y + z * z); M[0, 1] = 2 * (x * y - z * w); M[0, 2] = 2 * (x * z + y * w)
    M[1, 0] = 2 * (x * y + z * w); M[1, 1] = 1 - 2 * (x * x + z * z); M[1, 2] = 2 * (y * z - x * w)
    M[2, 0] = 2 * (x * z - y * w); M[2, 1] = 2 * (y * z + x * w); M[2, 2] = 1 - 2 * (x * x + y * y)
    return M


def _local_matrix(node):
    T = np.eye(4, dtype=np.float32)
    R = np.eye(4, dtype=np.float32)
    S = np.eye(4, dtype=np.float32)
    M = np.eye(4, dtype=np.float32)
    if "translation" in node:
        T[:3, 3] = np.array(node["translation"], np.float32)
    if "rotation" in node:
        R = _quat_to_mat(node["rotation"])
    if "scale" in node:
        S[0, 0], S[1, 1], S[2, 2] = [np.float32(v) for v in node["scale"]]
    if "matrix" in node:
        M = np.array(node["matrix"], np.float32).reshape(4, 4).T  # glTF stores column-major
    return (T @ R @ S @ M).astype(np.float32)


def _gen_tangents(pos, nrm, uv, idx):
    """Per-vertex tangents from UV derivatives (Lengyel), SURVEY Appendix D."""
    V = pos.shape[0]
    tri = idx.reshape(-1, 3)
    p0, p1, p2 = pos[tri[:, 0]], pos[tri[:, 1]], pos[tri[:, 2]]
    w0, w1, w2 = uv[tri[:, 0]], uv[tri[:, 1]], uv[tri[:, 2]]
    e1, e2 = p1 - p0, p2 - p0
    d1, d2 = w1 - w0, w2 - w0
    a = d1[:, 0] * d2[:, 1] - d2[:, 0] * d1[:, 1]
    r = np.where(np.abs(a) > 0, np.float32(1.0) / np.where(a == 0, np.float32(1), a), np.float32(1.0)).astype(np.float32)
    t = (e1 * d2[:, 1:2] - e2 * d1[:, 1:2]) * r[:, None]
    b = (e2 * d1[:, 0:1] - e1 * d2[:, 0:1]) * r[:, None]
    tan = np.zeros((V, 3), np.float32)
    bit = np.zeros((V, 3), np.float32)
    for k in range(3):
        np.add.at(tan, tri[:, k], t)
        np.add.at(bit, tri[:, k], b)
    ndt = np.sum(nrm * tan, axis=1, keepdims=True)
    ot = tan - ndt * nrm
    ln = np.sqrt(np.sum(ot * ot, axis=1, keepdims=True))
    bad = ~(ln[:, 0] > 0) | ~np.isfinite(ln[:, 0])
    with np.errstate(invalid="ignore", divide="ignore"):
        ot = ot / ln
    if bad.any():  # fallback axis as in random.glsl:47-54
        n = nrm[bad]
        use_x = np.abs(n[:, 0]) > np.abs(n[:, 1])
        with np.errstate(invalid="ignore", divide="ignore"):
            fx = np.stack([n[:, 2], np.zeros_like(n[:, 0]), -n[:, 0]], 1) / np.sqrt(n[:, 0] ** 2 + n[:, 2] ** 2)[:, None]
            fy = np.stack([np.zeros_like(n[:, 0]), -n[:, 2], n[:, 1]], 1) / np.sqrt(n[:, 1] ** 2 + n[:, 2] ** 2)[:, None]
        fb = np.where(use_x[:, None], fx, fy)
        fb = np.where(np.isfinite(fb), fb, np.float32(0))
        fb[~np.any(fb != 0, axis=1)] = np.array([1, 0, 0], np.float32)
        ot[bad] = fb
    hand = np.where(np.sum(np.cross(nrm, tan) * bit, axis=1) < 0, np.float32(1.0), np.float32(-1.0))
    return np.concatenate([ot.astype(np.float32), hand[:, None].astype(np.float32)], axis=1)


def _texture_index(obj, key):
    t = obj.get(key)
    return int(t["index"]) if isinstance(t, dict) and "index" in t else -1


def load_gltf(path, image_decoder=None):
    """glTF/.glb -> FlatScene.  image_decoder(bytes_or_path) -> (H,W,4) uint8 (PIL by default)."""
    g, bufs, base = _load_json_and_buffers(path)

    # ---- materials (importMaterials + loadGltfMaterials, hello_vulkan.cpp:207-224) ------
    gm = g.get("materials", [])
    mats = np.zeros(max(1, len(gm)), MAT_DTYPE)
    mats["pbrBaseColorFactor"] = 1.0
    mats["metallicFactor"] = 1.0
    mats["roughnessFactor"] = 1.0
    for k in ("pbrBaseColorTexture", "metallicRoughnessTexture", "normalTexture", "emissiveTexture"):
        mats[k] = -1
    for i, m in enumerate(gm):
        pbr = m.get("pbrMetallicRoughness", {})
        mats[i]["pbrBaseColorFactor"] = np.array(pbr.get("baseColorFactor", [1, 1, 1, 1]), np.float32)
        mats[i]["pbrBaseColorTexture"] = _texture_index(pbr, "baseColorTexture")
        mats[i]["metallicFactor"] = np.float32(pbr.get("metallicFactor", 1.0))
        mats[i]["roughnessFactor"] = np.float32(pbr.get("roughnessFactor", 1.0))
        mats[i]["metallicRoughnessTexture"] = _texture_index(pbr, "metallicRoughnessTexture")
        mats[i]["normalTexture"] = _texture_index(m, "normalTexture")
        mats[i]["emissiveFactor"] = np.array(m.get("emissiveFactor", [0, 0, 0]), np.float32)
        mats[i]["emissiveTexture"] = _texture_index(m, "emissiveTexture")

    # ---- primitive meshes (importDrawableNodes: processMesh in mesh order) ----------------
    P, N, T, UV, IDX, prims = [], [], [], [], [], []
    mesh_to_prims, cache = [], {}
    voff, ioff = 0, 0
    for mesh in g.get("meshes", []):
        mine = []
        for prim in mesh["primitives"]:
            if prim.get("mode", 4) != 4:
                continue
            attr = prim["attributes"]
            pos = _accessor(g, bufs, attr["POSITION"]).astype(np.float32)
            if "indices" in prim:
                idx = _accessor(g, bufs, prim["indices"]).reshape(-1).astype(np.uint32)
            else:
                idx = np.arange(pos.shape[0], dtype=np.uint32)
            idx = idx[: (idx.shape[0] // 3) * 3]
            key = tuple(sorted(attr.items()))
            if key in cache:
                vo, vc = cache[key]
            else:
                vo, vc = voff, pos.shape[0]
                cache[key] = (vo, vc)
                if "NORMAL" in attr:
                    nrm = _accessor(g, bufs, attr["NORMAL"]).astype(np.float32)
                else:
                    nrm = np.zeros_like(pos)
                    tri = idx.reshape(-1, 3)
                    fn = np.cross(pos[tri[:, 1]] - pos[tri[:, 0]], pos[tri[:, 2]] - pos[tri[:, 0]])
                    with np.errstate(invalid="ignore", divide="ignore"):
                        fn = fn / np.sqrt(np.sum(fn * fn, axis=1, keepdims=True))
                    # sequential semantics: a later triangle overwrites the normals of shared vertices
                    nrm[tri.reshape(-1)] = np.repeat(fn, 3, axis=0)
                uv = _accessor(g, bufs, attr["TEXCOORD_0"]).astype(np.float32) if "TEXCOORD_0" in attr else np.zeros((vc, 2), np.float32)
                if "TANGENT" in attr:
                    tan = _accessor(g, bufs, attr["TANGENT"]).astype(np.float32)
                else:
                    tan = _gen_tangents(pos, nrm, uv, idx)
                P.append(pos); N.append(nrm); T.append(tan); UV.append(uv)
                voff += vc
            IDX.append(idx)
            prims.append((ioff, idx.shape[0], vo, vc, int(prim.get("material", -1))))
            ioff += idx.shape[0]
            mine.append(len(prims) - 1)
        mesh_to_prims.append(mine)

    # ---- node hierarchy (processNode) -------------------------------------------------------
    nodes_out, lights_out = [], []
    gl_lights = g.get("extensions", {}).get("KHR_lights_punctual", {}).get("lights", [])

    def walk(ni, parent):
        node = g["nodes"][ni]
        world = (parent @ _local_matrix(node)).astype(np.float32)
        if "mesh" in node:
            for pmi in mesh_to_prims[node["mesh"]]:
                nodes_out.append((world.T.reshape(-1).copy(), pmi))  # column-major storage
        lext = node.get("extensions", {}).get("KHR_lights_punctual")
        if lext is not None:
            lights_out.append((world, gl_lights[lext["light"]]))
        for c in node.get("children", []):
            walk(c, world)

    scene = g["scenes"][g.get("scene", 0)]
    for ni in scene["nodes"]:
        walk(ni, np.eye(4, dtype=np.float32))

    # ---- lights (loadGltfLights, hello_vulkan.cpp:226-325) -----------------------------------
    type_map = {"point": 0, "directional": 1, "spot": 2}
    if lights_out:
        L = np.zeros(len(lights_out), LIGHT_DTYPE)
        for i, (world, l) in enumerate(lights_out):
            L[i]["position"] = world[:3, 3]
            L[i]["color"] = np.array(l.get("color", [1, 1, 1]), np.float32)
            L[i]["intensity"] = np.float32(l.get("intensity", 1.0))
            L[i]["type"] = type_map.get(l.get("type", "point"), 0)
    else:
        L = reference_fallback_lights()

    # ---- textures (createTextureImages, hello_vulkan.cpp:417-513) ------------------------------
    textures = []
    gimages, gtex = g.get("images", []), g.get("textures", [])
    if gimages and gtex:
        if image_decoder is None:
            from PIL import Image
            import io

            def image_decoder(src):
                im = Image.open(io.BytesIO(src) if isinstance(src, (bytes, bytearray)) else src)
                return np.array(im.convert("RGBA"), np.uint8)
        decoded = {}
        srgb_tex = set()
        for m in gm:
            srgb_tex.add(_texture_index(m.get("pbrMetallicRoughness", {}), "baseColorTexture"))
            srgb_tex.add(_texture_index(m, "emissiveTexture"))
        for ti, t in enumerate(gtex):
            src = t.get("source", 0)
            # getImageFormat: image i is sRGB iff the FIRST texture whose source is i is used as a
            # base-colour or emissive texture by some material (hello_vulkan.cpp:417-443)
            first_tex = next(j for j, tj in enumerate(gtex) if tj.get("source", 0) == src)
            if src not in decoded:
                im = gimages[src]
                if "uri" in im and not im["uri"].startswith("data:"):
                    decoded[src] = image_decoder(os.path.join(base, im["uri"]))
                elif "uri" in im:
                    decoded[src] = image_decoder(base64.b64decode(im["uri"].split(",", 1)[1]))
                else:
                    bv = g["bufferViews"][im["bufferView"]]
                    o = bv.get("byteOffset", 0)
                    decoded[src] = image_decoder(bytes(bufs[bv["buffer"]][o:o + bv["byteLength"]]))
            textures.append({"rgba8": decoded[src], "is_srgb": first_tex in srgb_tex})

    cat = lambda xs, w, dt: (np.concatenate(xs, 0) if xs else np.zeros((0, w), dt))
    pm = np.array(prims, dtype=PRIM_DTYPE) if prims else np.zeros(0, PRIM_DTYPE)
    nd = np.zeros(len(nodes_out), NODE_DTYPE)
    for i, (wm, pmi) in enumerate(nodes_out):
        nd[i]["worldMatrix"] = wm
        nd[i]["primMesh"] = pmi
    return FlatScene(cat(P, 3, np.float32), cat(N, 3, np.float32), cat(T, 4, np.float32), cat(UV, 2, np.float32),
                     np.concatenate(IDX) if IDX else np.zeros(0, np.uint32), pm, mats, L, nd, textures)


if __name__ == "__main__":
    sc = load_gltf(sys.argv[1])
    print("vertices", sc.positions.shape[0], "indices", sc.indices.shape[0], "primMeshes", len(sc.prim_meshes),
          "nodes", len(sc.nodes), "materials", len(sc.materials), "lights", len(sc.lights),
          "instanced tris", sc.instanced_triangle_count)
    if len(sys.argv) > 2:
        sc.save_npz(sys.argv[2])
