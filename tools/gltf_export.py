"""Writes a FlatScene (harness container) as glTF 2.0 (.gltf + .bin + PNG textures, or .glb), so that the
C++ host loader (vk-raytracing-engine_amd/host/gltf_loader.cpp) and the CLI can be driven with
generated scenes on machines without the reference assets.  One glTF mesh per primMesh, one node per
flat node (matrix form); lights are written only when `write_lights`."""
import base64
import io
import json
import os
import struct

import numpy as np


def _png_bytes(rgba):
    from PIL import Image

    buf = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(rgba, np.uint8), "RGBA").save(buf, format="PNG")
    return buf.getvalue()


def export_gltf(flat, path, glb=False, embed=False, write_lights=False, index_type=None):
    base = os.path.dirname(os.path.abspath(path))
    stem = os.path.splitext(os.path.basename(path))[0]
    blob = bytearray()
    views, accessors = [], []

    def add_view(data, target=None):
        while len(blob) % 4:
            blob.append(0)
        v = {"buffer": 0, "byteOffset": len(blob), "byteLength": len(data)}
        if target:
            v["target"] = target
        blob.extend(data)
        views.append(v)
        return len(views) - 1

    def add_accessor(arr, ctype, atype, minmax=False, target=None):
        arr = np.ascontiguousarray(arr)
        a = {"bufferView": add_view(arr.tobytes(), target), "componentType": ctype, "count": int(arr.shape[0]), "type": atype}
        if minmax:
            a["min"] = [float(x) for x in arr.min(axis=0)]
            a["max"] = [float(x) for x in arr.max(axis=0)]
        accessors.append(a)
        return len(accessors) - 1

    meshes = []
    attr_cache = {}
    for pm in flat.prim_meshes:
        vo, vc = int(pm["vertexOffset"]), int(pm["vertexCount"])
        key = (vo, vc)
        if key not in attr_cache:
            sl = slice(vo, vo + vc)
            attr_cache[key] = {
                "POSITION": add_accessor(flat.positions[sl], 5126, "VEC3", True, 34962),
                "NORMAL": add_accessor(flat.normals[sl], 5126, "VEC3", False, 34962),
                "TANGENT": add_accessor(flat.tangents[sl], 5126, "VEC4", False, 34962),
                "TEXCOORD_0": add_accessor(flat.texcoords0[sl], 5126, "VEC2", False, 34962),
            }
        idx = flat.indices[int(pm["firstIndex"]): int(pm["firstIndex"]) + int(pm["indexCount"])]
        it = index_type or (np.uint16 if vc <= 65535 else np.uint32)
        ct = {np.uint8: 5121, np.uint16: 5123, np.uint32: 5125}[it]
        prim = {"attributes": dict(attr_cache[key]), "indices": add_accessor(idx.astype(it), ct, "SCALAR", False, 34963), "mode": 4}
        if int(pm["materialIndex"]) >= 0:
            prim["material"] = int(pm["materialIndex"])
        meshes.append({"primitives": [prim]})

    nodes = [{"mesh": int(n["primMesh"]), "matrix": [float(x) for x in n["worldMatrix"]]} for n in flat.nodes]
    g = {"asset": {"version": "2.0", "generator": "vkrt tools/gltf_export.py"}, "scene": 0}
    images, textures = [], []
    for i, t in enumerate(flat.textures):
        png = _png_bytes(t["rgba8"])
        if glb or embed:
            if glb:
                images.append({"bufferView": add_view(png), "mimeType": "image/png"})
            else:
                images.append({"uri": "data:image/png;base64," + base64.b64encode(png).decode()})
        else:
            name = f"{stem}_tex{i}.png"
            with open(os.path.join(base, name), "wb") as f:
                f.write(png)
            images.append({"uri": name})
        textures.append({"source": i})
    mats = []
    for m in flat.materials:
        pbr = {"baseColorFactor": [float(x) for x in m["pbrBaseColorFactor"]], "metallicFactor": float(m["metallicFactor"]),
               "roughnessFactor": float(m["roughnessFactor"])}
        if m["pbrBaseColorTexture"] >= 0:
            pbr["baseColorTexture"] = {"index": int(m["pbrBaseColorTexture"])}
        if m["metallicRoughnessTexture"] >= 0:
            pbr["metallicRoughnessTexture"] = {"index": int(m["metallicRoughnessTexture"])}
        mj = {"pbrMetallicRoughness": pbr, "emissiveFactor": [float(x) for x in m["emissiveFactor"]]}
        if m["normalTexture"] >= 0:
            mj["normalTexture"] = {"index": int(m["normalTexture"])}
        if m["emissiveTexture"] >= 0:
            mj["emissiveTexture"] = {"index": int(m["emissiveTexture"])}
        mats.append(mj)
    if write_lights:
        names = {0: "point", 1: "directional", 2: "spot"}
        g["extensionsUsed"] = ["KHR_lights_punctual"]
        g["extensions"] = {"KHR_lights_punctual": {"lights": [
            {"type": names[int(l["type"])], "color": [float(x) for x in l["color"]], "intensity": float(l["intensity"])} for l in flat.lights]}}
        for i, l in enumerate(flat.lights):
            nodes.append({"translation": [float(x) for x in l["position"]], "extensions": {"KHR_lights_punctual": {"light": i}}})
    g["scenes"] = [{"nodes": list(range(len(nodes)))}]
    g["nodes"] = nodes
    g["meshes"] = meshes
    g["materials"] = mats
    if images:
        g["images"], g["textures"] = images, textures
    g["accessors"], g["bufferViews"] = accessors, views
    while len(blob) % 4:
        blob.append(0)
    if glb:
        g["buffers"] = [{"byteLength": len(blob)}]
        js = json.dumps(g).encode()
        js += b" " * ((4 - len(js) % 4) % 4)
        with open(path, "wb") as f:
            f.write(b"glTF" + struct.pack("<II", 2, 12 + 8 + len(js) + 8 + len(blob)))
            f.write(struct.pack("<II", len(js), 0x4E4F534A) + js)
            f.write(struct.pack("<II", len(blob), 0x004E4942) + bytes(blob))
    else:
        if embed:
            g["buffers"] = [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(bytes(blob)).decode()}]
        else:
            with open(os.path.join(base, stem + ".bin"), "wb") as f:
                f.write(bytes(blob))
            g["buffers"] = [{"byteLength": len(blob), "uri": stem + ".bin"}]
        with open(path, "w") as f:
            json.dump(g, f)
    return path
