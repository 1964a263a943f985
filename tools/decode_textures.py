"""Texture pre-pass for glTF scenes whose images the native loader cannot decode (CMYK / arithmetic-coded JPEG, interlaced
PNG, other formats), or whose decode should be pinned to Pillow's.

The C++ host layer (vk-raytracing-engine_amd/host/gltf_loader.cpp: decodeImageFile) decodes PNG and JPEG itself
(jpeg_decode.cpp) and first looks for `<image uri>.rgba8` next to the image: two little-endian u32 (width, height) followed by width*height RGBA8 texels,
rows top to bottom -- the layout stb_image hands the reference (hello_vulkan.cpp:482-485, 4 channels forced).  This tool
writes those sidecars with PIL.  Usage:

    python tools/decode_textures.py scene.gltf [--all] [--force]

By default only images that are neither PNG nor JPEG get a sidecar; --all converts every external image (the sidecar wins over
the native decoders); --force overwrites existing sidecars.  Embedded (bufferView / data-URI) images are not handled:
export the scene with external images first.
"""
import argparse
import json
import os
import struct
import sys
import urllib.parse


def write_sidecar(src, dst):
    from PIL import Image  # imported here so the module can be inspected without PIL

    with Image.open(src) as im:
        rgba = im.convert("RGBA")  # grey -> r=g=b, missing alpha -> 255 (stb_image's req_comp=4 rule)
        w, h = rgba.size
        data = rgba.tobytes()
    with open(dst, "wb") as f:
        f.write(struct.pack("<II", w, h))
        f.write(data)
    return w, h


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("gltf")
    ap.add_argument("--all", action="store_true", help="also convert PNG and JPEG images")
    ap.add_argument("--force", action="store_true", help="overwrite existing sidecars")
    a = ap.parse_args(argv)
    if a.gltf.lower().endswith(".glb"):
        sys.exit("decode_textures: .glb keeps its images in the binary chunk; export as .gltf with external images first")
    doc = json.load(open(a.gltf))
    base = os.path.dirname(os.path.abspath(a.gltf))
    done = 0
    for i, img in enumerate(doc.get("images", [])):
        uri = img.get("uri")
        if not uri or uri.startswith("data:"):
            print(f"image {i}: embedded, skipped")
            continue
        path = os.path.join(base, urllib.parse.unquote(uri))
        if not a.all and path.lower().endswith((".png", ".jpg", ".jpeg")):
            continue
        dst = path + ".rgba8"
        if os.path.exists(dst) and not a.force:
            continue
        w, h = write_sidecar(path, dst)
        print(f"image {i}: {uri} -> {os.path.basename(dst)} ({w}x{h})")
        done += 1
    print(f"{done} sidecar(s) written")
    return 0


if __name__ == "__main__":
    sys.exit(main())
