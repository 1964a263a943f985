"""ctypes access to the C++ host layer (libvkrt_host.so) for the test harness."""
import ctypes as C
import os

import numpy as np

from . import HOST_LIB_PATH, abi
from .flat_scene import FlatScene, LIGHT_DTYPE, MAT_DTYPE, NODE_DTYPE, PRIM_DTYPE

_lib = None


def lib():
    global _lib
    if _lib is None:
        import torch  # noqa: F401  (same HIP-runtime load-order rule as renderer.load_library)

        if not os.path.exists(HOST_LIB_PATH):
            raise RuntimeError(f"{HOST_LIB_PATH} missing: run __graft_entry__.build()")
        L = C.CDLL(HOST_LIB_PATH)
        L.vkrt_host_last_error.restype = C.c_char_p
        L.vkrt_host_load_gltf.argtypes = [C.c_char_p]
        L.vkrt_host_load_gltf.restype = C.c_void_p
        L.vkrt_host_free_scene.argtypes = [C.c_void_p]
        L.vkrt_host_scene_counts.argtypes = [C.c_void_p, C.c_void_p]
        L.vkrt_host_scene_copy.argtypes = [C.c_void_p] + [C.c_void_p] * 9
        L.vkrt_host_texture_info.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.vkrt_host_texture_copy.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.vkrt_host_global_uniforms.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.POINTER(abi.GlobalUniforms)]
        L.vkrt_host_parse_config.argtypes = [C.c_char_p, C.c_void_p, C.c_char_p, C.c_int]
        L.vkrt_host_parse_config.restype = C.c_int
        L.vkrt_host_render_gltf.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_uint32, C.c_void_p]
        L.vkrt_host_render_gltf.restype = C.c_int
        L.vkrt_host_render_gltf_hybrid.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.c_float, C.c_uint32, C.c_uint32, C.c_void_p]
        L.vkrt_host_render_gltf_hybrid.restype = C.c_int
        L.vkrt_host_decode_png.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64]
        L.vkrt_host_decode_png.restype = C.c_int
        L.vkrt_host_decode_image.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64]
        L.vkrt_host_decode_image.restype = C.c_int
        L.vkrt_host_write_png.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
        L.vkrt_host_write_png.restype = C.c_int
        L.vkrt_host_strip_rows.argtypes = [C.c_uint32] * 4
        L.vkrt_host_strip_rows.restype = C.c_uint32
        L.vkrt_host_strip_source.argtypes = [C.c_uint32] * 4 + [C.POINTER(C.c_uint32)] * 2
        L.vkrt_host_unpack_strips.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.vkrt_host_unpack_strips.restype = C.c_int
        _lib = L
    return _lib


def strip_rows(height, world, rank, strip_rows_=16):
    """rows of `rank`'s strip buffer in the C++ host's multi-GPU layout (host/strip_gather.h)"""
    return int(lib().vkrt_host_strip_rows(height, strip_rows_, world, rank))


def strip_source(height, world, y, strip_rows_=16):
    """(rank, local row) that holds global row y"""
    r, l = C.c_uint32(), C.c_uint32()
    lib().vkrt_host_strip_source(height, strip_rows_, world, y, C.byref(r), C.byref(l))
    return int(r.value), int(l.value)


def unpack_strips(gathered, full, world, strip_rows_=16, stream=None):
    """HIP un-interleave kernel of the C++ host: gathered [world, cap, W, 4] -> full [H, W, 4] (torch CUDA tensors)."""
    H, W = int(full.shape[0]), int(full.shape[1])
    rc = lib().vkrt_host_unpack_strips(C.c_void_p(gathered.data_ptr()), C.c_void_p(full.data_ptr()), W, H, strip_rows_, world,
                                       C.c_void_p(stream.cuda_stream) if stream is not None else None)
    if rc != 0:
        raise RuntimeError(lib().vkrt_host_last_error().decode())
    return full


def load_gltf(path):
    """C++ loader -> FlatScene."""
    L = lib()
    h = L.vkrt_host_load_gltf(os.fsencode(path))
    if not h:
        raise RuntimeError("vkrt_host_load_gltf: " + L.vkrt_host_last_error().decode())
    try:
        cnt = np.zeros(7, np.uint32)
        L.vkrt_host_scene_counts(h, cnt.ctypes.data)
        V, I, P, N, M, Lc, T = (int(x) for x in cnt)
        pos, nrm = np.zeros((V, 3), np.float32), np.zeros((V, 3), np.float32)
        tan, uv = np.zeros((V, 4), np.float32), np.zeros((V, 2), np.float32)
        idx = np.zeros(I, np.uint32)
        pm, nd = np.zeros(P, PRIM_DTYPE), np.zeros(N, NODE_DTYPE)
        mats, lights = np.zeros(M, MAT_DTYPE), np.zeros(Lc, LIGHT_DTYPE)
        L.vkrt_host_scene_copy(h, *(a.ctypes.data for a in (pos, nrm, tan, uv, idx, pm, nd, mats, lights)))
        tex = []
        for i in range(T):
            whs = np.zeros(3, np.uint32)
            L.vkrt_host_texture_info(h, i, whs.ctypes.data)
            px = np.zeros((int(whs[1]), int(whs[0]), 4), np.uint8)
            L.vkrt_host_texture_copy(h, i, px.ctypes.data)
            tex.append({"rgba8": px, "is_srgb": bool(whs[2])})
        return FlatScene(pos, nrm, tan, uv, idx, pm, mats, lights, nd, tex)
    finally:
        L.vkrt_host_free_scene(h)


def global_uniforms(eye=(0, 0, 15), center=(0, 0, 0), up=(0, 1, 0), fov=60.0, width=1280, height=720):
    u = abi.GlobalUniforms()
    e, c, p = (np.asarray(v, np.float32) for v in (eye, center, up))
    lib().vkrt_host_global_uniforms(e.ctypes.data, c.ctypes.data, p.ctypes.data, fov, width, height, C.byref(u))
    return u


def parse_config(text):
    out = np.zeros(13, np.int32)
    path = C.create_string_buffer(1024)
    rc = lib().vkrt_host_parse_config(text.encode(), out.ctypes.data, path, 1024)
    if rc != 0:
        raise ValueError(lib().vkrt_host_last_error().decode())
    keys = ["scene", "vsync", "width", "height", "samples", "depth", "frames", "seed", "nscenes", "framesPerCall", "watertight", "anyHitDissolve",
            "skipDeadShadowRays"]
    d = dict(zip(keys, (int(x) for x in out)))
    d["scene_path"] = path.value.decode()
    return d


def render_gltf(path, width, height, samples=1, depth=3, frames=1, seed0=0, eye=(0, 0, 15), center=(0, 0, 0), up=(0, 1, 0), fov=60.0,
                build=abi.VKRT_BUILD_PLOC_GPU, device=0):
    img = np.zeros((height, width, 4), np.float32)
    e, c, p = (np.asarray(v, np.float32) for v in (eye, center, up))
    rc = lib().vkrt_host_render_gltf(os.fsencode(path), device, width, height, samples, depth, frames, seed0, e.ctypes.data,
                                     c.ctypes.data, p.ctypes.data, fov, build, img.ctypes.data)
    if rc != 0:
        raise RuntimeError("vkrt_host_render_gltf: " + lib().vkrt_host_last_error().decode())
    return img


def render_gltf_hybrid(path, width, height, depth=3, frames=1, seed0=0, eye=(0, 0, 15), center=(0, 0, 0), up=(0, 1, 0), fov=60.0, rank=0, world=1, device=0):
    """The reference's hybrid frame sequence through the C++ HelloVkrt for one rank of `world`: that rank's display strips [rows, W, 4]."""
    rows = strip_rows(height, world, rank) if world > 1 else height
    img = np.zeros((rows, width, 4), np.float32)
    e, c, p = (np.asarray(v, np.float32) for v in (eye, center, up))
    rc = lib().vkrt_host_render_gltf_hybrid(os.fsencode(path), device, width, height, depth, frames, seed0, e.ctypes.data, c.ctypes.data, p.ctypes.data, fov,
                                            rank, world, img.ctypes.data)
    if rc != 0:
        raise RuntimeError("vkrt_host_render_gltf_hybrid: " + lib().vkrt_host_last_error().decode())
    return img


def decode_png(data):
    wh = np.zeros(2, np.uint32)
    buf = np.frombuffer(data, np.uint8)
    if lib().vkrt_host_decode_png(buf.ctypes.data, buf.size, wh.ctypes.data, None, 0) != 0:
        raise ValueError(lib().vkrt_host_last_error().decode())
    out = np.zeros((int(wh[1]), int(wh[0]), 4), np.uint8)
    lib().vkrt_host_decode_png(buf.ctypes.data, buf.size, wh.ctypes.data, out.ctypes.data, out.size)
    return out


def decode_image(data):
    """PNG or JPEG bytes -> (H, W, 4) uint8 through the C++ host's decoders (the texels the loader hands to the library)."""
    wh = np.zeros(2, np.uint32)
    buf = np.frombuffer(data, np.uint8)
    if lib().vkrt_host_decode_image(buf.ctypes.data, buf.size, wh.ctypes.data, None, 0) != 0:
        raise ValueError(lib().vkrt_host_last_error().decode())
    out = np.zeros((int(wh[1]), int(wh[0]), 4), np.uint8)
    lib().vkrt_host_decode_image(buf.ctypes.data, buf.size, wh.ctypes.data, out.ctypes.data, out.size)
    return out


def write_png(path, display_rgba):
    """8-bit RGBA PNG of a display image (float32 [H,W,4] in [0,1], i.e. after post.frag's gamma)."""
    a = np.ascontiguousarray(display_rgba, np.float32)
    if lib().vkrt_host_write_png(os.fsencode(path), a.ctypes.data, a.shape[1], a.shape[0]) != 0:
        raise RuntimeError(lib().vkrt_host_last_error().decode())
