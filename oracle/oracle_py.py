"""ctypes wrapper around oracle/liboracle.so (TEST INFRASTRUCTURE ONLY, see oracle.cpp).

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the
product package."""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
import vkrt_amd  # noqa: E402,F401
from vkrt_amd import abi  # noqa: E402

LIB_PATH = os.environ.get("ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")  # ORACLE_LIB: the sanitizer build (tests/test_host_asan.py)
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("oracle/liboracle.so missing: run `make -C oracle` or __graft_entry__.build()")
        L = C.CDLL(LIB_PATH)
        P = C.POINTER
        L.orc_last_error.restype = C.c_char_p
        L.orc_tea.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_tea.restype = C.c_uint32
        L.orc_lcg.argtypes = [P(C.c_uint32)]
        L.orc_lcg.restype = C.c_uint32
        L.orc_rnd.argtypes = [P(C.c_uint32)]
        L.orc_rnd.restype = C.c_float
        L.orc_scene_create.argtypes = [P(abi.SceneDesc)]
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_triangle_count.argtypes = [C.c_void_p]
        L.orc_triangle_count.restype = C.c_uint32
        L.orc_build_bvh.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_bvh_info.argtypes = [C.c_void_p, P(C.c_uint32), P(C.c_uint32), P(C.c_uint32), P(C.c_double)]
        L.orc_get_triangles.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_render_rows.argtypes = [C.c_void_p, P(abi.PushConstantRay), P(abi.GlobalUniforms), C.c_uint32, C.c_uint32,
                                      C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_render_rows.restype = C.c_int
        L.orc_pixel_log.argtypes = [C.c_void_p, P(abi.PushConstantRay), P(abi.GlobalUniforms), C.c_uint32, C.c_uint32,
                                    C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_pixel_log.restype = C.c_int
        L.orc_trace_rays.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_eval_math.argtypes = [C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_eval_shade.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_camera_ray.argtypes = [P(abi.GlobalUniforms), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_void_p]
        L.orc_gbuffer_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int, P(abi.GlobalUniforms), C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.orc_gbuffer_rows.restype = C.c_int
        L.orc_hybrid_rows.argtypes = [C.c_void_p, P(abi.PushConstantRay), P(abi.GlobalUniforms), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                      C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_hybrid_rows.restype = C.c_int
        L.orc_hybrid_pixel_rays.argtypes = [C.c_void_p, P(abi.PushConstantRay), P(abi.GlobalUniforms), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                            C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_hybrid_pixel_rays.restype = C.c_int
        L.orc_gbuffer_rows_nrd.argtypes = [C.c_void_p, C.c_void_p, C.c_int, P(abi.GlobalUniforms), C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_gbuffer_rows_nrd.restype = C.c_int
        L.orc_hybrid_rows_nrd.argtypes = [C.c_void_p, P(abi.PushConstantRay), P(abi.GlobalUniforms), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                          C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_hybrid_rows_nrd.restype = C.c_int
        L.orc_post.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_quantize_half.argtypes = [C.c_float]
        L.orc_quantize_half.restype = C.c_float
        L.orc_sample_texture.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_sample_texture_grad.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_set_gbuffer_mips.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_watertight.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_dissolve.argtypes = [C.c_void_p, C.c_int]
        L.orc_texture_levels.argtypes = [C.c_void_p, C.c_int]
        L.orc_texture_levels.restype = C.c_uint32
        L.orc_texture_level.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


COUNTER_NAMES = [n for n, _ in abi.Counters._fields_]


class OracleScene:
    def __init__(self, flat, build_bvh=True, max_leaf=4):
        self._flat = flat
        desc, self._keep = flat.to_desc()
        self._h = lib().orc_scene_create(C.byref(desc))
        if not self._h:
            raise RuntimeError("orc_scene_create: " + lib().orc_last_error().decode())
        self.has_bvh = False
        if build_bvh:
            self.build_bvh(max_leaf)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_scene_destroy(self._h)
            self._h = None

    @property
    def triangle_count(self):
        return int(lib().orc_triangle_count(self._h))

    def build_bvh(self, max_leaf=4):
        lib().orc_build_bvh(self._h, max_leaf)
        self.has_bvh = True

    def bvh_info(self):
        n, l, d, s = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_double()
        lib().orc_bvh_info(self._h, C.byref(n), C.byref(l), C.byref(d), C.byref(s))
        return {"nodes": n.value, "leaves": l.value, "max_depth": d.value, "sah_cost": s.value}

    def triangles(self):
        n = self.triangle_count
        v = np.zeros((n, 9), np.float32)
        ids = np.zeros((n, 3), np.uint32)
        lib().orc_get_triangles(self._h, v.ctypes.data, ids.ctypes.data)
        return v, ids

    def render(self, pc, cam, width, height, seed=0, flags=0, rows=None, image=None, use_bvh=True, threads=0):
        """Returns (image[nrows,W,4] float32, counters dict).  image is in/out when pc.frame > 0."""
        rows = np.arange(height, dtype=np.uint32) if rows is None else np.ascontiguousarray(rows, np.uint32)
        if image is None:
            image = np.zeros((rows.shape[0], width, 4), np.float32)
        assert image.shape == (rows.shape[0], width, 4) and image.dtype == np.float32 and image.flags.c_contiguous
        cnt = np.zeros(8, np.uint64)
        rc = lib().orc_render_rows(self._h, C.byref(pc), C.byref(cam), seed, flags, width, height, rows.ctypes.data,
                                   rows.shape[0], image.ctypes.data, 1 if use_bvh else 0, threads, cnt.ctypes.data)
        if rc != 0:
            raise RuntimeError("orc_render_rows: " + lib().orc_last_error().decode())
        return image, dict(zip(COUNTER_NAMES, (int(c) for c in cnt)))

    def pixel_log(self, pc, cam, width, height, x, y, seed=0, flags=0, use_bvh=True):
        px = np.zeros(4, np.float32)
        log = np.zeros(32 * 64 * max(1, pc.samples), np.float32)
        n = lib().orc_pixel_log(self._h, C.byref(pc), C.byref(cam), seed, flags, width, height, x, y, 1 if use_bvh else 0,
                                px.ctypes.data, log.ctypes.data, log.shape[0])
        return px, log[: min(n, log.shape[0])].reshape(-1, 8)

    def trace_rays(self, origins, directions, tmin=0.001, tmax=10000.0, any_hit=False, use_bvh=True):
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, np.float32).reshape(-1, 3)
        n = o.shape[0]
        t, u, v = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
        gid = np.zeros(n, np.int32)
        cnt = np.zeros(8, np.uint64)
        lib().orc_trace_rays(self._h, n, o.ctypes.data, d.ctypes.data, tmin, tmax, 1 if any_hit else 0, 1 if use_bvh else 0,
                             t.ctypes.data, u.ctypes.data, v.ctypes.data, gid.ctypes.data, cnt.ctypes.data)
        return t, u, v, gid, {"nodes_visited": int(cnt[6]), "tris_tested": int(cnt[7])}

    def gbuffer(self, cam, width, height, lights_count, clear_color=(1.0, 1.0, 1.0, 1.0), rows=None, use_bvh=True, threads=0):
        rows = np.arange(height, dtype=np.uint32) if rows is None else np.ascontiguousarray(rows, np.uint32)
        n = rows.shape[0]
        g = {"color": np.zeros((n, width, 4), np.float32), "position": np.zeros((n, width, 4), np.float32),
             "normal": np.zeros((n, width, 4), np.float32), "roughMetal": np.zeros((n, width, 2), np.float32)}
        cc = np.asarray(clear_color, np.float32)
        rc = lib().orc_gbuffer_rows(self._h, cc.ctypes.data, lights_count, C.byref(cam), width, height, rows.ctypes.data, n, g["color"].ctypes.data,
                                    g["position"].ctypes.data, g["normal"].ctypes.data, g["roughMetal"].ctypes.data, 1 if use_bvh else 0, threads)
        if rc != 0:
            raise RuntimeError("orc_gbuffer_rows: " + lib().orc_last_error().decode())
        return g

    def hybrid(self, pc, cam, width, height, g, seed=0, flags=0, rows=None, accum=None, use_bvh=True, threads=0):
        rows = np.arange(height, dtype=np.uint32) if rows is None else np.ascontiguousarray(rows, np.uint32)
        n = rows.shape[0]
        if accum is None:
            accum = np.zeros((n, width, 4), np.float32)
        cnt = np.zeros(8, np.uint64)
        rc = lib().orc_hybrid_rows(self._h, C.byref(pc), C.byref(cam), seed, flags, width, height, rows.ctypes.data, n, g["color"].ctypes.data,
                                   g["position"].ctypes.data, g["normal"].ctypes.data, g["roughMetal"].ctypes.data, accum.ctypes.data,
                                   1 if use_bvh else 0, threads, cnt.ctypes.data)
        if rc != 0:
            raise RuntimeError("orc_hybrid_rows: " + lib().orc_last_error().decode())
        return accum, dict(zip(COUNTER_NAMES, (int(c) for c in cnt)))

    def hybrid_pixel_rays(self, pc, cam, width, x, y, g, seed=0, flags=0, use_bvh=False):
        """The rays raytraceHybrid.rgen traces for pixel (x, y), in order: (accum texel, array [n, 9] = o, d, tmin, tmax, any flag).
        g: G-buffer planes of the whole frame (numpy)."""
        gp = np.concatenate([g["color"][y, x], g["position"][y, x], g["normal"][y, x], g["roughMetal"][y, x]]).astype(np.float32)
        acc = np.zeros(4, np.float32)
        rays = np.zeros(9 * 4096, np.float32)
        n = lib().orc_hybrid_pixel_rays(self._h, C.byref(pc), C.byref(cam), seed, flags, width, x, y, gp.ctypes.data, 1 if use_bvh else 0,
                                        acc.ctypes.data, rays.ctypes.data, rays.shape[0])
        return acc, rays[: min(n, rays.shape[0])].reshape(-1, 9)

    def gbuffer_nrd(self, cam, view_matrix, width, height, lights_count, clear_color=(1.0, 1.0, 1.0, 1.0), rows=None, use_bvh=True):
        """gbuffer() plus the NRD front-end planes of the raster pass (frag_shader.frag:133-136)."""
        rows = np.arange(height, dtype=np.uint32) if rows is None else np.ascontiguousarray(rows, np.uint32)
        n = rows.shape[0]
        g = {"color": np.zeros((n, width, 4), np.float32), "position": np.zeros((n, width, 4), np.float32),
             "normal": np.zeros((n, width, 4), np.float32), "roughMetal": np.zeros((n, width, 2), np.float32),
             "nrdNormalRoughness": np.zeros((n, width, 4), np.float32), "nrdViewZ": np.zeros((n, width), np.float32)}
        cc = np.asarray(clear_color, np.float32)
        vm = np.ascontiguousarray(view_matrix, np.float32).reshape(16)
        rc = lib().orc_gbuffer_rows_nrd(self._h, cc.ctypes.data, lights_count, C.byref(cam), vm.ctypes.data, width, height, rows.ctypes.data, n,
                                        g["color"].ctypes.data, g["position"].ctypes.data, g["normal"].ctypes.data, g["roughMetal"].ctypes.data,
                                        g["nrdNormalRoughness"].ctypes.data, g["nrdViewZ"].ctypes.data, 1 if use_bvh else 0)
        if rc != 0:
            raise RuntimeError("orc_gbuffer_rows_nrd: " + lib().orc_last_error().decode())
        return g

    def hybrid_nrd(self, pc, cam, width, height, g, seed=0, flags=0, rows=None, accum=None, use_bvh=True):
        """hybrid() plus the REBLUR input plane (raytraceHybrid.rgen:273-281); returns (accum, radianceHitDist)."""
        rows = np.arange(height, dtype=np.uint32) if rows is None else np.ascontiguousarray(rows, np.uint32)
        n = rows.shape[0]
        if accum is None:
            accum = np.zeros((n, width, 4), np.float32)
        rad = np.zeros((n, width, 4), np.float32)
        rc = lib().orc_hybrid_rows_nrd(self._h, C.byref(pc), C.byref(cam), seed, flags, width, height, rows.ctypes.data, n, g["color"].ctypes.data,
                                       g["position"].ctypes.data, g["normal"].ctypes.data, g["roughMetal"].ctypes.data, g["nrdViewZ"].ctypes.data,
                                       accum.ctypes.data, rad.ctypes.data, 1 if use_bvh else 0)
        if rc != 0:
            raise RuntimeError("orc_hybrid_rows_nrd: " + lib().orc_last_error().decode())
        return accum, rad

    def sample_texture(self, tex_index, uv):
        uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
        out = np.zeros((uv.shape[0], 4), np.float32)
        lib().orc_sample_texture(self._h, tex_index, uv.shape[0], uv.ctypes.data, out.ctypes.data)
        return out

    def sample_texture_grad(self, tex_index, uv, grads):
        """texture() with explicit derivatives (dudx, dvdx, dudy, dvdy per sample): trilinear + anisotropy 4."""
        uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
        grads = np.ascontiguousarray(grads, np.float32).reshape(-1, 4)
        assert grads.shape[0] == uv.shape[0]
        out = np.zeros((uv.shape[0], 4), np.float32)
        lib().orc_sample_texture_grad(self._h, tex_index, uv.shape[0], uv.ctypes.data, grads.ctypes.data, out.ctypes.data)
        return out

    def set_watertight(self, on):
        """ray/triangle test of every later query: False = Moeller-Trumbore (default), True = the watertight test (VKRT_OPT_WATERTIGHT)"""
        lib().orc_set_watertight(self._h, 1 if on else 0)

    def set_dissolve(self, on):
        """any-hit alpha / dissolve stage (VKRT_OPT_ANYHIT_DISSOLVE): False = all geometry opaque (default)"""
        lib().orc_set_dissolve(self._h, 1 if on else 0)

    def set_gbuffer_mips(self, on):
        lib().orc_set_gbuffer_mips(self._h, 1 if on else 0)

    def texture_levels(self, tex_index):
        """The mip chain of one texture as a list of (h, w, 4) uint8 arrays (level 0 first)."""
        out = []
        for level in range(lib().orc_texture_levels(self._h, tex_index)):
            w, h = C.c_uint32(), C.c_uint32()
            lib().orc_texture_level(self._h, tex_index, level, C.byref(w), C.byref(h), None)
            px = np.zeros((h.value, w.value, 4), np.uint8)
            lib().orc_texture_level(self._h, tex_index, level, None, None, px.ctypes.data)
            out.append(px)
        return out


def tea(a, b):
    return int(lib().orc_tea(a & 0xFFFFFFFF, b & 0xFFFFFFFF))


def lcg_sequence(state, n):
    s = C.c_uint32(state)
    out = []
    for _ in range(n):
        bits = int(lib().orc_lcg(C.byref(s)))
        out.append((int(s.value), bits, bits / 16777216.0))
    return out


def eval_math(op, a, b=None):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(a if b is None else b, np.float32)
    out = np.zeros_like(a)
    lib().orc_eval_math(op, a.size, a.ctypes.data, b.ctypes.data, out.ctypes.data)
    return out


def eval_shade(inputs):
    inputs = np.ascontiguousarray(inputs, np.float32).reshape(-1, 40)
    out = np.zeros((inputs.shape[0], 20), np.float32)
    lib().orc_eval_shade(inputs.shape[0], inputs.ctypes.data, out.ctypes.data)
    return out


def camera_ray(cam, x, y, W, H, jx=0.5, jy=0.5):
    out = np.zeros(6, np.float32)
    lib().orc_camera_ray(C.byref(cam), x, y, W, H, jx, jy, out.ctypes.data)
    return out[:3], out[3:]


def post(main_img, rt_img, rt_mode=0, view_accumulated=0, use_gi=0):
    m = np.ascontiguousarray(main_img, np.float32)
    r = np.ascontiguousarray(rt_img if rt_img is not None else main_img, np.float32)
    out = np.zeros_like(m)
    lib().orc_post(rt_mode, view_accumulated, use_gi, m.size // 4, m.ctypes.data, r.ctypes.data, out.ctypes.data)
    return out


def quantize_half(x):
    return np.array([lib().orc_quantize_half(float(v)) for v in np.asarray(x, np.float32).ravel()], np.float32).reshape(np.shape(x))


def algorithmic_bytes(counters, frame_gt0=False):
    """SURVEY.md 8(d) contract figure: 64 B/visited node, 48 B/triangle test, 220 B/closest hit,
    32 B/diffuse hit (light), 16 B/texture tap, 16 B/pixel written (+16 read when frame>0)."""
    c = counters
    return (64 * c["nodes_visited"] + 48 * c["tris_tested"] + 220 * c["hits"] + 32 * c["diffuse_hits"]
            + 16 * c["tex_taps"] + (32 if frame_gt0 else 16) * c["pixels"])
