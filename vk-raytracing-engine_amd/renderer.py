"""Harness-side wrapper of the C ABI (include/vkrt.h): torch owns device memory and streams,
libvkrt.so does all the work.  Mirrors the call order of the reference's main():
loadGltfScene -> createBottomLevelASGltf/createTopLevelAsGltf -> per frame pathtrace
(main.cpp:226-240, 504-508)."""
import ctypes as C
import os

import numpy as np

from . import LIB_PATH, abi

_lib = None


def load_library():
    """Load libvkrt.so; fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        # torch bundles its own libamdhip64.so.7; it must be the first HIP runtime mapped into the
        # process (loading /opt/rocm's copy first leaves torch with "No HIP GPUs are available").
        import torch  # noqa: F401

        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} missing: build it with __graft_entry__.build() "
                               "(there is no CPU fallback for the ray-tracing path)")
        _lib = abi.declare_vkrt(C.CDLL(LIB_PATH))
        if _lib.vkrt_abi_version() != abi.VKRT_ABI_VERSION:
            raise RuntimeError("libvkrt.so ABI version mismatch")
    return _lib


class VkrtError(RuntimeError):
    pass


def _check(rc, what):
    if rc != 0:
        raise VkrtError(f"{what} failed ({rc}): {load_library().vkrt_last_error().decode()}")


def whole_image_shard(width, height):
    return abi.Shard(width, height, 0, 1, 0)


class Renderer:
    def __init__(self, flat, device=0, build="ploc", options=None):
        """options: {abi.VKRT_OPT_*: value} applied before the build (per-handle execution options, include/vkrt.h)."""
        import time

        self.lib = load_library()
        self.device = device
        desc, keep = flat.to_desc()
        h = C.c_void_p()
        t0 = time.perf_counter()
        _check(self.lib.vkrt_scene_create(C.byref(desc), device, C.byref(h)), "vkrt_scene_create")
        self.upload_ms = (time.perf_counter() - t0) * 1e3  # scene upload (hello_vulkan.cpp:353-381), host wall time
        self._h = h
        del keep
        self.lights_count = int(flat.lights.shape[0])
        for k, v in (options or {}).items():
            self.set_option(k, v)
        if build:
            self.build(build)

    def set_option(self, option, value):
        _check(self.lib.vkrt_scene_set_option(self._h, int(option), int(value)), "vkrt_scene_set_option")

    def get_option(self, option):
        v = C.c_int()
        _check(self.lib.vkrt_scene_get_option(self._h, int(option), C.byref(v)), "vkrt_scene_get_option")
        return int(v.value)

    def reserve(self, shard, stream=None, frames_per_call=None):
        """Size the working set for launches of this shard geometry (no allocation / host sync inside later pathtrace calls);
        frames_per_call: the frames the caller hands to one pathtrace_frames call (None: whatever the options keep in flight)."""
        st = C.c_void_p(stream.cuda_stream) if stream is not None else None
        if frames_per_call is None:
            _check(self.lib.vkrt_reserve(self._h, C.byref(shard), st), "vkrt_reserve")
        else:
            _check(self.lib.vkrt_reserve_frames(self._h, C.byref(shard), int(frames_per_call), st), "vkrt_reserve_frames")

    def build(self, kind="ploc"):
        flags = {"sah": abi.VKRT_BUILD_SAH_HOST, "lbvh": abi.VKRT_BUILD_LBVH_GPU, "ploc": abi.VKRT_BUILD_PLOC_GPU}[kind]
        _check(self.lib.vkrt_accel_build(self._h, flags, None), "vkrt_accel_build")
        self.build_kind = kind

    def close(self):
        if getattr(self, "_h", None):
            self.lib.vkrt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def accel_info(self):
        info = abi.AccelInfo()
        _check(self.lib.vkrt_accel_get_info(self._h, C.byref(info)), "vkrt_accel_get_info")
        return {n: getattr(info, n) for n, _ in info._fields_}

    def check_accel(self):
        """Structural check of the built tree (vkrt_debug_check_accel): dict of counts; a sound tree has every triangle once and
        no violations."""
        c = abi.AccelCheck()
        _check(self.lib.vkrt_debug_check_accel(self._h, C.byref(c)), "vkrt_debug_check_accel")
        return {n: getattr(c, n) for n, _ in c._fields_}

    def shard_rows(self, shard):
        return int(self.lib.vkrt_shard_rows(C.byref(shard)))

    def pathtrace(self, pc, cam, width, height, seed=0, flags=0, shard=None, image=None, stream=None):
        """One launch (one frame).  image: torch float32 CUDA tensor [rows, width, 4] (in/out when pc.frame > 0)."""
        import torch

        shard = shard or whole_image_shard(width, height)
        rows = self.shard_rows(shard)
        if image is None:
            image = torch.zeros((rows, width, 4), dtype=torch.float32, device=f"cuda:{self.device}")
        assert image.is_cuda and image.dtype == torch.float32 and image.is_contiguous() and tuple(image.shape) == (rows, width, 4)
        if stream is None:
            stream = torch.cuda.current_stream(image.device)
        opts = abi.TraceOpts(seed & 0xFFFFFFFF, flags)
        _check(self.lib.vkrt_pathtrace(self._h, C.byref(pc), C.byref(cam), C.byref(opts), C.byref(shard),
                                       C.c_void_p(image.data_ptr()), C.c_void_p(stream.cuda_stream)), "vkrt_pathtrace")
        return image

    def pathtrace_frames(self, pc, cam, width, height, n_frames, seed=0, flags=0, shard=None, image=None, stream=None):
        """n_frames progressive frames in one call (vkrt_pathtrace_frames): frame i uses pc.frame + i and seed + i."""
        import torch

        shard = shard or whole_image_shard(width, height)
        rows = self.shard_rows(shard)
        if image is None:
            image = torch.zeros((rows, width, 4), dtype=torch.float32, device=f"cuda:{self.device}")
        assert image.is_cuda and image.dtype == torch.float32 and image.is_contiguous() and tuple(image.shape) == (rows, width, 4)
        if stream is None:
            stream = torch.cuda.current_stream(image.device)
        opts = abi.TraceOpts(seed & 0xFFFFFFFF, flags)
        _check(self.lib.vkrt_pathtrace_frames(self._h, C.byref(pc), C.byref(cam), C.byref(opts), C.byref(shard),
                                              C.c_void_p(image.data_ptr()), int(n_frames), C.c_void_p(stream.cuda_stream)), "vkrt_pathtrace_frames")
        return image

    # ---- hybrid mode (reference rtMode == 0) ---------------------------------------------------------
    def gbuffer_raycast(self, cam, width, height, lights_count=None, clear_color=(1.0, 1.0, 1.0, 1.0), shard=None, stream=None, view_matrix=None):
        """Stand-in for rasterizeGltf: returns dict of torch CUDA planes color/position/normal [rows,W,4], roughMetal [rows,W,2].
        view_matrix (16 floats, column-major pcRaster.viewMatrix): also the NRD front-end planes nrdNormalRoughness [rows,W,4],
        nrdViewZ [rows,W], nrdRadianceHitDist [rows,W,4] (vkrt_gbuffer_raycast_nrd)."""
        import torch

        shard = shard or whole_image_shard(width, height)
        rows = self.shard_rows(shard)
        dev = f"cuda:{self.device}"
        g = {"color": torch.zeros((rows, width, 4), dtype=torch.float32, device=dev), "position": torch.zeros((rows, width, 4), dtype=torch.float32, device=dev),
             "normal": torch.zeros((rows, width, 4), dtype=torch.float32, device=dev), "roughMetal": torch.zeros((rows, width, 2), dtype=torch.float32, device=dev)}
        stream = stream or torch.cuda.current_stream(g["color"].device)
        gb = abi.Gbuffer(*(g[k].data_ptr() for k in ("color", "position", "normal", "roughMetal")))
        cc = (C.c_float * 4)(*clear_color)
        n = self.lights_count if lights_count is None else lights_count
        if view_matrix is not None:
            g["nrdNormalRoughness"] = torch.zeros((rows, width, 4), dtype=torch.float32, device=dev)
            g["nrdViewZ"] = torch.zeros((rows, width), dtype=torch.float32, device=dev)
            g["nrdRadianceHitDist"] = torch.zeros((rows, width, 4), dtype=torch.float32, device=dev)
            nrd = abi.NrdPlanes(g["nrdNormalRoughness"].data_ptr(), g["nrdViewZ"].data_ptr(), g["nrdRadianceHitDist"].data_ptr())
            vm = (C.c_float * 16)(*[float(v) for v in view_matrix])
            _check(self.lib.vkrt_gbuffer_raycast_nrd(self._h, C.byref(cc), n, C.byref(cam), C.byref(vm), C.byref(shard), C.byref(gb), C.byref(nrd),
                                                     C.c_void_p(stream.cuda_stream)), "vkrt_gbuffer_raycast_nrd")
            return g
        _check(self.lib.vkrt_gbuffer_raycast(self._h, C.byref(cc), n, C.byref(cam), C.byref(shard), C.byref(gb), C.c_void_p(stream.cuda_stream)),
               "vkrt_gbuffer_raycast")
        return g

    def hybrid_trace(self, pc, cam, width, height, gbuffer, seed=0, flags=0, shard=None, accum=None, stream=None):
        """raytraceHybrid.rgen: accum [rows,W,4] (in/out when pc.frame > 0)."""
        import torch

        shard = shard or whole_image_shard(width, height)
        rows = self.shard_rows(shard)
        if accum is None:
            accum = torch.zeros((rows, width, 4), dtype=torch.float32, device=f"cuda:{self.device}")
        stream = stream or torch.cuda.current_stream(accum.device)
        gb = abi.Gbuffer(*(gbuffer[k].data_ptr() for k in ("color", "position", "normal", "roughMetal")))
        opts = abi.TraceOpts(seed & 0xFFFFFFFF, flags)
        if "nrdViewZ" in gbuffer:  # planes from gbuffer_raycast(view_matrix=...): also pack the REBLUR input (rgen:273-281)
            nrd = abi.NrdPlanes(gbuffer["nrdNormalRoughness"].data_ptr(), gbuffer["nrdViewZ"].data_ptr(), gbuffer["nrdRadianceHitDist"].data_ptr())
            _check(self.lib.vkrt_hybrid_trace_nrd(self._h, C.byref(pc), C.byref(cam), C.byref(opts), C.byref(shard), C.byref(gb), C.byref(nrd),
                                                  C.c_void_p(accum.data_ptr()), C.c_void_p(stream.cuda_stream)), "vkrt_hybrid_trace_nrd")
            return accum
        _check(self.lib.vkrt_hybrid_trace(self._h, C.byref(pc), C.byref(cam), C.byref(opts), C.byref(shard), C.byref(gb),
                                          C.c_void_p(accum.data_ptr()), C.c_void_p(stream.cuda_stream)), "vkrt_hybrid_trace")
        return accum

    def post(self, main_img, rt_img=None, rt_mode=0, view_accumulated=0, use_gi=0, stream=None):
        """post.frag composite + gamma; returns a new [.,.,4] tensor."""
        import torch

        out = torch.empty_like(main_img)
        stream = stream or torch.cuda.current_stream(main_img.device)
        pcp = abi.PushConstantPost(1.0, rt_mode, view_accumulated, use_gi)
        _check(self.lib.vkrt_post(self.device, C.byref(pcp), main_img.numel() // 4, C.c_void_p(main_img.data_ptr()),
                                  C.c_void_p(rt_img.data_ptr()) if rt_img is not None else None, C.c_void_p(out.data_ptr()),
                                  C.c_void_p(stream.cuda_stream)), "vkrt_post")
        return out

    def reset_counters(self, stream=None):
        _check(self.lib.vkrt_counters_reset(self._h, C.c_void_p(stream.cuda_stream) if stream is not None else None),
               "vkrt_counters_reset")

    def counters(self):
        c = abi.Counters()
        _check(self.lib.vkrt_counters_read(self._h, C.byref(c)), "vkrt_counters_read")
        return c.as_dict()

    def last_trace_ms(self):
        ms = C.c_float()
        _check(self.lib.vkrt_last_trace_ms(self._h, C.byref(ms)), "vkrt_last_trace_ms")
        return float(ms.value)

    def last_trace_timing(self):
        t = abi.TraceTiming()
        _check(self.lib.vkrt_last_trace_timing(self._h, C.byref(t)), "vkrt_last_trace_timing")
        return {"total_ms": float(t.total_ms), "traverse_ms": float(t.traverse_ms), "traverse_launches": int(t.traverse_launches),
                "mode": "wavefront" if t.mode == 1 else "megakernel", "shade_ms": float(t.shade_ms), "shade_launches": int(t.shade_launches)}

    def trace_rays(self, origins, directions, tmin=0.001, tmax=10000.0, any_hit=False):
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, np.float32).reshape(-1, 3)
        n = o.shape[0]
        t, u, v = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
        gid = np.zeros(n, np.int32)
        _check(self.lib.vkrt_debug_trace_rays(self._h, n, o.ctypes.data, d.ctypes.data, tmin, tmax, 1 if any_hit else 0,
                                              t.ctypes.data, u.ctypes.data, v.ctypes.data, gid.ctypes.data), "vkrt_debug_trace_rays")
        return t, u, v, gid


def eval_math(op, a, b=None, device=0):
    lib = load_library()
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(a if b is None else b, np.float32)
    out = np.zeros_like(a)
    _check(lib.vkrt_debug_eval_math(device, op, a.size, a.ctypes.data, b.ctypes.data, out.ctypes.data), "vkrt_debug_eval_math")
    return out
