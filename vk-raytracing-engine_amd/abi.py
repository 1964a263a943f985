"""ctypes mirror of include/vkrt.h and include/vkrt_host_device.h.

Harness-side only: the product is the C-ABI library (csrc/ -> libvkrt.so); this module
declares its structs and prototypes so tests and bench.py can call it.  Layout sizes
are asserted against the reference contract (reference: shaders/host_device.h:68-137,
SURVEY.md Appendix B).
"""
import ctypes as C

c_f = C.c_float
c_i = C.c_int32
c_u = C.c_uint32
c_u64 = C.c_uint64


class Mat4(C.Structure):
    _fields_ = [("m", c_f * 16)]


class GlobalUniforms(C.Structure):
    _fields_ = [("viewProj", Mat4), ("viewInverse", Mat4), ("projInverse", Mat4)]


class PushConstantRay(C.Structure):
    _fields_ = [
        ("clearColor", c_f * 4),
        ("frame", c_i),
        ("lightsCount", c_i),
        ("samples", c_i),
        ("depth", c_i),
        ("useShadows", c_i),
        ("useAO", c_i),
        ("useGI", c_i),
    ]


class PrimMeshInfo(C.Structure):
    _fields_ = [("indexOffset", c_u), ("vertexOffset", c_u), ("materialIndex", c_i)]


class GltfPBRMaterial(C.Structure):
    _fields_ = [
        ("pbrBaseColorFactor", c_f * 4),
        ("pbrBaseColorTexture", c_i),
        ("metallicFactor", c_f),
        ("roughnessFactor", c_f),
        ("metallicRoughnessTexture", c_i),
        ("normalTexture", c_i),
        ("emissiveFactor", c_f * 3),
        ("emissiveTexture", c_i),
    ]


class GltfLight(C.Structure):
    _fields_ = [("position", c_f * 3), ("color", c_f * 3), ("intensity", c_f), ("type", c_i)]


class PrimMesh(C.Structure):
    _fields_ = [
        ("firstIndex", c_u),
        ("indexCount", c_u),
        ("vertexOffset", c_u),
        ("vertexCount", c_u),
        ("materialIndex", c_i),
    ]


class Node(C.Structure):
    _fields_ = [("worldMatrix", c_f * 16), ("primMesh", c_i)]


class Texture(C.Structure):
    _fields_ = [("width", c_u), ("height", c_u), ("rgba8", C.c_void_p), ("is_srgb", c_i)]


class SceneDesc(C.Structure):
    _fields_ = [
        ("struct_size", c_u),
        ("vertex_count", c_u),
        ("positions", C.c_void_p),
        ("normals", C.c_void_p),
        ("tangents", C.c_void_p),
        ("texcoords0", C.c_void_p),
        ("indices", C.c_void_p),
        ("index_count", c_u),
        ("prim_mesh_count", c_u),
        ("prim_meshes", C.c_void_p),
        ("materials", C.c_void_p),
        ("material_count", c_u),
        ("light_count", c_u),
        ("lights", C.c_void_p),
        ("nodes", C.c_void_p),
        ("node_count", c_u),
        ("texture_count", c_u),
        ("textures", C.c_void_p),
    ]


class Shard(C.Structure):
    _fields_ = [
        ("full_width", c_u),
        ("full_height", c_u),
        ("strip_rows", c_u),
        ("shard_count", c_u),
        ("shard_index", c_u),
    ]


class AccelCheck(C.Structure):
    _fields_ = [("nodes_reached", C.c_uint64), ("triangles_referenced", C.c_uint64), ("triangles_missing", C.c_uint64),
                ("triangles_repeated", C.c_uint64), ("box_violations", C.c_uint64), ("bad_references", C.c_uint64),
                ("max_depth", c_u), ("layout", c_u), ("triangles_uncovered", C.c_uint64), ("triangles_split", C.c_uint64)]


class TraceOpts(C.Structure):
    _fields_ = [("seed", c_u), ("flags", c_u)]


class Counters(C.Structure):
    _fields_ = [
        ("rays_closest", c_u64),
        ("rays_shadow", c_u64),
        ("hits", c_u64),
        ("diffuse_hits", c_u64),
        ("tex_taps", c_u64),
        ("pixels", c_u64),
        ("nodes_visited", c_u64),
        ("tris_tested", c_u64),
        ("wave_node_steps", c_u64),
        ("wave_tri_steps", c_u64),
        ("traversal_faults", c_u64),
        ("pair_records", c_u64),
    ]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class AccelInfo(C.Structure):
    _fields_ = [
        ("triangle_count", c_u),
        ("node_count", c_u),
        ("max_depth", c_u),
        ("build_flags", c_u),
        ("sah_cost", c_f),
        ("build_ms", c_f),
        ("node_bytes", c_u64),
        ("triangle_bytes", c_u64),
        ("reference_count", c_u),
        ("reserved", c_u),
    ]


class Gbuffer(C.Structure):
    _fields_ = [("color", C.c_void_p), ("position", C.c_void_p), ("normal", C.c_void_p), ("roughMetal", C.c_void_p)]


class NrdPlanes(C.Structure):
    _fields_ = [("normalRoughness", C.c_void_p), ("viewZ", C.c_void_p), ("diffRadianceHitDist", C.c_void_p)]


class PushConstantPost(C.Structure):
    _fields_ = [("aspectRatio", c_f), ("rtMode", c_i), ("viewAccumulated", c_i), ("useGI", c_i)]


class TraceTiming(C.Structure):
    _fields_ = [("total_ms", c_f), ("traverse_ms", c_f), ("traverse_launches", c_u), ("mode", c_u), ("shade_ms", c_f), ("shade_launches", c_u)]


# layout contract
assert C.sizeof(GlobalUniforms) == 192
assert C.sizeof(PushConstantRay) == 44
assert C.sizeof(PrimMeshInfo) == 12
assert C.sizeof(GltfPBRMaterial) == 52
assert C.sizeof(GltfLight) == 32
assert C.sizeof(PrimMesh) == 20
assert C.sizeof(Node) == 68

VKRT_OK, VKRT_ERR_INVALID_ARGUMENT, VKRT_ERR_NO_DEVICE, VKRT_ERR_HIP, VKRT_ERR_OUT_OF_MEMORY, VKRT_ERR_NOT_BUILT, VKRT_ERR_UNSUPPORTED = range(7)
VKRT_BUILD_LBVH_GPU = 0x1
VKRT_BUILD_SAH_HOST = 0x2
VKRT_BUILD_PLOC_GPU = 0x4
VKRT_TRACE_SEED_INDEX_ROW_MAJOR = 0x1
VKRT_TRACE_COUNT_TRAVERSAL = 0x2
VKRT_TRACE_TIME_KERNELS = 0x4
VKRT_TRACE_SAME_SEED_EVERY_FRAME = 0x8
VKRT_ABI_VERSION = 4
# vkrt_option
VKRT_OPT_MODE, VKRT_OPT_BVH_LAYOUT, VKRT_OPT_WF_SUBFRAMES, VKRT_OPT_WF_TRAV_BLOCK = 1, 2, 3, 4
VKRT_OPT_WF_SHARE, VKRT_OPT_TRI_THRESHOLD, VKRT_OPT_WF_SHARE_PERIOD, VKRT_OPT_WF_SHARE_FLAGS = 5, 6, 7, 8
VKRT_OPT_GBUFFER_MIPS = 9
VKRT_OPT_WATERTIGHT, VKRT_OPT_SKIP_DEAD_SHADOW_RAYS, VKRT_OPT_ANYHIT_DISSOLVE = 10, 11, 12
VKRT_OPT_WF_FRAMES_IN_FLIGHT, VKRT_OPT_SPLIT_BUDGET = 13, 14
VKRT_INFO_ANYHIT_ORDER = 100  # read-only: what the build resolved the any-hit child order to
VKRT_INFO_SPLIT_BUDGET = 101  # read-only: the pre-splitting budget the build used (what -1 resolved to)

# every symbol include/vkrt.h declares (tests check the built library exports them all)
VKRT_SYMBOLS = [
    "vkrt_abi_version",
    "vkrt_last_error",
    "vkrt_device_count",
    "vkrt_scene_create",
    "vkrt_scene_destroy",
    "vkrt_scene_set_option",
    "vkrt_scene_get_option",
    "vkrt_reserve",
    "vkrt_reserve_frames",
    "vkrt_accel_build",
    "vkrt_accel_get_info",
    "vkrt_shard_rows",
    "vkrt_pathtrace",
    "vkrt_pathtrace_frames",
    "vkrt_gbuffer_raycast",
    "vkrt_hybrid_trace",
    "vkrt_gbuffer_raycast_nrd",
    "vkrt_hybrid_trace_nrd",
    "vkrt_post",
    "vkrt_counters_reset",
    "vkrt_counters_read",
    "vkrt_last_trace_ms",
    "vkrt_last_trace_timing",
    "vkrt_debug_check_accel",
    "vkrt_debug_trace_rays",
    "vkrt_debug_eval_math",
]


def declare_vkrt(lib):
    """Attach argtypes/restype for the C ABI to a loaded libvkrt.so."""
    P = C.POINTER
    lib.vkrt_abi_version.restype = C.c_int
    lib.vkrt_last_error.restype = C.c_char_p
    lib.vkrt_device_count.restype = C.c_int
    lib.vkrt_scene_create.argtypes = [P(SceneDesc), C.c_int, P(C.c_void_p)]
    lib.vkrt_scene_create.restype = C.c_int
    lib.vkrt_scene_destroy.argtypes = [C.c_void_p]
    lib.vkrt_scene_destroy.restype = None
    lib.vkrt_scene_set_option.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.vkrt_scene_set_option.restype = C.c_int
    lib.vkrt_scene_get_option.argtypes = [C.c_void_p, C.c_int, P(C.c_int)]
    lib.vkrt_scene_get_option.restype = C.c_int
    lib.vkrt_reserve.argtypes = [C.c_void_p, P(Shard), C.c_void_p]
    lib.vkrt_reserve.restype = C.c_int
    lib.vkrt_reserve_frames.argtypes = [C.c_void_p, P(Shard), C.c_uint32, C.c_void_p]
    lib.vkrt_reserve_frames.restype = C.c_int
    lib.vkrt_accel_build.argtypes = [C.c_void_p, c_u, C.c_void_p]
    lib.vkrt_accel_build.restype = C.c_int
    lib.vkrt_accel_get_info.argtypes = [C.c_void_p, P(AccelInfo)]
    lib.vkrt_accel_get_info.restype = C.c_int
    lib.vkrt_debug_check_accel.argtypes = [C.c_void_p, P(AccelCheck)]
    lib.vkrt_debug_check_accel.restype = C.c_int
    lib.vkrt_shard_rows.argtypes = [P(Shard)]
    lib.vkrt_shard_rows.restype = c_u
    lib.vkrt_pathtrace.argtypes = [
        C.c_void_p, P(PushConstantRay), P(GlobalUniforms), P(TraceOpts), P(Shard), C.c_void_p, C.c_void_p,
    ]
    lib.vkrt_pathtrace.restype = C.c_int
    lib.vkrt_pathtrace_frames.argtypes = [
        C.c_void_p, P(PushConstantRay), P(GlobalUniforms), P(TraceOpts), P(Shard), C.c_void_p, c_u, C.c_void_p,
    ]
    lib.vkrt_pathtrace_frames.restype = C.c_int
    lib.vkrt_gbuffer_raycast.argtypes = [C.c_void_p, P(c_f * 4), C.c_int, P(GlobalUniforms), P(Shard), P(Gbuffer), C.c_void_p]
    lib.vkrt_gbuffer_raycast.restype = C.c_int
    lib.vkrt_hybrid_trace.argtypes = [C.c_void_p, P(PushConstantRay), P(GlobalUniforms), P(TraceOpts), P(Shard), P(Gbuffer), C.c_void_p, C.c_void_p]
    lib.vkrt_hybrid_trace.restype = C.c_int
    lib.vkrt_gbuffer_raycast_nrd.argtypes = [C.c_void_p, P(c_f * 4), C.c_int, P(GlobalUniforms), P(c_f * 16), P(Shard), P(Gbuffer), P(NrdPlanes), C.c_void_p]
    lib.vkrt_gbuffer_raycast_nrd.restype = C.c_int
    lib.vkrt_hybrid_trace_nrd.argtypes = [C.c_void_p, P(PushConstantRay), P(GlobalUniforms), P(TraceOpts), P(Shard), P(Gbuffer), P(NrdPlanes), C.c_void_p,
                                          C.c_void_p]
    lib.vkrt_hybrid_trace_nrd.restype = C.c_int
    lib.vkrt_post.argtypes = [C.c_int, P(PushConstantPost), c_u, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.vkrt_post.restype = C.c_int
    lib.vkrt_counters_reset.argtypes = [C.c_void_p, C.c_void_p]
    lib.vkrt_counters_reset.restype = C.c_int
    lib.vkrt_counters_read.argtypes = [C.c_void_p, P(Counters)]
    lib.vkrt_counters_read.restype = C.c_int
    lib.vkrt_last_trace_ms.argtypes = [C.c_void_p, P(c_f)]
    lib.vkrt_last_trace_ms.restype = C.c_int
    lib.vkrt_last_trace_timing.argtypes = [C.c_void_p, P(TraceTiming)]
    lib.vkrt_last_trace_timing.restype = C.c_int
    lib.vkrt_debug_trace_rays.argtypes = [
        C.c_void_p, c_u, C.c_void_p, C.c_void_p, c_f, c_f, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
    ]
    lib.vkrt_debug_trace_rays.restype = C.c_int
    lib.vkrt_debug_eval_math.argtypes = [C.c_int, C.c_int, c_u, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.vkrt_debug_eval_math.restype = C.c_int
    return lib
