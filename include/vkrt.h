/*
 * vkrt.h -- C ABI of the MI355X-native ray-tracing path.
 *
 * This is the drop-in boundary for the path-tracing path of vk-raytracing-engine.
 * The reference has no plugin ABI; its ray-tracing path is entered through member
 * functions of `HelloVulkan` plus the data contract of shaders/host_device.h.  Each
 * entry point below names the reference interface it replaces (file:line relative to
 * the reference tree).  Plain pointers and sizes only; no C++ or torch types.
 *
 *   reference call (main.cpp)                      this ABI
 *   ---------------------------------------------  -------------------------------
 *   loadGltfScene()        hello_vulkan.cpp:327     vkrt_scene_create (flat arrays)
 *   createBottomLevelASGltf()  :1001                vkrt_accel_build
 *   createTopLevelAsGltf()     :1031                vkrt_accel_build (same call)
 *   updateUniformBuffer()      :61                  GlobalUniforms* argument
 *   pathtrace()                :1423                vkrt_pathtrace
 *   the frame loop at rest     main.cpp:503-508     vkrt_pathtrace_frames (n progressive frames per call)
 *   resetFrame()/updateFrame() :1501-1521           caller-owned PushConstantRay.frame
 *   destroyResources()         :518                 vkrt_scene_destroy
 *
 * Ownership: every input array is copied at vkrt_scene_create (the caller may free
 * it on return, like the reference's staging upload, hello_vulkan.cpp:353-357).  The
 * radiance image is caller-owned device memory (rgba32f, tightly packed rows); it is
 * read-modify-written when PushConstantRay.frame > 0 (raytrace.rgen:136-141).
 * Errors: int return codes, 0 = VKRT_OK; text via vkrt_last_error(); nothing throws
 * across the ABI.  Threading: one host thread per scene handle at a time; work is
 * enqueued on the caller's HIP stream (hipStream_t passed as void*; NULL = default).
 * The library never falls back to a CPU path: without a HIP device every compute
 * entry point returns VKRT_ERR_NO_DEVICE.
 */
#ifndef VKRT_H
#define VKRT_H

#include <stdint.h>
#include "vkrt_host_device.h"

#ifdef __cplusplus
extern "C" {
#endif

#define VKRT_ABI_VERSION 4 /* 4: vkrt_reserve_frames; vkrt_counters.pair_records, vkrt_trace_timing.shade_ms / shade_launches (appended).
                              2: vkrt_scene_set_option / vkrt_reserve, vkrt_counters.traversal_faults.
                              3: vkrt_pathtrace_frames, VKRT_TRACE_SAME_SEED_EVERY_FRAME, options 10-14, vkrt_accel_info.reference_count,
                                 vkrt_accel_check.triangles_uncovered, VKRT_INFO_ANYHIT_ORDER (the options and the query
                                 that round 3 had added under version 2 are part of 3: a client that needs them asks for >= 3) */

enum vkrt_status {
  VKRT_OK = 0,
  VKRT_ERR_INVALID_ARGUMENT = 1,
  VKRT_ERR_NO_DEVICE = 2,
  VKRT_ERR_HIP = 3,
  VKRT_ERR_OUT_OF_MEMORY = 4,
  VKRT_ERR_NOT_BUILT = 5,     /* vkrt_pathtrace before vkrt_accel_build */
  VKRT_ERR_UNSUPPORTED = 6
};

typedef struct vkrt_scene vkrt_scene; /* opaque; one per GPU */

/* One glTF primitive-mesh, as nvh::GltfPrimMesh is consumed by the reference
 * (hello_vulkan.cpp:363-368 for the shader lookup, :955-987 for the BLAS ranges:
 * primitiveCount = indexCount/3, firstVertex = vertexOffset, maxVertex = vertexCount). */
typedef struct vkrt_prim_mesh {
  uint32_t firstIndex;    /* into indices[] */
  uint32_t indexCount;    /* multiple of 3 */
  uint32_t vertexOffset;  /* added to every index value */
  uint32_t vertexCount;
  int32_t  materialIndex; /* may be -1; the shader clamps with max(0, .) (raytrace.rchit:38) */
} vkrt_prim_mesh;

/* One drawable node = one TLAS instance (hello_vulkan.cpp:1035-1043):
 * transform = worldMatrix, instanceCustomIndex = primMesh, mask 0xFF, cull disabled. */
typedef struct vkrt_node {
  float   worldMatrix[16]; /* column-major object->world */
  int32_t primMesh;
} vkrt_node;

/* One sampled image: 8-bit RGBA, row-major, top row first (what tinygltf/stb hand the
 * reference, hello_vulkan.cpp:482-499).  is_srgb follows getImageFormat
 * (hello_vulkan.cpp:417-443).  Sampler (:448-454 and SURVEY Appendix A 27-29): linear, REPEAT;
 * the ray tracing stages read LOD 0 (no derivatives there), the hybrid mode's G-buffer pass samples
 * like the fragment shader it stands for -- implicit LOD over the mip chain the library generates at
 * vkrt_scene_create (:499), anisotropy 4.  textures[i] here is glTF texture i (already resolved to
 * its source image, :505-509). */
typedef struct vkrt_texture {
  uint32_t       width;
  uint32_t       height;
  const uint8_t* rgba8;
  int32_t        is_srgb;
} vkrt_texture;

/* Flat scene arrays exactly as the reference uploads them (hello_vulkan.cpp:353-379):
 * SoA vertex attributes shared by all primitive-meshes, one u32 index buffer. */
typedef struct vkrt_scene_desc {
  uint32_t struct_size;        /* = sizeof(vkrt_scene_desc), ABI check */
  uint32_t vertex_count;
  const float*    positions;   /* vec3[vertex_count]  (m_gltfScene.m_positions)  */
  const float*    normals;     /* vec3[vertex_count]  (m_normals)                */
  const float*    tangents;    /* vec4[vertex_count]  (m_tangents, w=handedness) */
  const float*    texcoords0;  /* vec2[vertex_count]  (m_texcoords0)             */
  const uint32_t* indices;     /* u32[index_count]    (m_indices)                */
  uint32_t index_count;
  uint32_t prim_mesh_count;
  const vkrt_prim_mesh*  prim_meshes;
  const GltfPBRMaterial* materials;
  uint32_t material_count;
  uint32_t light_count;
  const GltfLight*       lights;
  const vkrt_node*       nodes;
  uint32_t node_count;
  uint32_t texture_count;      /* 0 => a 1x1 white dummy is bound (hello_vulkan.cpp:468-472) */
  const vkrt_texture*    textures;
} vkrt_scene_desc;

/* Acceleration-structure build selection (replaces
 * VK_BUILD_ACCELERATION_STRUCTURE_PREFER_FAST_TRACE_BIT_KHR, hello_vulkan.cpp:1010,1046). */
enum vkrt_build_flags {
  VKRT_BUILD_LBVH_GPU = 0x1,  /* Morton-code radix tree built by HIP kernels on the device (fastest build)  */
  VKRT_BUILD_SAH_HOST = 0x2,  /* binned-SAH BVH built by the C++ host, then uploaded (best tree, ~15x the build time)  */
  VKRT_BUILD_PLOC_GPU = 0x4,  /* device build for trace speed: locally-ordered clustering over the Morton order (ploc.hip), upper
                                 levels re-built with a full-sweep SAH over the clustered subtrees (lbvh.hip)  */
  VKRT_BUILD_DEFAULT  = 0x4   /* the reference builds on the device with PREFER_FAST_TRACE: so does the default  */
};

/* Image-space sharding (replaces the single vkCmdTraceRaysKHR(W,H,1) grid,
 * hello_vulkan.cpp:1446).  The full launch size stays gl_LaunchSizeEXT for every
 * shard so seeds and camera rays depend on global pixel coordinates only.
 * Rows are dealt in strips: strip s = rows [s*strip_rows, (s+1)*strip_rows) belongs to
 * shard (s % shard_count).  The shard's output buffer holds its strips stacked in
 * increasing s, full_width pixels per row.  strip_rows = 0 means "the whole image". */
typedef struct vkrt_shard {
  uint32_t full_width;
  uint32_t full_height;
  uint32_t strip_rows;
  uint32_t shard_count;
  uint32_t shard_index;
} vkrt_shard;

enum vkrt_trace_flags {
  /* Default reproduces raytrace.rgen:27: seed index = y*x + x.  This flag selects the
   * collision-free y*full_width + x instead (not parity). */
  VKRT_TRACE_SEED_INDEX_ROW_MAJOR = 0x1,
  /* Instrumented launch (slower): also count BVH nodes visited / triangles tested / wavefront steps and, in the default
   * wavefront pipeline, the hit / diffuse-lobe / texture-tap tallies of the shading stage.  Rays and pixels are always counted. */
  VKRT_TRACE_COUNT_TRAVERSAL = 0x2,
  /* Record HIP events around every traversal-kernel launch of the frame (vkrt_last_trace_timing). */
  VKRT_TRACE_TIME_KERNELS = 0x4,
  /* vkrt_pathtrace_frames: every frame of the call uses opts->seed instead of opts->seed + i (a host that does not advance its seed) */
  VKRT_TRACE_SAME_SEED_EVERY_FRAME = 0x8
};

typedef struct vkrt_trace_opts {
  uint32_t seed;   /* replaces int(clockARB()) in raytrace.rgen:27 */
  uint32_t flags;  /* vkrt_trace_flags */
} vkrt_trace_opts;

/* Totals accumulated by device atomics since the last vkrt_counters_reset. */
typedef struct vkrt_counters {
  uint64_t rays_closest;   /* traceRayEXT calls of raytrace.rgen:64-75          */
  uint64_t rays_shadow;    /* traceRayEXT calls of raytrace.rgen:85-97          */
  uint64_t hits;           /* raytrace.rchit invocations                 (wavefront pipeline: with COUNT_TRAVERSAL) */
  uint64_t diffuse_hits;   /* rchit invocations that took the diffuse lobe           (same)                    */
  uint64_t tex_taps;       /* texture() calls                                         (same)                    */
  uint64_t pixels;         /* rgen invocations                                  */
  uint64_t nodes_visited;  /* only with VKRT_TRACE_COUNT_TRAVERSAL              */
  uint64_t tris_tested;    /* only with VKRT_TRACE_COUNT_TRAVERSAL              */
  uint64_t wave_node_steps; /* COUNT_TRAVERSAL, wide8 layout: node steps per wavefront (one per 64 lanes);
                              nodes_visited / (64 * wave_node_steps) = lane efficiency of the node phase */
  uint64_t wave_tri_steps;  /* same for the triangle phase                        */
  uint64_t traversal_faults; /* always counted: stack pushes dropped (stack sized from the builder's depth) + walks cut by the step
                               bound.  Non-zero means a builder / traversal mismatch and possibly wrong pixels; tests assert 0. */
  uint64_t pair_records;   /* ABI 4, wavefront pipeline: path records that carried TWO rays through a round (the shadow ray of a segment and
                              the closest-hit ray of the next one): records moved = rays_closest + rays_shadow - pair_records */
} vkrt_counters;

typedef struct vkrt_accel_info {
  uint32_t triangle_count;   /* instanced (flattened) triangles */
  uint32_t node_count;       /* BVH nodes in the traversal layout */
  uint32_t max_depth;
  uint32_t build_flags;      /* which builder produced it */
  float    sah_cost;         /* SAH cost of the tree, traversal 1 / intersect 1 */
  float    build_ms;         /* wall time of the last build */
  uint64_t node_bytes;
  uint64_t triangle_bytes;
  uint32_t reference_count;  /* triangle slots of the tree = leaves' references; > triangle_count when VKRT_OPT_SPLIT_BUDGET split triangles */
  uint32_t reserved;
} vkrt_accel_info;

/* ---- library ---------------------------------------------------------------------- */
int         vkrt_abi_version(void);
const char* vkrt_last_error(void);       /* thread-local, never NULL */
int         vkrt_device_count(void);     /* 0 without a HIP device */

/* ---- scene (replaces loadGltfScene's uploads, hello_vulkan.cpp:353-381) ----------- */
int  vkrt_scene_create(const vkrt_scene_desc* desc, int device, vkrt_scene** out);
void vkrt_scene_destroy(vkrt_scene* scene);

/* ---- per-scene execution options ------------------------------------------------------
 * Scheduling knobs of the HIP path; none of them changes a pixel (tests/test_gpu_parity.py hashes every
 * combination).  They are fields of the scene handle, not process globals.  Their INITIAL values are read once, at
 * vkrt_scene_create, from the environment variables named below -- process-wide test hooks kept for A/B runs of
 * unmodified binaries (bench.py, vkrt_render); vkrt_scene_set_option overrides them per handle.
 * Options marked [build] are consumed by the next vkrt_accel_build, the others by the next vkrt_pathtrace. */
enum vkrt_option {
  VKRT_OPT_MODE            = 1, /* 1 = wavefront pipeline (default), 0 = one persistent megakernel (BVH2 only) [build]; env VKRT_MODE=mega */
  VKRT_OPT_BVH_LAYOUT      = 2, /* 1 = 8-wide compressed nodes (default with the wavefront pipeline), 0 = BVH2 [build]; env VKRT_BVH=bvh2 */
  VKRT_OPT_WF_SUBFRAMES    = 3, /* independent sub-frames of a launch on internal streams, 1..8 (default 3); env VKRT_WF_SUBFRAMES */
  VKRT_OPT_WF_TRAV_BLOCK   = 4, /* threads per traversal workgroup: 64 (default), 128, 256; env VKRT_WF_TRAV_BLOCK */
  VKRT_OPT_WF_SHARE        = 5, /* idle lanes of a traversal wave needed before they adopt subtrees, 0 = off (default 16) [build]; env VKRT_WF_SHARE */
  VKRT_OPT_TRI_THRESHOLD   = 6, /* lanes with pending triangles, per 64 lanes still walking, before a wave tests them; 1 = every step, 0 = test at once,
                                   no parking (default 32) [build]; env VKRT_TRI_THRESHOLD */
  VKRT_OPT_WF_SHARE_PERIOD = 7, /* sharing attempted on steps with (step & mask) == mask (default 0 = every step) [build]; env VKRT_WF_SHARE_PERIOD */
  VKRT_OPT_WF_SHARE_FLAGS  = 8, /* bit 0: lanes with an empty stack also donate a pending child of their current group.  Bits 1-3: child order of
                                   any-hit (shadow / AO) walks -- "is anything in the way" has the same answer in any order, so this is a cost
                                   heuristic only: bit 1 = always the FARTHEST pending child first; bit 2 = farthest first for rays that end outside
                                   the bounds of the scene (a shadow ray towards a light outside the building is stopped by the building's shell,
                                   the last thing a front-to-back walk reaches), front to back otherwise; bit 3 = automatic: like bit 2 unless the
                                   scene has room-sized triangles, which a front-to-back walk meets at once.  Bit 4 (round 4): lanes that have no node
                                   work to give but hold two or more pending triangles (a ray grazing a plane of thin strips leaves a node test
                                   with up to 24) give half of them to an idle lane.  Default 25 = bits 0, 3 and 4 [build];
                                   env VKRT_WF_SHARE_FLAGS */
  VKRT_OPT_GBUFFER_MIPS    = 9, /* NOT a scheduling knob: 1 (default) = vkrt_gbuffer_raycast samples textures like the fragment shader it replaces
                                   (implicit LOD over the mip chain, anisotropy 4; hello_vulkan.cpp:448-454, :499), 0 = LOD 0; env VKRT_GBUFFER_MIPS */
  VKRT_OPT_WATERTIGHT      = 10, /* NOT a scheduling knob [build]: 0 (default) = Moeller-Trumbore on pre-subtracted (v0, e1, e2) records in binary32
                                   (what BASELINE.json's north star names); 1 = the watertight ray/triangle test of Woop, Benthin, Wald 2013 on the exact
                                   vertices (p0, p1, p2): edge functions in ray space with a double-precision fallback on exact zeros, so a ray through a
                                   shared edge or vertex hits one of the triangles -- what the Vulkan specification demands of traceRayEXT
                                   (raytrace.rgen:64-75).  Hits differ from the default in the last bits of (t, u, v), and in the rare pixels where the
                                   default leaks through an edge; the oracle implements both (orc_set_watertight).  env VKRT_WATERTIGHT */
  VKRT_OPT_SKIP_DEAD_SHADOW_RAYS = 11, /* 0 (default) = every diffuse hit traces its shadow ray, as raytrace.rgen:79-97 does; 1 = a diffuse hit whose
                                   contribution min(prd.hitValue * curWeight, 10) is exactly zero (light behind the surface, no emission) traces none:
                                   raytrace.rgen:99-102 adds that zero whether or not the ray is occluded, so every pixel is bit-identical, only
                                   vkrt_counters.rays_shadow drops.  Path-tracing mode of the wavefront pipeline only.  env VKRT_SKIP_DEAD_SHADOW_RAYS */
  VKRT_OPT_ANYHIT_DISSOLVE = 12, /* [build] the any-hit alpha / dissolve stage of raytrace_rahit_todo.glsl:23-37 (hello_vulkan.cpp:1185-1191 keeps its
                                   registration commented out; every ray of the reference is gl_RayFlagsOpaqueEXT): 0 (default) = all geometry opaque,
                                   as the reference runs; 1 = a candidate hit on a triangle whose material has dissolve < 1 is ignored when dissolve == 0
                                   and otherwise with probability 1 - dissolve, for closest-hit and shadow rays of both modes (not for the ray-cast
                                   G-buffer: a raster pass has no any-hit stage).  The shader is written against the dead OBJ pipeline's WaveFrontMaterial;
                                   on the glTF material the live pipeline has, dissolve = pbrBaseColorFactor.a and "illum == 4" = dissolve < 1.  The GLSL
                                   draws rnd(prd.seed) per invocation, but Vulkan defines neither the order nor the number of any-hit invocations, so the
                                   decision here is a pure function of the ray and the triangle: rnd(tea(triangle id, prd.seed when the ray is traced)) >
                                   dissolve; prd.seed itself is not advanced.  The result stays a property of the triangle set (any tree, any schedule);
                                   the oracle implements the same rule (orc_set_dissolve).  env VKRT_ANYHIT_DISSOLVE */
  VKRT_OPT_WF_FRAMES_IN_FLIGHT = 13, /* vkrt_pathtrace_frames: consecutive frames of a call rendered at the same time, 1..8 (default 3), each on
                                   record streams of its own (544 B per pixel and frame in flight); their pixel values meet in the ordered blend at
                                   the end of a frame, so the image is bit-identical for every value.  A call with frames in flight does not split
                                   its frames into sub-frames (option 3): few, large launches overlap best.  env VKRT_WF_FRAMES_IN_FLIGHT */
  VKRT_OPT_SPLIT_BUDGET    = 14, /* [build] NOT a pixel-changing knob: triangle pre-splitting in the device builders (VKRT_BUILD_PLOC_GPU /
                                   VKRT_BUILD_LBVH_GPU), the part of PREFER_FAST_TRACE (hello_vulkan.cpp:1010, :1046) that matters on artist-made
                                   geometry.  Value = budget of EXTRA triangle references in percent of the triangle count, 0..100 (0 = off), or -1 = automatic (the default since ABI 4): triangles
                                   that are large against the scene grid enter the tree as several references, each with the box of one piece of
                                   the triangle.  Only references multiply (a copy of the 48-byte record per reference): the hit test, the triangle id
                                   of the tie rule and every pixel are unchanged.  vkrt_accel_info.reference_count reports the result.
                                   WHEN TO SET IT (measured, profiles/r05_split_rotated.jsonl): 10-30 for scenes whose large triangles are
                                   not aligned with the coordinate axes -- a building rotated 45 degrees about y traces +14 % faster, one
                                   rotated 35 / 20 degrees about y / x +97 % (the box of a room-sized diagonal triangle is mostly empty:
                                   61 -> 30 triangles tested per ray); 0 for axis-aligned architecture and for finely tessellated meshes,
                                   where the extra references cost 1-8 %.  -1 (ABI 4) = automatic: the device builders build with a 30 %
                                   budget and without, and keep the split tree only when its SAH cost is below 0.9 of the unsplit one
                                   (rotated buildings: 0.70-0.82; axis-aligned and finely tessellated scenes: 0.95-1.08) -- two or three
                                   builds of ~13 ms instead of one; VKRT_INFO_SPLIT_BUDGET reports the outcome.  env VKRT_SPLIT_BUDGET */
  VKRT_OPT_LAST            = 14,
  VKRT_INFO_ANYHIT_ORDER   = 100, /* read-only (vkrt_scene_get_option; set is refused): the child-order bits (2 | 4) that the last vkrt_accel_build
                                   resolved VKRT_OPT_WF_SHARE_FLAGS to, i.e. what bit 3 ("automatic") decided for this scene; 0 before a build */
  VKRT_INFO_SPLIT_BUDGET   = 101  /* read-only (ABI 4): the pre-splitting budget the last vkrt_accel_build used -- what VKRT_OPT_SPLIT_BUDGET = -1
                                   ("automatic") resolved to: 30 or 0 */
};
int vkrt_scene_set_option(vkrt_scene* scene, int option, int value);
int vkrt_scene_get_option(const vkrt_scene* scene, int option, int* value);

/* ---- acceleration structure (replaces createBottomLevelASGltf :1001-1011 and
 *      createTopLevelAsGltf :1031-1047) ------------------------------------------- */
int vkrt_accel_build(vkrt_scene* scene, uint32_t build_flags, void* hip_stream);
int vkrt_accel_get_info(const vkrt_scene* scene, vkrt_accel_info* out);

/* ---- path trace (replaces HelloVulkan::pathtrace :1423-1448 = one
 *      vkCmdTraceRaysKHR over raytrace.rgen/.rchit/.rmiss/raytraceShadow.rmiss) ---- */
uint32_t vkrt_shard_rows(const vkrt_shard* shard); /* rows of the shard's buffer */
/* Sizes the per-scene working set (path-record streams of the wavefront pipeline, internal streams and events) for
 * launches of this shard geometry, like the reference allocates its offscreen images at start-up and on resize
 * (createOffscreenRender, hello_vulkan.cpp:637-665).  After vkrt_reserve a vkrt_pathtrace of the same or a smaller
 * shard never allocates and never synchronises with the host.  Without it the first launch (and any launch larger
 * than every earlier one) grows the working set lazily: one hipStreamSynchronize + hipMalloc inside that call. */
int vkrt_reserve(vkrt_scene* scene, const vkrt_shard* shard, void* hip_stream);
/* The same for a caller that knows how many frames it hands to one vkrt_pathtrace_frames call (ABI 4).  The working set holds one
 * set of path-record streams (544 B per pixel of the shard) per frame the library keeps in flight inside a call, plus a 16-B staging
 * plane per pixel and frame in flight: frames_per_call = 1 (a client of vkrt_pathtrace only) sizes it for ONE set -- 1.1 GB at
 * 1920x1080, 4.5 GB at 3840x2160 --, frames_per_call >= VKRT_OPT_WF_FRAMES_IN_FLIGHT (default 3) for that many: 3.5 GB / 14 GB.
 * vkrt_reserve is vkrt_reserve_frames with frames_per_call = VKRT_OPT_WF_FRAMES_IN_FLIGHT, i.e. the larger figure whatever the
 * client goes on to call.  The working set only grows (a later, larger call or reservation re-allocates it once); it is released by
 * vkrt_scene_destroy.  Texture memory, for the record: vkrt_scene_create keeps, beside the RGBA8 pool with its mip chain (4/3 of the
 * texel bytes), a footprint pool of 16 B per level-0 texel -- 4x the level-0 texel bytes -- so that a bilinear tap of the path tracer
 * is one load; it is built while it stays under 2 GiB (above that, and with the test hook VKRT_TEX_QUADS=0 in the environment, taps
 * gather their four texels from the RGBA8 pool: same values, ~9 % slower hit shading). */
int vkrt_reserve_frames(vkrt_scene* scene, const vkrt_shard* shard, uint32_t frames_per_call, void* hip_stream);
/* Asynchronous like the command-buffer recording it replaces: returns after enqueueing.  All work is ordered after what
 * was enqueued on `hip_stream` before the call and complete before anything enqueued on it afterwards (the library may
 * run parts of a frame on internal streams that fork from and join `hip_stream` through events).  No host
 * synchronisation happens inside the call once the working set is large enough (vkrt_reserve above).  pc->frame > 0 blends into the image the caller kept from the previous
 * frame (raytrace.rgen:136-145), so the image buffer is caller-owned and persistent.  Calls on one scene handle must
 * be serialised by the caller.  A shard without rows (more shards than strips) is a no-op and may pass NULL buffers. */
int vkrt_pathtrace(vkrt_scene* scene, const PushConstantRay* pc, const GlobalUniforms* cam,
                   const vkrt_trace_opts* opts, const vkrt_shard* shard,
                   float* rgba32f_device, void* hip_stream);

/* n_frames progressive frames of an unchanged camera in one call: the reference's render loop with the camera at rest
 * (main.cpp:503-508: updateFrame() -> pathtrace(), frame after frame; hello_vulkan.cpp:1501-1521 advances pcRay.frame, raytrace.rgen:136-145
 * blends frame f into the image with weight 1 / (f + 1)).  Frame i of the call (0 <= i < n_frames) is traced with pc->frame + i and
 * opts->seed + i (opts->seed with VKRT_TRACE_SAME_SEED_EVERY_FRAME); the image afterwards is bit for bit what n_frames vkrt_pathtrace calls
 * with those values would have left.  Inside the call consecutive frames run at the same time (VKRT_OPT_WF_FRAMES_IN_FLIGHT): the
 * launches of one frame fill the tails of the other's, which a sequence of single-frame calls -- each complete before the next
 * begins -- cannot do.  Same asynchrony, ordering and
 * shard rules as vkrt_pathtrace (which is this call with n_frames = 1); vkrt_counters and vkrt_last_trace_ms cover the whole call.
 * Working set: vkrt_reserve_frames(scene, shard, n_frames, stream) sizes it for such calls (vkrt_reserve: for whatever the current
 * options keep in flight). */
int vkrt_pathtrace_frames(vkrt_scene* scene, const PushConstantRay* pc, const GlobalUniforms* cam,
                          const vkrt_trace_opts* opts, const vkrt_shard* shard,
                          float* rgba32f_device, uint32_t n_frames, void* hip_stream);

/* ---- hybrid mode (reference rtMode == 0; SURVEY.md 8f row 1, BASELINE config 5) --------------------- */
/* The four raster planes that raytraceHybrid.rgen reads (RtxBindings 1,3,4,6; host_device.h:51-63,
 * attachments hello_vulkan.cpp:690-734).  Caller-owned device memory, rows of the shard stacked like the
 * path-trace image. */
typedef struct vkrt_gbuffer {
  float* color;      /* rgba32f eOutImage : rgb = emission + un-shadowed direct light (frag_shader.frag:190-214), a = albedo.r */
  float* position;   /* rgba32f ePosMap   : xyz = world position, w = albedo.g; cleared to (0,0,0,1)                    */
  float* normal;     /* rgba32f eNormMap  : xyz = shading normal, w = albedo.b; cleared to (0,0,0,1)                     */
  float* roughMetal; /* 2 floats/pixel eRoughMap: roughness, metalness after the rg16f round trip                        */
} vkrt_gbuffer;
/* Replaces the raster pass HelloVulkan::rasterizeGltf (hello_vulkan.cpp:583-615, vert_shader.vert,
 * frag_shader.frag) by a primary ray cast per pixel centre evaluating the same shader math.  texture() takes
 * its LOD as in a fragment shader: dFdx / dFdy of the texture coordinate inside the pixel's 2x2 quad (the quad
 * partner evaluated on the same triangle's plane), then the Vulkan scale-factor / LOD / anisotropy formulas
 * with the reference's sampler (trilinear, maxAnisotropy 4); VKRT_OPT_GBUFFER_MIPS = 0 reads LOD 0 instead.
 * lightsCount = PushConstantRaster.lightsCount. */
int vkrt_gbuffer_raycast(vkrt_scene* scene, const float clearColor[4], int lightsCount, const GlobalUniforms* cam,
                         const vkrt_shard* shard, const vkrt_gbuffer* out, void* hip_stream);
/* Replaces HelloVulkan::raytraceRasterizedScene (hello_vulkan.cpp:1450-1473 = vkCmdTraceRaysKHR over
 * raytraceHybrid.rgen): shadow / AO / GI per PushConstantRay.useShadows/useAO/useGI, accumulated into
 * accum_rgba32f (eAccumMap; rgb = indirect light, a = visibility * (1 - ao)). */
int vkrt_hybrid_trace(vkrt_scene* scene, const PushConstantRay* pc, const GlobalUniforms* cam, const vkrt_trace_opts* opts,
                      const vkrt_shard* shard, const vkrt_gbuffer* gbuffer, float* accum_rgba32f_device, void* hip_stream);
/* ---- NRD / REBLUR front-end planes ----------------------------------------------------------
 * The raster pass also writes the denoiser's inputs (frag_shader.frag:133-136, attachments hello_vulkan.cpp:690-741) and
 * raytraceHybrid.rgen:273-281 packs the GI radiance + normalised hit distance for REBLUR (gltf.glsl:156-273, hitDistParams
 * (3, 1, 20, -25)).  The reference's NRD.Denoise call is commented out (main.cpp:566-602), so these planes have no consumer
 * there; they are offered as optional outputs so a host that enables NRD finds its inputs.  All three are plain float planes
 * holding the values the Vulkan attachments would store (rgb10_a2 UNORM / r16f / rgba16f quantisation applied). */
typedef struct vkrt_nrd_planes {
  float* normalRoughness;     /* rgba32f eInNormRough: oct-encoded normal .xy, roughness, clamp(materialId / 3, 0, 1); cleared to 0   */
  float* viewZ;               /* 1 float/pixel eInViewZ: (pcRaster.viewMatrix * worldPos).z; cleared to 0                             */
  float* diffRadianceHitDist; /* rgba32f eInRadHitD: YCoCg radiance of the GI path + normalised hit distance; 0 where GI did not run */
} vkrt_nrd_planes;
/* vkrt_gbuffer_raycast that also fills nrd->normalRoughness / viewZ and clears nrd->diffRadianceHitDist.
 * view_matrix = PushConstantRaster.viewMatrix (hello_vulkan.cpp:600), column-major 4x4. */
int vkrt_gbuffer_raycast_nrd(vkrt_scene* scene, const float clearColor[4], int lightsCount, const GlobalUniforms* cam, const float view_matrix[16],
                             const vkrt_shard* shard, const vkrt_gbuffer* out, const vkrt_nrd_planes* nrd, void* hip_stream);
/* vkrt_hybrid_trace that also writes nrd->diffRadianceHitDist where PushConstantRay.useGI ran (reads nrd->viewZ). */
int vkrt_hybrid_trace_nrd(vkrt_scene* scene, const PushConstantRay* pc, const GlobalUniforms* cam, const vkrt_trace_opts* opts,
                          const vkrt_shard* shard, const vkrt_gbuffer* gbuffer, const vkrt_nrd_planes* nrd, float* accum_rgba32f_device,
                          void* hip_stream);
/* Replaces drawPost's fragment stage (post.frag:36-58): hybrid composite main.rgb * rt.a + rt.rgb (rtMode 0) or
 * pass-through (rtMode 1), then gamma 1/2.2 on all four channels.  n_pixels rgba32f device buffers.
 * NaN texels are part of the result, as in the reference: post.frag:57 applies pow(x, 1/2.2) to whatever the composite holds
 * and pow of a negative base is undefined in GLSL (NaN here, on the oracle and on the GPUs the reference targets).  The
 * hybrid accumulation plane does go negative -- raytrace.rchit's specular branch returns negative weights when
 * dot(N, L) < 0 and raytraceHybrid.rgen adds the GI radiance unclamped -- so a hybrid frame shows them:
 * 898 of the 46,080 texels sampled by the BASELINE config-5 test (1920x1080 atrium, shadows + AO + GI depth 8, two frames),
 * at the same texels on both sides (tests/test_gpu_configs.py asserts the coincidence).  The path-tracing mode's image
 * (rtMode 1) is not affected in practice: raytrace.rgen clamps every contribution with min(., 10) but sums can still be
 * negative; callers that display the plane should treat NaN as black, like a UNORM swapchain write does. */
int vkrt_post(int device, const PushConstantPost* pc, uint32_t n_pixels, const float* main_rgba32f, const float* rt_rgba32f,
              float* out_rgba32f, void* hip_stream);

/* ---- counters / timing ------------------------------------------------------------ */
int vkrt_counters_reset(vkrt_scene* scene, void* hip_stream);
int vkrt_counters_read(vkrt_scene* scene, vkrt_counters* out); /* synchronises the device */
/* Device time of the most recent vkrt_pathtrace kernel on this scene in milliseconds,
 * from HIP events recorded on the launch stream (synchronises on the stop event). */
int vkrt_last_trace_ms(vkrt_scene* scene, float* ms);
/* Breakdown of the most recent vkrt_pathtrace: whole frame, and (with VKRT_TRACE_TIME_KERNELS, or always
 * in megakernel mode) the summed duration and number of launches of the traversal kernel -- the dominant
 * kernel of the path (k_wf_traverse / k_pathtrace). */
typedef struct vkrt_trace_timing {
  float    total_ms;
  float    traverse_ms;
  uint32_t traverse_launches;
  uint32_t mode;            /* 1 wavefront pipeline, 0 megakernel (VKRT_MODE=mega) */
  float    shade_ms;        /* ABI 4, with VKRT_TRACE_TIME_KERNELS: summed time between the end of a traversal launch and the start of the
                               next one of the same frame = the k_wf_shade launch between them (and the launch gap on either side) */
  uint32_t shade_launches;  /* the shade launches shade_ms covers (the last one of each frame is not bracketed and not counted) */
} vkrt_trace_timing;
int vkrt_last_trace_timing(vkrt_scene* scene, vkrt_trace_timing* out);

/* ---- test hooks (used by tests/ to compare single pieces with the oracle) ---------- */
/* Structural check of the built acceleration structure (downloads it; host walk).  A tree is sound when every triangle slot
 * is referenced by exactly one leaf, every node is reached exactly once from the root, every instanced triangle owns at least one
 * slot, and every triangle is covered by the boxes above its slots: with one slot, its three vertices lie inside the decoded box
 * of each of that slot's ancestors (the quantised boxes are conservative); with several (VKRT_OPT_SPLIT_BUDGET), every point of a
 * 45-point barycentric lattice on the triangle (vertices, edges, interior) lies inside ALL ancestor boxes of at least one of its
 * slots -- the slots together must leave no part of the triangle unreachable.  Any builder, both layouts. */
typedef struct vkrt_accel_check {
  uint64_t nodes_reached;        /* == vkrt_accel_info.node_count */
  uint64_t triangles_referenced; /* leaf references in total */
  uint64_t triangles_missing;    /* slots no leaf references */
  uint64_t triangles_repeated;   /* references beyond the first of a slot */
  uint64_t box_violations;       /* single-slot triangles: (triangle, ancestor slot) pairs with a vertex outside the slot's box */
  uint64_t bad_references;       /* child / triangle indices out of range, nodes reached twice */
  uint32_t max_depth;            /* nodes on the longest root-to-leaf path */
  uint32_t layout;               /* 1 = 8-wide compressed, 0 = BVH2 */
  uint64_t triangles_uncovered;  /* multi-slot triangles with a lattice point that no slot's chain of boxes contains; triangles without any slot */
  uint64_t triangles_split;      /* triangles that own more than one slot */
} vkrt_accel_check;
int vkrt_debug_check_accel(vkrt_scene* scene, vkrt_accel_check* out);
/* Closest-hit query for n rays: o,d = vec3[n] host arrays; tmin/tmax scalars.
 * Writes t,u,v (float[n]) and the flattened triangle id gid (int32[n], -1 = miss). */
int vkrt_debug_trace_rays(vkrt_scene* scene, uint32_t n, const float* origins,
                          const float* directions, float tmin, float tmax, int any_hit,
                          float* t, float* u, float* v, int32_t* gid);
/* Evaluate a device math primitive elementwise (op: 0 sin, 1 cos, 2 sqrt, 3 a/b,
 * 4 pow5, 5 1/sqrt-normalise x of (a,b,0)); host arrays in/out. */
int vkrt_debug_eval_math(int device, int op, uint32_t n, const float* a, const float* b,
                         float* out);

#ifdef __cplusplus
}
#endif
#endif /* VKRT_H */
