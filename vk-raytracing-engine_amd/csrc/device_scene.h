// device_scene.h -- HBM-resident scene + acceleration structure as the kernels see it.
//
// Flat geometry buffers are the reference's own (hello_vulkan.cpp:353-379, read by
// raytrace.rchit:41-66): SoA positions/normals/tangents/uv + u32 indices + PrimMeshInfo.
// The acceleration structure replaces VK_KHR_acceleration_structure (hello_vulkan.cpp:955-1047):
// all TLAS instances are flattened to ONE world-space BVH.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vkrt_host_device.h"
#include "device_math.h"

// One TLAS instance (hello_vulkan.cpp:1035-1043): object->world rows + inverse 3x3 + primMesh.
struct DevInstance
{
  float o2w[12];  // row-major 3x4
  float w2o[9];   // row-major 3x3 inverse of the upper-left block
  int32_t primMesh;
  int32_t pad[2];
};
static_assert(sizeof(DevInstance) == 96, "DevInstance");

// Shading-side repack (done once at upload; the API still takes the reference's SoA arrays):
// the per-CU L1/TA processes one 16-byte lane-load per tag lookup, so records are laid out for few,
// wide, aligned loads instead of the ~70 dword loads per hit that the raw SoA layout needs.
//   vertex  : 3 x float4 = (pos.xyz, nrm.x) (nrm.y, nrm.z, uv.x, uv.y) (tangent.xyzw): 48 bytes, one or two cache lines per vertex
//             (round 4: the tangent used to live in an array of its own, a third line per vertex)
//   material: 128 bytes = GltfPBRMaterial (52 B) padded to four aligned float4 + the descriptors of its four
//             textures (base colour, metallic-roughness, normal, emissive), so a hit reaches its texels in
//             one hop from the material record instead of two (no separate descriptor-table lookup)
#define VKRT_VERTEX_QUADS 3
#define VKRT_VERTEX_BYTES 48u
struct DevTexRef
{
  uint32_t offset;  // first texel in the RGBA8 pool (0 when invalid)
  uint32_t wh;      // width | height << 16 (1 | 1 << 16 when invalid)
  uint32_t flags;   // bit 0: index refers to an uploaded texture; bit 1: sRGB
  uint32_t quads;   // first record of the texture's level 0 in the footprint pool (DevScene::texQuads; 0 when invalid or no such pool)
};
#define VKRT_TEXREF_BASE 0
#define VKRT_TEXREF_MR 1
#define VKRT_TEXREF_NORMAL 2
#define VKRT_TEXREF_EMISSIVE 3
struct DevMaterial
{
  GltfPBRMaterial m;
  int32_t pad[3];
  DevTexRef tex[4];
};
static_assert(sizeof(DevMaterial) == 128, "DevMaterial");
// The hit shader's view of a material (shade.h closestHitFront): the factors raytrace.rchit reads and the four texture references
// in 64 bytes = four aligned 16-byte loads per hit (DevMaterial: eight).
//   f = (baseColor.xyz, metallic, roughness, emissive.xyz)   ref[2 k], ref[2 k + 1] = reference k (VKRT_TEXREF_*):
//   dims = (width - 1) | used << 15 | (height - 1) << 16 | valid << 31   (used: the material's texture index is > -1; valid: it
//          names an uploaded texture; an invalid reference has width = height = 1), sides up to 32768
//   base = first record of the texture in the footprint pool (DevScene::texQuads), or its first texel when the scene has no such
//          pool; | sRGB << 31
struct DevShadeMaterial
{
  float f[8];
  uint32_t ref[8];
};
static_assert(sizeof(DevShadeMaterial) == 64, "DevShadeMaterial");

#define VKRT_MAX_MIPS 16
struct DevTexture
{
  uint32_t offset;  // first texel of level 0 in the RGBA8 pool
  uint32_t width, height;
  uint32_t srgb;    // bit 0: sRGB; bits 8..: number of mip levels (>= 1); level offsets in DevScene::texMips
};

// BVH2 node, 64 bytes = 4 x float4 (one 64-B line, four dwordx4 loads):
//   q0 = (lo0.x, lo0.y, lo0.z, hi0.x)  q1 = (hi0.y, hi0.z, lo1.x, lo1.y)
//   q2 = (lo1.z, hi1.x, hi1.y, hi1.z)  q3 = (child0, child1, 0, 0) as int bits
// child >= 0: internal node index; child < 0: leaf, ~child = (firstTri << 3) | (count-1).
#define VKRT_NODE_QUADS 4
#define VKRT_LEAF_MAX 8
#define VKRT_TRAV_DONE ((int)0x80000000)

// Triangle record, 48 bytes = 3 x float4, in leaf order:
//   a = (v0.x, v0.y, v0.z, e1.x)  b = (e1.y, e1.z, e2.x, e2.y)  c = (e2.z, gid, inst, prim) (ints as bits)
#define VKRT_TRI_QUADS 3

struct DevScene
{
  const float* positions;     // vec3[]
  const uint32_t* indices;
  const float4* vertexPN;     // VKRT_VERTEX_QUADS float4 per vertex (see above)
  const DevMaterial* materials;
  const DevShadeMaterial* shadeMaterials;  // the same materials as the path tracer's hit shader reads them
  const GltfLight* lights;
  const DevInstance* instances;
  const DevTexture* textures;
  const uint32_t* texMips;    // [textureCount][VKRT_MAX_MIPS]: first texel of every mip level (hello_vulkan.cpp:496 cmdGenerateMipmaps)
  const uint32_t* texels;     // RGBA8 pool (all levels)
  const uint4* texQuads;      // footprint pool of every texture's level 0, or NULL: record (x, y) = the texels (x, y) (x+1, y) (x, y+1) (x+1, y+1) with REPEAT
                              // wrap -- the 2x2 footprint of a bilinear tap as ONE 16-byte lane-load instead of four 4-byte ones (the hit shader is
                              // bound by the L1's lane-load rate, not by bytes: profiles/r04_experiments.md #115)
  const float* srgbLut;       // 512 floats: [0,256) sRGB decode of i/255, [256,512) i/255 (UNORM decode)
  const float4* nodes;
  const float4* tris;
  const uint4* triShade;      // per triangle slot: absolute vertex indices i0,i1,i2 and max(0, materialIndex)
  uint32_t textureCount;
  uint32_t triCount;
  int32_t rootRef;            // 0 (internal root) or a leaf ref for tiny scenes
  uint32_t stackCap;          // traversal stack: 4-byte LDS words per lane
  uint32_t layout;            // 0 = BVH2 (64-B nodes), 1 = wide8 (80-B compressed nodes)
  uint32_t stepLimit;         // traversal step bound (termination safety net)
  uint32_t triThreshold;      // wide8: lanes with pending triangles needed before a wave tests them (0 = test at once)
  uint32_t sharePeriodMask;   // work sharing is attempted on steps with (step & mask) == mask (0 = every step)
  uint32_t shareMinIdle;      // wide8, wavefront mode: idle lanes of a wave take over pending subtrees of busy lanes once this many are idle (0 = off)
  uint32_t gbufferMips;       // hybrid G-buffer: 1 = implicit-LOD texture() as in a fragment shader (trilinear + 4x anisotropy), 0 = LOD 0
  uint32_t shareFlags;        // bit 0: lanes whose stack is empty also donate the farthest pending child of their current group; bits 1, 2: order of
                              // any-hit walks (traverse.h); bit 4: lanes with nothing else to give donate half of their pending triangles
  uint32_t watertight;        // 1: triangle records hold (p0, p1, p2) and the kernels run the watertight test (VKRT_OPT_WATERTIGHT)
  float sceneLo[3], sceneHi[3];  // world-space bounds of the instanced geometry (conservative; read by the any-hit order heuristic only)
  uint32_t dissolve;          // 1: any-hit alpha / dissolve stage (VKRT_OPT_ANYHIT_DISSOLVE): bit 31 of a record's id word flags a non-opaque triangle
  unsigned long long* faults; // sticky tally of dropped stack pushes + step-limit exits (a walk that was cut short); must stay 0
};
// a traversal could not keep a pending subtree (stack full) or ran into the step bound: the result may be wrong -> make it visible
#define VKRT_TRAV_FAULT(sc) atomicAdd((sc).faults, 1ull)

#ifndef VKRT_W8_MAX_POSTPONED
#define VKRT_W8_MAX_POSTPONED 8  // parked triangle groups per lane (traverse_wide.h, traverse_share.h), held in the free top end of its stack column
#endif
#ifndef VKRT_W8_POSTPONE_ROOM
#define VKRT_W8_POSTPONE_ROOM 0  // entries of the column reserved for them on top of the node stack's own depth (vkrt_api.cpp); groups parked beyond
                                 // these are tested out when a node push needs the slot (profiles/r04_experiments.md #117)
#endif

// Counter storage: 64 slots of 10 counters (padded to two 64-byte lines, order of vkrt_counters).  A workgroup
// adds its block-reduced totals to slot (blockIdx % 64), so same-address atomic serialisation is
// 64x lower than with one set of counters; vkrt_counters_read sums the slots.
#define VKRT_COUNTER_SLOTS 64
struct DevCounters
{
  unsigned long long v[VKRT_COUNTER_SLOTS][VKRT_COUNTER_STRIDE];
};

// TraceParams.flags: the public vkrt_trace_flags (include/vkrt.h) in the low bits + internal launch-uniform switches
#define VKRT_TRACE_PUBLIC_FLAGS 0xFu
#define VKRT_FLAG_SKIP_DEAD_SHADOW 0x100u  // VKRT_OPT_SKIP_DEAD_SHADOW_RAYS: a diffuse hit whose contribution is exactly zero emits no shadow ray
#define VKRT_FLAG_STORE_STAGED 0x200u      // frames in flight: `image` is the frame's staging plane; storePixel writes the pixel value unblended

struct TraceParams
{
  DevScene sc;
  PushConstantRay pc;
  float viewInverse[16];
  float projInverse[16];
  uint32_t seed;
  uint32_t flags;
  uint32_t fullW, fullH;
  uint32_t stripRows, shardCount, shardIndex;
  uint32_t localRows;       // rows in this shard's buffer
  uint32_t tilesX, tileCount;
  uint32_t tileFirst;       // wavefront mode: first 8x8 tile of this launch's sub-frame (0 elsewhere)
  float* image;             // rgba32f, localRows x fullW
  unsigned int* workCounter;
  DevCounters* counters;
};

// Wavefront-mode working set (wavefront.hip): six record streams [parity][type] of SoA float4 planes + their counts, once per
// frame group (frames in flight), and the staging plane of each group.
struct WfBuffers
{
  unsigned* ctrl;       // stream counts [parity * 4 + type]; 64 words per lane
  float4* planes;       // [group][parity][17 planes: 8 shared by the C and S streams, 9 of the pair stream][capacity] (wavefront.hip plane())
  float4* stage;        // [group][capacity]: pixel values of a frame in flight, shard-local image layout (groups > 1 only)
  uint32_t capacity;    // paths (pixels of the shard, rounded up to whole 8x8 tiles)
  uint32_t groups;      // frame groups the allocation holds
};
