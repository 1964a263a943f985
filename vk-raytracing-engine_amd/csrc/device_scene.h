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

// One TLAS instance (hello_vulkan.cpp:1035-1043): object->world rows + inverse 3x3 + primMesh.
struct DevInstance
{
  float o2w[12];  // row-major 3x4
  float w2o[9];   // row-major 3x3 inverse of the upper-left block
  int32_t primMesh;
  int32_t pad[2];
};
static_assert(sizeof(DevInstance) == 96, "DevInstance");

struct DevTexture
{
  uint32_t offset;  // first texel in the RGBA8 pool
  uint32_t width, height;
  uint32_t srgb;
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
  const float* normals;       // vec3[]
  const float* tangents;      // vec4[]
  const float* texcoords;     // vec2[]
  const uint32_t* indices;
  const PrimMeshInfo* primInfo;
  const GltfPBRMaterial* materials;
  const GltfLight* lights;
  const DevInstance* instances;
  const DevTexture* textures;
  const uint32_t* texels;     // RGBA8 pool
  const float* srgbLut;       // 256 floats
  const float4* nodes;
  const float4* tris;
  uint32_t textureCount;
  uint32_t triCount;
  int32_t rootRef;            // 0 (internal root) or a leaf ref for tiny scenes
  uint32_t stackCap;          // traversal stack entries per lane
  uint32_t stepLimit;         // traversal step bound (termination safety net)
};

struct DevCounters  // order of vkrt_counters
{
  unsigned long long v[8];
};

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
  float* image;             // rgba32f, localRows x fullW
  unsigned int* workCounter;
  DevCounters* counters;
};
