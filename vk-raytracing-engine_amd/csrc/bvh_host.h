// bvh_host.h -- host-side geometry flattening and BVH2 construction (see bvh_host.cpp).
#pragma once
#include <cstdint>
#include <vector>
#include "../../include/vkrt.h"

namespace vkrt {

// One instanced triangle in world space: (v0, e1 = v1-v0, e2 = v2-v0) + ids.
// gid = position in instance-major, primitive-minor order (the closest-hit tie-break key).
struct FlatTri
{
  float v0[3], e1[3], e2[3];
  uint32_t gid, inst, prim;
};

struct BuiltBvh
{
  std::vector<float> nodes;        // 16 floats (64 B) per node, device layout (device_scene.h)
  std::vector<uint32_t> triOrder;  // triangle slot -> index into the FlatTri array
  int32_t rootRef = (int32_t)0x80000000;
  uint32_t maxDepth = 0;
  float sahCost = 0;
};

// object->world rows (3x4 row-major, from the column-major node matrix) -> inverse 3x3 rows
void invert3x3_rows(const float o2w[12], float w2o[9]);

// TLAS semantics of hello_vulkan.cpp:1035-1043: one instance per node, geometry = the node's primMesh
// (primitiveCount = indexCount/3, firstVertex = vertexOffset; hello_vulkan.cpp:955-987).
void flatten_instances(const float* positions, const uint32_t* indices, const vkrt_prim_mesh* pm, const vkrt_node* nodes,
                       uint32_t nodeCount, std::vector<FlatTri>& out);

void build_sah_host(const std::vector<FlatTri>& tris, uint32_t maxLeaf, BuiltBvh& out);

// 48-byte device triangle records in slot order
void pack_triangles(const std::vector<FlatTri>& tris, const std::vector<uint32_t>& order, std::vector<float>& out);

}  // namespace vkrt
