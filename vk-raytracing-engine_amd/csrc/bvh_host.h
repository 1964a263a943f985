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
  float p1[3], p2[3];  // the exact world-space vertices 1 and 2 (v0 + e1 rounds away from p1): what the watertight records hold
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

void build_sah_host(const std::vector<FlatTri>& tris, uint32_t maxLeaf, BuiltBvh& out, bool watertight = false);

// 16-byte shading records in slot order: absolute vertex indices + max(0, materialIndex) (raytrace.rchit:34-50)
void pack_tri_shade(const std::vector<FlatTri>& tris, const std::vector<uint32_t>& order, const uint32_t* indices, const vkrt_prim_mesh* pm,
                    const vkrt_node* nodes, std::vector<uint32_t>& out);

// 48-byte device triangle records in slot order: (v0, e1, e2, ids) for Moeller-Trumbore, (p0, p1, p2, ids) when watertight
void pack_triangles(const std::vector<FlatTri>& tris, const std::vector<uint32_t>& order, std::vector<float>& out, bool watertight = false,
                    const std::vector<uint8_t>* instDissolves = nullptr);  // instDissolves[inst] != 0: the instance's material is not opaque

}  // namespace vkrt

namespace vkrt {

// 8-wide compressed BVH ("wide8"), 80 bytes per node = 5 x float4 (layout after Ylitie, Karras, Laine 2017,
// "Efficient Incoherent Ray Traversal on GPUs Through Compressed Wide BVHs"; re-derived for gfx950 where the
// per-CU L1/TA tag rate bounds incoherent traversal: 5 lane-loads cover what 3 binary levels need 12 for):
//   q0 = (origin.x, origin.y, origin.z, ex | ey<<8 | ez<<16 | imask<<24)        e* = biased exponents of the grid
//   q1 = (childBase, triBase, meta[0..3], meta[4..7])                            one meta byte per child slot
//   q2 = (qlo.x[0..3], qlo.x[4..7], qlo.y[0..3], qlo.y[4..7])                   8-bit quantised child boxes
//   q3 = (qlo.z[0..3], qlo.z[4..7], qhi.x[0..3], qhi.x[4..7])
//   q4 = (qhi.y[0..3], qhi.y[4..7], qhi.z[0..3], qhi.z[4..7])
// meta: 0 empty; internal child = 0x20 | (24 + slot); leaf = (unary triangle count: 1,3,7) << 5 | offset in the
// node's triangle block (0..23).  Child box k decodes to origin + q * 2^(e-127), conservatively (floor/ceil
// verified in double).  Internal children are stored consecutively from childBase in slot order.
struct BuiltWide8
{
  std::vector<uint32_t> nodes;     // 20 dwords per node
  std::vector<uint32_t> triOrder;  // triangle slot -> index into the FlatTri array
  uint32_t maxDepth = 0;           // in wide nodes
  uint32_t nodeCount = 0;
  float sahCost = 0;
};
void build_wide8_host(const std::vector<FlatTri>& tris, BuiltWide8& out, bool watertight = false);
// SAH-optimal collapse of an existing binary tree whose leaves hold <= 3 triangles each (ideally 1).
// triCount0Box: bounds of the first triangle, used only when the binary root is a leaf.
void collapse_wide8(const BuiltBvh& b2, const std::vector<FlatTri>& tris, BuiltWide8& out, bool watertight = false);

}  // namespace vkrt
