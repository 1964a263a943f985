// lbvh.h -- device LBVH builder entry (see lbvh.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/vkrt.h"
#include "device_scene.h"

namespace vkrt {

// Device-side collapse of the binary tree into the 8-wide compressed layout (wide_collapse.hip).
struct WideCollapseIn
{
  uint32_t triCount;
  const float4* nodes2;        // BVH2 nodes, one triangle per leaf (k_emit with leaf size 1): node i = radix-tree node i
  const int* parentInternal;   // radix-tree parents (k_hierarchy)
  const int* parentLeaf;
  const float4* tris;          // 48-B records in sorted (leaf) order
  const uint4* triShade;       // 16-B shading records in the same order
};
struct WideCollapseOut
{
  void* nodes = nullptr;       // device, 80 B per wide node, breadth-first (caller frees with hipFree)
  void* tris = nullptr;        // device, 48 B per triangle in wide-tree order
  void* triShade = nullptr;    // device, 16 B per triangle in the same order
  uint32_t nodeCount = 0, maxDepth = 0;
  float sahCost = 0;
  bool overflow = false;       // tree deeper than the level budget: results unusable, collapse on the host instead
};
int collapse_wide8_device(const WideCollapseIn& in, hipStream_t stream, WideCollapseOut& out, std::string& err);

struct LbvhResult
{
  void* nodes = nullptr;  // device, 64 B per node (caller frees with hipFree)
  void* tris = nullptr;   // device, 48 B per triangle in leaf order
  void* triShade = nullptr;  // device, 16 B per triangle in leaf order (vertex indices + material)
  uint32_t triCount = 0, nodeCount = 0, maxDepth = 0;  // triCount = triangle SLOTS = references (leaves) of the tree
  uint32_t uniqueTris = 0;                             // instanced triangles (== triCount without pre-splitting)
  int32_t rootRef = (int32_t)0x80000000;
  float sahCost = 0;
  std::string error;
  // filled when the caller asked for the wide layout (leafSize 1, wantWide): the binary arrays above stay valid as well
  WideCollapseOut wide;
  bool hasWide = false;
};

// Binary hierarchy over the Morton-sorted triangles by parallel locally-ordered clustering (ploc.hip): fills the same arrays as
// k_hierarchy + k_fit of lbvh.hip (children, parents, boxes; range[i] = (0, triangles below i - 1)); node 0 is the root.
int ploc_cluster_device(uint32_t T, const unsigned* order, const float* triBox, hipStream_t stream, int2* children, int2* range, int* parentInternal,
                        int* parentLeaf, float* nodeBox, unsigned* passes, std::string& err);

// sc must already hold the uploaded positions / indices / instances.  ploc: cluster (ploc.hip, one triangle per leaf) instead of
// the Morton radix tree.  splitPercent: budget of extra references for triangle pre-splitting, in percent of the triangle count (0 = off).
int build_lbvh_device(const DevScene& sc, uint32_t instCount, const std::vector<vkrt_prim_mesh>& pm, const std::vector<vkrt_node>& nodes,
                      hipStream_t stream, LbvhResult& out, unsigned leafSize = 4, bool wantWide = false, bool ploc = false, bool watertight = false, bool dissolve = false,
                      unsigned splitPercent = 0);

}  // namespace vkrt
