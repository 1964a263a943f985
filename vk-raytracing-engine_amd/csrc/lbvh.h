// lbvh.h -- device LBVH builder entry (see lbvh.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/vkrt.h"
#include "device_scene.h"

namespace vkrt {

struct LbvhResult
{
  void* nodes = nullptr;  // device, 64 B per node (caller frees with hipFree)
  void* tris = nullptr;   // device, 48 B per triangle in leaf order
  void* triShade = nullptr;  // device, 16 B per triangle in leaf order (vertex indices + material)
  uint32_t triCount = 0, nodeCount = 0, maxDepth = 0;
  int32_t rootRef = (int32_t)0x80000000;
  float sahCost = 0;
  std::string error;
};

// sc must already hold the uploaded positions / indices / instances.
int build_lbvh_device(const DevScene& sc, uint32_t instCount, const std::vector<vkrt_prim_mesh>& pm, const std::vector<vkrt_node>& nodes,
                      hipStream_t stream, LbvhResult& out, unsigned leafSize = 4);

}  // namespace vkrt
