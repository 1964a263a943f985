// Layout of one 8-wide BVH node, shared by the two encoders (bvh_host.cpp, wide_collapse.hip) and the node test (traverse_wide.h).
//
// 80 B = 5 x float4.  Words 0..7 (bvh_host.h): origin, grid exponents | imask, childBase, triBase, eight meta bytes.  Words 8..19: the
// child boxes as 8-bit grid coordinates, four children per dword: lo.x[0..3] lo.x[4..7] lo.y.. lo.z.. hi.x.. hi.y.. hi.z..
// (A 128-B variant with binary16 planes fed to v_fma_mix_f32 saved 48 VALU instructions per node test and was 13-15 % slower: eight
// 16-B loads per lane instead of five; profiles/r03_experiments.md #95.  It lives in the history, not here.)
#pragma once
#include <cstdint>

#define VKRT_WNODE_QUADS 5
#define VKRT_WNODE_QMAX 255
#define VKRT_WNODE_DWORDS (4 * VKRT_WNODE_QUADS)
#define VKRT_WNODE_BYTES (16 * VKRT_WNODE_QUADS)
#define VKRT_WNODE_MIN_ALLOC VKRT_WNODE_BYTES

#if defined(__HIPCC__)
#define VKRT_WN_HD __host__ __device__ inline
#else
#define VKRT_WN_HD inline
#endif

// words 8.. of a node from the grid coordinates of its eight child slots (empty slots: 0)
VKRT_WN_HD void vkrt_wnode_store_planes(uint32_t* n, const uint16_t qlo[3][8], const uint16_t qhi[3][8])
{
  for(int p = 0; p < 6; p++)
  {
    const uint16_t* q = p < 3 ? qlo[p] : qhi[p - 3];
    for(int w = 0; w < 2; w++)
      n[8 + 2 * p + w] = (uint32_t)q[4 * w] | ((uint32_t)q[4 * w + 1] << 8) | ((uint32_t)q[4 * w + 2] << 16) | ((uint32_t)q[4 * w + 3] << 24);
  }
}
