// Layout of one 8-wide BVH node, shared by the two encoders (bvh_host.cpp, wide_collapse.hip) and the node test (traverse_wide.h).
//
// Words 0..7 are the same in both formats (bvh_host.h): origin, grid exponents | imask, childBase, triBase, eight meta bytes.
//   VKRT_WNODE_F16 = 0: 80 B.  Child boxes as 8-bit grid coordinates, four children per dword: words 8..19 =
//                       lo.x[0..3] lo.x[4..7] lo.y.. lo.z.. hi.x.. hi.y.. hi.z..
//   VKRT_WNODE_F16 = 1: 128 B = one cache line.  The same grid coordinates stored as binary16 (integers up to 2048 are exact), two
//                       children per dword: words 8 + 4 p + (slot >> 1), p = 0..5 for lo.x lo.y lo.z hi.x hi.y hi.z.  The node test
//                       then feeds them to v_fma_mix_f32 (conversion + FMA in one half-rate instruction) instead of
//                       v_cvt_f32_ubyteN (half rate) + v_fma_f32, and the grid may have VKRT_WNODE_QMAX = 2047 cells per axis.
#pragma once
#include <cstdint>

#ifndef VKRT_WNODE_F16
#if defined(VKRT_EXP) && (VKRT_EXP == 6 || VKRT_EXP == 7)
#define VKRT_WNODE_F16 1
#else
#define VKRT_WNODE_F16 0
#endif
#endif

#if VKRT_WNODE_F16
#define VKRT_WNODE_QUADS 8
#if defined(VKRT_EXP) && VKRT_EXP == 6
#define VKRT_WNODE_QMAX 255  // (experiment: the 8-bit grid in the 128-B format, separates the instruction effect from the tighter boxes)
#else
#define VKRT_WNODE_QMAX 2047
#endif
#else
#define VKRT_WNODE_QUADS 5
#define VKRT_WNODE_QMAX 255
#endif
#define VKRT_WNODE_DWORDS (4 * VKRT_WNODE_QUADS)
#define VKRT_WNODE_BYTES (16 * VKRT_WNODE_QUADS)

// nodes [0, VKRT_TOP_NODES) are kept in LDS by the sharing traversal wave (0 = off); the node arrays are allocated with at least that
// many nodes so that the copy never reads past the end
#if defined(VKRT_EXP) && VKRT_EXP == 8
#define VKRT_TOP_NODES 9
#elif defined(VKRT_EXP) && VKRT_EXP == 9
#define VKRT_TOP_NODES 6
#else
#define VKRT_TOP_NODES 0
#endif
#define VKRT_WNODE_MIN_ALLOC ((VKRT_TOP_NODES > 1 ? VKRT_TOP_NODES : 1) * VKRT_WNODE_BYTES)

#if defined(__HIPCC__)
#define VKRT_WN_HD __host__ __device__ inline
#else
#define VKRT_WN_HD inline
#endif

// binary16 bits of an integer 0..2048 (exact)
VKRT_WN_HD uint32_t vkrt_wnode_half_bits(uint32_t q)
{
  if(q == 0u)
    return 0u;
  uint32_t e = 0;
  while((q >> (e + 1u)) != 0u) e++;
  return ((e + 15u) << 10) | ((q << (10u - e)) & 0x3ffu);
}

// words 8.. of a node from the grid coordinates of its eight child slots (empty slots: 0)
VKRT_WN_HD void vkrt_wnode_store_planes(uint32_t* n, const uint16_t qlo[3][8], const uint16_t qhi[3][8])
{
  for(int p = 0; p < 6; p++)
  {
    const uint16_t* q = p < 3 ? qlo[p] : qhi[p - 3];
#if VKRT_WNODE_F16
    for(int w = 0; w < 4; w++)
      n[8 + 4 * p + w] = vkrt_wnode_half_bits(q[2 * w]) | (vkrt_wnode_half_bits(q[2 * w + 1]) << 16);
#else
    for(int w = 0; w < 2; w++)
      n[8 + 2 * p + w] = (uint32_t)q[4 * w] | ((uint32_t)q[4 * w + 1] << 8) | ((uint32_t)q[4 * w + 2] << 16) | ((uint32_t)q[4 * w + 3] << 24);
#endif
  }
}
