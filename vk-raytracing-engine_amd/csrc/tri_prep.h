// tri_prep.h -- per-triangle set-up shared by the three builders (device flatten in lbvh.hip, host SAH build in
// bvh_host.cpp) and restated in oracle/oracle.cpp: the triangle's bounding box as the builders see it.
//
// Why the box is wider than the three vertices (round 3, tools/fuzz_parity.py seed 1301004260).  The binary32
// Moeller-Trumbore test (traverse.h tri_test) computes u and v with an absolute error of about
//     eps * |o - v0| / (sin(phi) * cos(theta))        phi = angle between e1 and e2 at v0, theta = incidence angle,
// in units of the edge length, i.e. it accepts points up to  eps * |o - v0| / (sin phi cos theta)  OUTSIDE the triangle
// along its long direction.  For ordinary triangles that is far below the box-test margins (traverse.h); for a needle
// (two edges of 750 units enclosing 7e-4 rad in the failing case, ray origin near its tip) it was 0.06 units -- beyond
// the box of a one-triangle leaf, so whether the "hit" was found depended on the tree (brute force found it, a tight leaf
// box pruned it).  The result must be a property of the triangle set, so every builder widens the box of each triangle
// by the reach of the test for origins within one triangle length,
//     slop = 16 eps * max(|e1|, |e2|) / sin(phi)      (capped at the triangle's own length; zero when sin(phi) >= 1/8),
// which is zero for ordinary triangles (the bench scene's trees do not change) and grows only for needles.  Origins much farther away than the triangle's length are covered by the box tests' relative
// margins (2e-5 of the distance) down to sin(phi) ~ 5e-3; flatter needles seen from afar remain the documented limit
// of tree-independence.  With the watertight test (VKRT_OPT_WATERTIGHT) the reach is ~1e-7 of
// the distance for every shape and the slop is not needed (it is applied all the same: one rule for the boxes).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define VKRT_HD __host__ __device__ inline
#else
#define VKRT_HD inline
#endif

// (plain float arithmetic in source order; the translation units that include this are built with -ffp-contract=off)
VKRT_HD float vkrt_tri_slop(const float e1[3], const float e2[3])
{
  const float l1 = (e1[0] * e1[0] + e1[1] * e1[1]) + e1[2] * e1[2];
  const float l2 = (e2[0] * e2[0] + e2[1] * e2[1]) + e2[2] * e2[2];
  const float cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
  const float a2 = (cx * cx + cy * cy) + cz * cz;
  if(!(a2 > 0.0f))
    return 0.0f;  // degenerate: the test's determinant is zero or garbage either way, nothing to cover
  const float len = sqrtf(l1 > l2 ? l1 : l2);
  const float invSin = sqrtf((l1 * l2) / a2);              // 1 / sin(phi) >= 1 (inf for l1 * l2 overflow: capped below)
  // Corners wider than ~7 degrees: the reach (< 8 eps |o - v0| / cos theta) is inside the box tests' own margins, as 144 k cases of
  // the round-2 campaign showed, and the box stays exact -- it matters: on grid-aligned geometry (the bench scene) an extra 1e-7
  // pushes every 8-bit quantised child box of the wide nodes out by a whole cell (+4.5 % node visits, measured).
  if(!(invSin > 8.0f))
    return 0.0f;
  const float s = (9.5367431640625e-07f * len) * invSin;    // 16 * 2^-24 * len / sin(phi)
  return s < len ? s : len;                                 // (NaN -> len)
}

// Box of the triangle as the selected test sees it -- (p0, p0 + e1, p0 + e2), the vertices the Moeller-Trumbore records reconstruct,
// or, watertight, the exact vertices (p0, p1, p2) its records hold -- widened by the slop.
VKRT_HD void vkrt_tri_bounds(const float p0[3], const float p1[3], const float p2[3], const float e1[3], const float e2[3], int watertight, float lo[3],
                             float hi[3])
{
  const float slop = vkrt_tri_slop(e1, e2);
  for(int k = 0; k < 3; k++)
  {
    const float q1 = watertight ? p1[k] : p0[k] + e1[k], q2 = watertight ? p2[k] : p0[k] + e2[k];
    float l = p0[k], h = p0[k];
    l = q1 < l ? q1 : l; h = q1 > h ? q1 : h;
    l = q2 < l ? q2 : l; h = q2 > h ? q2 : h;
    lo[k] = l - slop;
    hi[k] = h + slop;
  }
}
