// lbvh.hip -- GPU LBVH builder (VKRT_BUILD_LBVH_GPU): the device-side replacement for the
// driver's acceleration-structure build that the reference triggers through
// nvvk::RaytracingBuilderKHR::buildBlas/buildTlas (hello_vulkan.cpp:1001-1047).
//
// Pipeline (all on the caller's stream, one-off per scene):
//   k_flatten   instances x primitives -> world-space (v0,e1,e2) records + boxes + scene bounds
//   k_morton    63-bit Morton code of each box centre (21 bits per axis)
//   rocprim     radix sort of (code, gid) pairs (stable: equal codes stay in gid order)
//   k_hierarchy Karras 2012 radix tree over the sorted codes (ties resolved by position)
//   k_fit       bottom-up boxes with per-node arrival counters
//   k_emit      64-byte two-child nodes; subtrees of <= LEAF triangles collapse into one leaf
//   k_pack      48-byte triangle records in sorted (leaf) order
// Triangle arithmetic is the same sequence as the host flatten (bvh_host.cpp) so both builders
// hand bit-identical triangles to the traversal kernel.
// VKRT_BUILD_PLOC_GPU swaps k_hierarchy + k_fit for the clustering of ploc.hip (better trees, a few more passes).
#include <hip/hip_runtime.h>
#include <cstring>
#include <string>
#include <algorithm>
#include <cstdlib>
#include <vector>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "device_math.h"
#include "lbvh.h"
#include "tri_prep.h"

namespace vkrt {

namespace {


struct FlatArgs
{
  const float* positions;
  const uint32_t* indices;
  const DevInstance* instances;
  const uint32_t* instFirstGid;  // [instCount+1]
  const uint32_t* instFirstIndex;
  const uint32_t* instVertexOffset;
  uint32_t instCount;
  uint32_t triCount;
};

VKRT_DEV unsigned encodeOrdered(float f)
{
  const unsigned b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ inline float decodeOrdered(unsigned u)
{
  const unsigned b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  float f;
#if defined(__HIP_DEVICE_COMPILE__)
  f = __uint_as_float(b);
#else
  memcpy(&f, &b, 4);
#endif
  return f;
}

// watertight != 0: the records hold the exact world-space vertices (p0, p1, p2) for the watertight test (traverse.h tri_test_wt)
// instead of (v0, e1, e2) for Moeller-Trumbore
__global__ void k_flatten(FlatArgs A, int watertight, float4* triU, float* triBox /*6 per tri*/, unsigned* sceneBounds /*lo3 hi3 ordered*/)
{
  const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  if(gid < A.triCount)
  {
    // instance lookup: largest n with instFirstGid[n] <= gid
    unsigned a = 0, b = A.instCount;
    while(b - a > 1u)
    {
      const unsigned m = (a + b) >> 1;
      if(A.instFirstGid[m] <= gid) a = m; else b = m;
    }
    const unsigned inst = a, prim = gid - A.instFirstGid[a];
    const DevInstance in = A.instances[inst];
    const uint32_t base = A.instFirstIndex[inst] + 3u * prim, vo = A.instVertexOffset[inst];
    const uint32_t i0 = A.indices[base] + vo, i1 = A.indices[base + 1] + vo, i2 = A.indices[base + 2] + vo;
    f3 p[3];
    const uint32_t ii[3] = {i0, i1, i2};
#pragma unroll
    for(int k = 0; k < 3; k++)
    {
      const f3 q = mk3(A.positions[3 * ii[k]], A.positions[3 * ii[k] + 1], A.positions[3 * ii[k] + 2]);
      p[k].x = ((in.o2w[0] * q.x + in.o2w[1] * q.y) + in.o2w[2] * q.z) + in.o2w[3];
      p[k].y = ((in.o2w[4] * q.x + in.o2w[5] * q.y) + in.o2w[6] * q.z) + in.o2w[7];
      p[k].z = ((in.o2w[8] * q.x + in.o2w[9] * q.y) + in.o2w[10] * q.z) + in.o2w[11];
    }
    const f3 v0 = p[0], e1 = p[1] - p[0], e2 = p[2] - p[0];
    const f3 r1 = watertight ? p[1] : e1, r2 = watertight ? p[2] : e2;
    triU[3 * (size_t)gid + 0] = make_float4(v0.x, v0.y, v0.z, r1.x);
    triU[3 * (size_t)gid + 1] = make_float4(r1.y, r1.z, r2.x, r2.y);
    triU[3 * (size_t)gid + 2] = make_float4(r2.z, __int_as_float((int)gid), __int_as_float((int)inst), __int_as_float((int)prim));
    // the box every builder starts from (tri_prep.h): both forms of the vertices, widened by the reach of the triangle test
    const float q0[3] = {p[0].x, p[0].y, p[0].z}, q1[3] = {p[1].x, p[1].y, p[1].z}, q2[3] = {p[2].x, p[2].y, p[2].z};
    const float a1[3] = {e1.x, e1.y, e1.z}, a2[3] = {e2.x, e2.y, e2.z};
    vkrt_tri_bounds(q0, q1, q2, a1, a2, watertight, lo, hi);
#pragma unroll
    for(int k = 0; k < 3; k++)
    {
      triBox[6 * (size_t)gid + k] = lo[k];
      triBox[6 * (size_t)gid + 3 + k] = hi[k];
    }
  }
  // wave reduce then one atomic per wave per component
#pragma unroll
  for(int k = 0; k < 3; k++)
  {
    float l = lo[k], h = hi[k];
#pragma unroll
    for(int off = 32; off > 0; off >>= 1)
    {
      l = fminf(l, __shfl_xor(l, off));
      h = fmaxf(h, __shfl_xor(h, off));
    }
    if((threadIdx.x & 63u) == 0u)
    {
      if(l <= h)
      {
        atomicMin(&sceneBounds[k], encodeOrdered(l));
        atomicMax(&sceneBounds[3 + k], encodeOrdered(h));
      }
    }
  }
}

VKRT_DEV unsigned long long expand21(unsigned v)
{
  unsigned long long x = v & 0x1fffffull;
  x = (x | x << 32) & 0x1f00000000ffffull;
  x = (x | x << 16) & 0x1f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}

__global__ void k_morton(unsigned n, const float* triBox, const unsigned* sceneBounds, unsigned long long* keys, unsigned* vals)
{
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  unsigned q[3];
#pragma unroll
  for(int k = 0; k < 3; k++)
  {
    const float lo = decodeOrdered(sceneBounds[k]), hi = decodeOrdered(sceneBounds[3 + k]);
    const float c = 0.5f * (triBox[6 * (size_t)i + k] + triBox[6 * (size_t)i + 3 + k]);
    const float ext = hi - lo;
    float t = ext > 0.0f ? (c - lo) / ext : 0.0f;
    t = fminf(fmaxf(t * 2097152.0f, 0.0f), 2097151.0f);
    q[k] = (unsigned)t;
  }
  keys[i] = expand21(q[0]) | (expand21(q[1]) << 1) | (expand21(q[2]) << 2);
  vals[i] = i;
}

// common-prefix length of sorted keys i and j (position breaks ties); -1 outside the array
VKRT_DEV int delta(const unsigned long long* keys, int n, int i, int j)
{
  if(j < 0 || j >= n)
    return -1;
  const unsigned long long a = keys[i], b = keys[j];
  if(a == b)
    return 64 + __clz((unsigned)i ^ (unsigned)j);
  return __clzll((long long)(a ^ b));
}

// Karras 2012, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees", Alg. in section 4.
// child index encoding: >= 0 internal node, < 0 : ~leafPosition
__global__ void k_hierarchy(int n, const unsigned long long* keys, int2* children, int2* range, int* parentInternal, int* parentLeaf)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n - 1)
    return;
  const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
  const int dmin = delta(keys, n, i, i - d);
  int lmax = 2;
  while(delta(keys, n, i, i + lmax * d) > dmin)
    lmax *= 2;
  int l = 0;
  for(int t = lmax / 2; t >= 1; t /= 2)
    if(delta(keys, n, i, i + (l + t) * d) > dmin)
      l += t;
  const int j = i + l * d;
  const int dnode = delta(keys, n, i, j);
  int s = 0;
  int t = l;
  do
  {
    t = (t + 1) >> 1;
    if(delta(keys, n, i, i + (s + t) * d) > dnode)
      s += t;
  } while(t > 1);
  const int gamma = i + s * d + min(d, 0);
  const int lo = min(i, j), hi = max(i, j);
  int2 ch;
  if(lo == gamma) { ch.x = ~gamma; parentLeaf[gamma] = i; }
  else { ch.x = gamma; parentInternal[gamma] = i; }
  if(hi == gamma + 1) { ch.y = ~(gamma + 1); parentLeaf[gamma + 1] = i; }
  else { ch.y = gamma + 1; parentInternal[gamma + 1] = i; }
  children[i] = ch;
  range[i] = make_int2(lo, hi);
  if(i == 0)
    parentInternal[0] = -1;
}

// bottom-up boxes: the second thread to arrive at a node owns it.  Cross-CU visibility of the
// sibling's box follows the agent-scope release/acquire rule (guide section 6, Guideline 16):
// __threadfence() before the arrival atomic (release) and after it (acquire).
__global__ void k_fit(int n, const unsigned* order, const float* triBox, const int2* children, const int* parentInternal,
                      const int* parentLeaf, float* nodeBox, unsigned* arrive)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if(k >= n)
    return;
  int node = parentLeaf[k];
  while(node >= 0)
  {
    __threadfence();
    const unsigned prev = atomicAdd(&arrive[node], 1u);
    if(prev == 0u)
      return;
    __threadfence();
    const int2 ch = children[node];
    float b[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    const int cc[2] = {ch.x, ch.y};
#pragma unroll
    for(int c = 0; c < 2; c++)
    {
      const float* src = cc[c] < 0 ? &triBox[6 * (size_t)order[~cc[c]]] : &nodeBox[6 * (size_t)cc[c]];
      // bypass this CU's L1 (boxes written by other CUs): agent-scope relaxed atomic loads
#pragma unroll
      for(int q = 0; q < 3; q++)
      {
        b[q] = fminf(b[q], __hip_atomic_load(&src[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        b[3 + q] = fmaxf(b[3 + q], __hip_atomic_load(&src[3 + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      }
    }
#pragma unroll
    for(int q = 0; q < 6; q++)
      __hip_atomic_store(&nodeBox[6 * (size_t)node + q], b[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    node = parentInternal[node];
  }
}

VKRT_DEV float boxArea(const float* b)
{
  const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
  return 2.0f * (dx * dy + dy * dz + dz * dx);
}

__global__ void k_emit(unsigned kLeaf, int n, const unsigned* order, const float* triBox, const int2* children, const int2* range, const float* nodeBox,
                       float4* outNodes, float* sahAccum)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n - 1)
    return;
  const int2 rg = range[i];
  const unsigned cnt = (unsigned)(rg.y - rg.x + 1);
  if(cnt <= kLeaf && i != 0)
    return;  // lives inside a collapsed leaf
  const int2 ch = children[i];
  const int cc[2] = {ch.x, ch.y};
  float bx[2][6];
  int ref[2];
  float cost = 1.0f * boxArea(&nodeBox[6 * (size_t)i]);
#pragma unroll
  for(int c = 0; c < 2; c++)
  {
    const float* src;
    if(cc[c] < 0)
    {
      const unsigned pos = (unsigned)~cc[c];
      src = &triBox[6 * (size_t)order[pos]];
      ref[c] = (int)~((pos << 3) | 0u);
      cost += boxArea(src) * 1.0f;
    }
    else
    {
      src = &nodeBox[6 * (size_t)cc[c]];
      const int2 cr = range[cc[c]];
      const unsigned ccnt = (unsigned)(cr.y - cr.x + 1);
      if(ccnt <= kLeaf)
      {
        ref[c] = (int)~(((unsigned)cr.x << 3) | (ccnt - 1u));
        cost += boxArea(src) * (float)ccnt;
      }
      else
        ref[c] = cc[c];
    }
#pragma unroll
    for(int q = 0; q < 6; q++) bx[c][q] = src[q];
  }
  outNodes[4 * (size_t)i + 0] = make_float4(bx[0][0], bx[0][1], bx[0][2], bx[0][3]);
  outNodes[4 * (size_t)i + 1] = make_float4(bx[0][4], bx[0][5], bx[1][0], bx[1][1]);
  outNodes[4 * (size_t)i + 2] = make_float4(bx[1][2], bx[1][3], bx[1][4], bx[1][5]);
  outNodes[4 * (size_t)i + 3] = make_float4(__int_as_float(ref[0]), __int_as_float(ref[1]), 0.0f, 0.0f);
  atomicAdd(sahAccum, cost);
}

// materials != NULL (VKRT_OPT_ANYHIT_DISSOLVE): bit 31 of the id word of every triangle whose material has dissolve
// (pbrBaseColorFactor.a) < 1 is set -- the flag the traversal's any-hit stage looks at (traverse.h anyhit_ignores)
// refTri != NULL (triangle pre-splitting): order[] holds reference numbers and refTri[r] is the triangle reference r belongs to; a
// triangle with several references gets a copy of its record (and of its shading record) in every slot
__global__ void k_pack(unsigned n, const unsigned* order, const float4* triU, float4* outTris, FlatArgs A, const int* instMaterial,
                       uint4* outShade, const DevMaterial* materials, const unsigned* refTri)
{
  const unsigned s = blockIdx.x * blockDim.x + threadIdx.x;
  if(s >= n)
    return;
  const unsigned g = refTri ? refTri[order[s]] : order[s];
  {  // shading record: absolute vertex indices + max(0, materialIndex) (raytrace.rchit:34-50)
    const float4 c = triU[3 * (size_t)g + 2];
    const unsigned inst = (unsigned)__float_as_int(c.z), prim = (unsigned)__float_as_int(c.w);
    const uint32_t base = A.instFirstIndex[inst] + 3u * prim, vo = A.instVertexOffset[inst];
    outShade[s] = make_uint4(A.indices[base] + vo, A.indices[base + 1] + vo, A.indices[base + 2] + vo, (uint32_t)max(0, instMaterial[inst]));
  }
  outTris[3 * (size_t)s + 0] = triU[3 * (size_t)g + 0];
  outTris[3 * (size_t)s + 1] = triU[3 * (size_t)g + 1];
  float4 c = triU[3 * (size_t)g + 2];
  if(materials)
  {
    const unsigned inst = (unsigned)__float_as_int(c.z);
    if(materials[max(0, instMaterial[inst])].m.pbrBaseColorFactor[3] < 1.0f)
      c.y = __int_as_float(__float_as_int(c.y) | (int)0x80000000);
  }
  outTris[3 * (size_t)s + 2] = c;
}

// depth of the emitted tree = max over leaves of the number of emitted ancestors
__global__ void k_depth(unsigned kLeaf, int n, const int2* range, const int* parentInternal, const int* parentLeaf, unsigned* maxDepth)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if(k >= n)
    return;
  unsigned d = 0;
  int node = parentLeaf[k];
  while(node >= 0)
  {
    const int2 rg = range[node];
    if((unsigned)(rg.y - rg.x + 1) > kLeaf || node == 0)
      d++;
    node = parentInternal[node];
  }
  atomicMax(maxDepth, d);
}

// ---- triangle pre-splitting ----------------------------------------------------------------------------------------------
// The reference asks its driver for PREFER_FAST_TRACE (hello_vulkan.cpp:1010, :1046); what separates a trace-quality builder from a
// plain one on artist-made geometry is what it does with triangles that are large against their neighbours: a wall of two room-sized
// triangles, a 15-m moulding, a drapery strip hanging diagonally.  One box around such a triangle overlaps everything near it.
// Here such triangles enter the build as SEVERAL references, each with the box of one piece of the triangle (the triangle clipped to
// a cell of the Morton grid), after Karras & Aila 2013, "Fast Parallel Construction of High-Quality Bounding Volume Hierarchies",
// section 4 (restated from the paper's description):
//   priority of a triangle  p = (2^-level * (area(box) - A_ideal))^(1/3),  level = depth of the most important spatial-median plane of
//                           the scene grid that cuts the box (0 = the plane halving the scene), A_ideal = |n.x| + |n.y| + |n.z| with
//                           n = e1 x e2: the surface area of the boxes of infinitely small pieces;
//   splits of a triangle    s = floor(D * p), D found by bisection so that the splits of all triangles fill the budget;
//   one split               at the most important median plane cutting the piece's box; the remaining splits are dealt to the two
//                           halves in proportion to the longest sides of their boxes.
// Only references multiply: every piece points at the ORIGINAL 48-byte record (copied into each slot by k_pack), the hit test, the
// triangle id of the tie rule and every pixel stay what they were.  A piece's box = bounds of (triangle cut by the plane), clamped
// to the box of the piece it came from -- conservative in floating point (the cut points are padded by 8 ulp of their coordinates)
// and never larger than the triangle's own box.  Needle triangles (tri_prep.h: slop > 0) are not split (k_split_priority).
#define VKRT_SPLIT_GRID_BITS 21          // the Morton grid of k_morton
#define VKRT_SPLIT_MAX_PER_TRI 255u
#define VKRT_SPLIT_STACK 16

struct SplitTri
{
  float v[3][3];   // the vertices as the selected triangle test sees them
  float full[6];   // the triangle's box as every builder sees it (tri_prep.h): exact bounds + slop
  float slop;
};
struct SplitGrid
{
  float lo[3], ext[3];
};

VKRT_DEV void loadSplitTri(const float4* __restrict__ triU, const float* __restrict__ triBox, unsigned g, int watertight, SplitTri& t)
{
  const float4 a = triU[3 * (size_t)g], b = triU[3 * (size_t)g + 1], c = triU[3 * (size_t)g + 2];
  const float r1[3] = {a.w, b.x, b.y}, r2[3] = {b.z, b.w, c.x};
  t.v[0][0] = a.x; t.v[0][1] = a.y; t.v[0][2] = a.z;
  float e1[3], e2[3];
#pragma unroll
  for(int k = 0; k < 3; k++)
  {
    t.v[1][k] = watertight ? r1[k] : t.v[0][k] + r1[k];
    t.v[2][k] = watertight ? r2[k] : t.v[0][k] + r2[k];
    e1[k] = watertight ? r1[k] - t.v[0][k] : r1[k];
    e2[k] = watertight ? r2[k] - t.v[0][k] : r2[k];
    t.full[k] = triBox[6 * (size_t)g + k];
    t.full[3 + k] = triBox[6 * (size_t)g + 3 + k];
  }
  t.slop = vkrt_tri_slop(e1, e2);
}

VKRT_DEV SplitGrid loadSplitGrid(const unsigned* __restrict__ sceneBounds)
{
  SplitGrid G;
#pragma unroll
  for(int k = 0; k < 3; k++)
  {
    G.lo[k] = decodeOrdered(sceneBounds[k]);
    G.ext[k] = decodeOrdered(sceneBounds[3 + k]) - G.lo[k];
  }
  return G;
}

VKRT_DEV unsigned splitCell(const SplitGrid& G, int k, float x)
{
  const float cells = (float)(1u << VKRT_SPLIT_GRID_BITS);
  float t = G.ext[k] > 0.0f ? (x - G.lo[k]) / G.ext[k] : 0.0f;
  t = fminf(fmaxf(t * cells, 0.0f), cells - 1.0f);
  return (unsigned)t;
}

// The most important spatial-median plane that cuts box b: axis (or -1), its position, and the bit index m of the plane in the grid
// (VKRT_SPLIT_GRID_BITS - 1 = the plane halving the scene).  Ties between axes go to the axis on which the box is longest.
VKRT_DEV int splitPlane(const SplitGrid& G, const float* b, float& pos, int& mOut)
{
  int axis = -1, mBest = -1;
  float extBest = 0.0f;
#pragma unroll
  for(int k = 0; k < 3; k++)
  {
    const unsigned qa = splitCell(G, k, b[k]), qb = splitCell(G, k, b[3 + k]);
    if(qa == qb)
      continue;
    const int m = 31 - __clz((int)(qa ^ qb));
    const float c = G.lo[k] + (float)((qb >> m) << m) * (G.ext[k] / (float)(1u << VKRT_SPLIT_GRID_BITS));
    if(!(c > b[k] && c < b[3 + k]))
      continue;  // (rounding put the plane on the box's face: nothing to cut on this axis)
    const float ext = b[3 + k] - b[k];
    if(m > mBest || (m == mBest && ext > extBest))
    {
      axis = k; mBest = m; extBest = ext; pos = c;
    }
  }
  mOut = mBest;
  return axis;
}

// Karras & Aila's priority (see above); 0 = not worth a split
VKRT_DEV float splitPriority(const SplitGrid& G, const SplitTri& t)
{
  float b[6];
#pragma unroll
  for(int k = 0; k < 3; k++)
  {
    b[k] = fminf(t.v[0][k], fminf(t.v[1][k], t.v[2][k]));
    b[3 + k] = fmaxf(t.v[0][k], fmaxf(t.v[1][k], t.v[2][k]));
  }
  float pos;
  int m;
  if(splitPlane(G, b, pos, m) < 0)
    return 0.0f;
  const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
  const float e1[3] = {t.v[1][0] - t.v[0][0], t.v[1][1] - t.v[0][1], t.v[1][2] - t.v[0][2]};
  const float e2[3] = {t.v[2][0] - t.v[0][0], t.v[2][1] - t.v[0][1], t.v[2][2] - t.v[0][2]};
  const float nx = e1[1] * e2[2] - e1[2] * e2[1], ny = e1[2] * e2[0] - e1[0] * e2[2], nz = e1[0] * e2[1] - e1[1] * e2[0];
  const float gain = 2.0f * (dx * dy + dy * dz + dz * dx) - (fabsf(nx) + fabsf(ny) + fabsf(nz));
  if(!(gain > 0.0f))
    return 0.0f;
  // areas relative to the scene's: the priorities of a scene in millimetres and of the same scene in kilometres are the same
  const float sx = G.ext[0], sy = G.ext[1], sz = G.ext[2];
  const float sceneArea = 2.0f * (sx * sy + sy * sz + sz * sx);
  if(!(sceneArea > 0.0f))
    return 0.0f;
  const float p = cbrtf(ldexpf(gain / sceneArea, m - (VKRT_SPLIT_GRID_BITS - 1)));
  return (p == p && p < INFINITY) ? p : 0.0f;
}

// boxes of the two halves of piece `b` of triangle t cut at x_axis = pos; false when one half has no extent
VKRT_DEV bool splitClip(const SplitTri& t, const float* b, int axis, float pos, float* L, float* R)
{
#pragma unroll
  for(int k = 0; k < 3; k++)
  {
    L[k] = INFINITY; L[3 + k] = -INFINITY; R[k] = INFINITY; R[3 + k] = -INFINITY;
  }
  auto grow = [](float* d, const float* p, const float* pad) {
#pragma unroll
    for(int k = 0; k < 3; k++)
    {
      d[k] = fminf(d[k], p[k] - pad[k]);
      d[3 + k] = fmaxf(d[3 + k], p[k] + pad[k]);
    }
  };
  const float zero[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
  for(int e = 0; e < 3; e++)
  {
    const float* va = t.v[e];
    const float* vb = t.v[(e + 1) % 3];
    const float da = va[axis], db = vb[axis];
    if(da <= pos) grow(L, va, zero);
    if(da >= pos) grow(R, va, zero);
    if((da < pos && db > pos) || (da > pos && db < pos))
    {
      const float w = (pos - da) / (db - da);
      float p[3], pad[3];
#pragma unroll
      for(int k = 0; k < 3; k++)
      {
        p[k] = va[k] + w * (vb[k] - va[k]);
        pad[k] = 4.76837158203125e-07f * fmaxf(fabsf(va[k]), fabsf(vb[k]));  // 8 * 2^-24 of the coordinates that went into p[k]
      }
      p[axis] = pos;
      pad[axis] = 0.0f;
      grow(L, p, pad);
      grow(R, p, pad);
    }
  }
  bool ok = true;
#pragma unroll
  for(int k = 0; k < 3; k++)
  {
    L[k] = fmaxf(L[k], b[k]); L[3 + k] = fminf(L[3 + k], b[3 + k]);
    R[k] = fmaxf(R[k], b[k]); R[3 + k] = fminf(R[3 + k], b[3 + k]);
  }
  L[3 + axis] = fminf(L[3 + axis], pos);
  R[axis] = fmaxf(R[axis], pos);
#pragma unroll
  for(int k = 0; k < 3; k++) ok = ok && L[k] <= L[3 + k] && R[k] <= R[3 + k];
  return ok;
}

// The pieces of triangle g for a budget of s splits, depth first with the smaller share first (stack depth <= log2(s) + 2).
// EMIT: writes (g, box) of every piece from slot `base` on.  Returns the number of pieces (1 .. s + 1); the same for both EMIT values
// -- they are two instantiations of one routine, and the emitting one is additionally held to the `limit` slots the counting one
// claimed for this triangle: a piece beyond them is merged into the last slot (box union) and slots left over repeat the last piece, so
// that a divergence of the two (code generation, contraction) could cost tree quality but never write into a neighbour's slots or
// leave a slot unwritten.
template <bool EMIT>
VKRT_DEV unsigned splitTriangle(const SplitGrid& G, const SplitTri& t, unsigned s, unsigned g, unsigned base, unsigned limit, unsigned* refTri, float* refBox)
{
  struct Item { float b[6]; unsigned s; };
  Item stack[VKRT_SPLIT_STACK];
  int sp = 1;
#pragma unroll
  for(int k = 0; k < 3; k++)
  {
    stack[0].b[k] = fminf(t.v[0][k], fminf(t.v[1][k], t.v[2][k]));
    stack[0].b[3 + k] = fmaxf(t.v[0][k], fmaxf(t.v[1][k], t.v[2][k]));
  }
  stack[0].s = s;
  unsigned n = 0;
  while(sp > 0)
  {
    Item it = stack[--sp];
    float pos = 0.0f;
    int m;
    int axis = -1;
    float L[6], R[6];
    bool cut = false;
    while(it.s > 0u && !cut)
    {
      axis = splitPlane(G, it.b, pos, m);
      if(axis < 0)
        break;
      if(splitClip(t, it.b, axis, pos, L, R))
        cut = true;
      else
      {
        // the triangle lies on one side of the plane inside this box: the box shrinks to that side, one split is spent
        const bool leftOk = L[0] <= L[3] && L[1] <= L[4] && L[2] <= L[5];
        const float* keep = leftOk ? L : R;
        if(!(keep[0] <= keep[3] && keep[1] <= keep[4] && keep[2] <= keep[5]))
          break;  // (cannot happen for a box that holds a piece of the triangle; keep the box as it is)
#pragma unroll
        for(int k = 0; k < 6; k++) it.b[k] = keep[k];
        it.s--;
      }
    }
    if(!cut || sp + 2 > VKRT_SPLIT_STACK)
    {
      if(EMIT)
      {
        const float pad = t.slop;
        if(n < limit)
        {
          refTri[base + n] = g;
#pragma unroll
          for(int k = 0; k < 3; k++)
          {
            refBox[6 * (size_t)(base + n) + k] = fmaxf(it.b[k] - pad, t.full[k]);
            refBox[6 * (size_t)(base + n) + 3 + k] = fminf(it.b[3 + k] + pad, t.full[3 + k]);
          }
        }
        else if(limit > 0u)
        {
          float* last = refBox + 6 * (size_t)(base + limit - 1u);
#pragma unroll
          for(int k = 0; k < 3; k++)
          {
            last[k] = fminf(last[k], fmaxf(it.b[k] - pad, t.full[k]));
            last[3 + k] = fmaxf(last[3 + k], fminf(it.b[3 + k] + pad, t.full[3 + k]));
          }
        }
      }
      n++;
      continue;
    }
    const float wa = fmaxf(L[3] - L[0], fmaxf(L[4] - L[1], L[5] - L[2])), wb = fmaxf(R[3] - R[0], fmaxf(R[4] - R[1], R[5] - R[2]));
    const unsigned rest = it.s - 1u;
    unsigned sa = (wa + wb) > 0.0f ? (unsigned)((float)rest * (wa / (wa + wb)) + 0.5f) : rest / 2u;
    sa = min(sa, rest);
    const unsigned sb = rest - sa;
    Item A, B;
#pragma unroll
    for(int k = 0; k < 6; k++) { A.b[k] = L[k]; B.b[k] = R[k]; }
    A.s = sa; B.s = sb;
    if(sa >= sb) { stack[sp++] = A; stack[sp++] = B; }  // the smaller share is taken first
    else { stack[sp++] = B; stack[sp++] = A; }
  }
  if(EMIT && n > 0u)
    for(unsigned k = min(n, limit); k < limit; k++)  // (never taken while the two instantiations agree)
    {
      refTri[base + k] = g;
      for(int c = 0; c < 6; c++) refBox[6 * (size_t)(base + k) + c] = refBox[6 * (size_t)(base + min(n, limit) - 1u) + c];
    }
  return n;
}

__global__ void k_split_priority(unsigned T, const float4* __restrict__ triU, const float* __restrict__ triBox, const unsigned* __restrict__ sceneBounds,
                                 int watertight, float* __restrict__ prio)
{
  const unsigned g = blockIdx.x * blockDim.x + threadIdx.x;
  if(g >= T)
    return;
  const SplitGrid G = loadSplitGrid(sceneBounds);
  SplitTri t;
  loadSplitTri(triU, triBox, g, watertight, t);
  // Needles (corner at v0 narrower than ~7 degrees: tri_prep.h) stay whole.  binary32 Moeller-Trumbore accepts points well outside such
  // a triangle; the slop of tri_prep.h is an ESTIMATE of that reach, and around a whole needle -- a long diagonal box -- nothing ever
  // depends on it being tight, while the boxes of a needle's pieces hug the sliver and do: campaign seed 42002111 (a 790-unit sliver,
  // 1.4e-3 rad) lost a hit that the loop over all triangles finds.  The result must stay a property of the triangle set.
  prio[g] = t.slop > 0.0f ? 0.0f : splitPriority(G, t);
}

// D with sum_t min(floor(D p_t), cap) <= budget, as large as 24 bisection steps find it (one workgroup; sums in a fixed order)
__global__ __launch_bounds__(1024) void k_split_budget(unsigned T, const float* __restrict__ prio, unsigned budget, float dmax, float* __restrict__ Dout)
{
  __shared__ double red[1024];
  __shared__ double total;
  double acc = 0.0;
  for(unsigned g = threadIdx.x; g < T; g += 1024u) acc += (double)prio[g];
  red[threadIdx.x] = acc;
  __syncthreads();
  for(unsigned off = 512u; off > 0u; off >>= 1)
  {
    if(threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if(threadIdx.x == 0) total = red[0];
  __syncthreads();
  const double sum = total;
  if(!(sum > 0.0) || budget == 0u)
  {
    if(threadIdx.x == 0) *Dout = 0.0f;
    return;
  }
  auto count = [&](float D) -> double {
    __syncthreads();
    double c = 0.0;
    for(unsigned g = threadIdx.x; g < T; g += 1024u) c += (double)min((unsigned)(D * prio[g]), VKRT_SPLIT_MAX_PER_TRI);
    red[threadIdx.x] = c;
    __syncthreads();
    for(unsigned off = 512u; off > 0u; off >>= 1)
    {
      if(threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
      __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
  };
  float lo = (float)((double)budget / sum);  // floor() only lowers the sum: within the budget
  if(count(lo) > (double)budget) lo = 0.0f;  // (float rounding of D itself)
  float hi = fmaxf(lo, 1e-30f) * 2.0f;
  for(int k = 0; k < 16 && count(hi) <= (double)budget; k++) { lo = hi; hi *= 2.0f; }
  for(int k = 0; k < 24; k++)
  {
    const float mid = 0.5f * (lo + hi);
    if(count(mid) <= (double)budget) lo = mid; else hi = mid;
  }
  if(threadIdx.x == 0) *Dout = fminf(lo, dmax);
}

template <bool EMIT>
__global__ void k_split_refs(unsigned T, const float4* __restrict__ triU, const float* __restrict__ triBox, const unsigned* __restrict__ sceneBounds, int watertight,
                             const float* __restrict__ prio, const float* __restrict__ D, unsigned* counts /*EMIT: exclusive offsets*/, unsigned* refTri,
                             float* refBox)
{
  const unsigned g = blockIdx.x * blockDim.x + threadIdx.x;
  if(g >= T)
    return;
  const SplitGrid G = loadSplitGrid(sceneBounds);
  SplitTri t;
  loadSplitTri(triU, triBox, g, watertight, t);
  const unsigned s = min((unsigned)(*D * prio[g]), VKRT_SPLIT_MAX_PER_TRI);
  if(EMIT)
    (void)splitTriangle<true>(G, t, s, g, counts[g], counts[g + 1] - counts[g], refTri, refBox);  // counts = exclusive offsets, T + 1 entries
  else
    counts[g] = splitTriangle<false>(G, t, s, g, 0u, 0u, nullptr, nullptr);
}

// ---- SAH top of the tree ------------------------------------------------------------------------------------------------
// The upper levels of a tree carry most of its SAH cost, and that is where a Morton split or a window-limited clustering is
// furthest from what the surface-area heuristic would choose.  After the binary hierarchy stands, the nodes with more than K
// triangles below them are re-built: the subtrees hanging under them (<= K triangles each, "frontier") become the primitives
// of a full-sweep SAH build (a few thousand boxes; on the host: microseconds per split, no device round trips per level), whose
// internal nodes take over the ids of the nodes they replace (a binary tree over F frontier entries has F - 1 internal nodes,
// exactly the nodes above the frontier; the root keeps id 0).  HLBVH's recipe (Garanzha, Pantaleoni, McAllister 2011,
// "Simpler and Faster HLBVH with Work Queues", section 4) applied to whatever built the bottom.
struct TopEntry  // 32 B
{
  int ref;  // >= 0 internal node, < 0 ~(leaf position)
  int count;
  float box[6];
};
struct TopNode  // 40 B
{
  int id, left, right, count;
  float box[6];
};

__global__ void k_top_select(int n, unsigned K, unsigned cap, const unsigned* __restrict__ order, const float* __restrict__ triBox,
                             const int2* __restrict__ range, const int* __restrict__ parentInternal, const int* __restrict__ parentLeaf,
                             const float* __restrict__ nodeBox, TopEntry* frontier, int* topIds, unsigned* counters)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if(k >= 2 * n - 1)
    return;
  const bool leaf = k >= n - 1;
  const int pos = k - (n - 1);
  const unsigned cnt = leaf ? 1u : (unsigned)(range[k].y - range[k].x + 1);
  if(cnt > K)
  {
    const unsigned at = atomicAdd(&counters[1], 1u);
    if(at < cap)
      topIds[at] = k;
    return;
  }
  const int parent = leaf ? parentLeaf[pos] : parentInternal[k];
  if(parent < 0 || (unsigned)(range[parent].y - range[parent].x + 1) <= K)
    return;  // inside a frontier subtree
  const unsigned at = atomicAdd(&counters[0], 1u);
  if(at >= cap)
    return;
  TopEntry e;
  e.ref = leaf ? ~pos : k;
  e.count = (int)cnt;
  const float* b = leaf ? &triBox[6 * (size_t)order[pos]] : &nodeBox[6 * (size_t)k];
#pragma unroll
  for(int q = 0; q < 6; q++) e.box[q] = b[q];
  frontier[at] = e;
}

__global__ void k_top_apply(unsigned m, const TopNode* __restrict__ top, int2* children, int2* range, int* parentInternal, int* parentLeaf, float* nodeBox)
{
  const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
  if(k >= m)
    return;
  const TopNode t = top[k];
  children[t.id] = make_int2(t.left, t.right);
  range[t.id] = make_int2(0, t.count - 1);  // (only the triangle count of a node is used from here on)
#pragma unroll
  for(int q = 0; q < 6; q++) nodeBox[6 * (size_t)t.id + q] = t.box[q];
  if(t.left >= 0) parentInternal[t.left] = t.id; else parentLeaf[~t.left] = t.id;
  if(t.right >= 0) parentInternal[t.right] = t.id; else parentLeaf[~t.right] = t.id;
  if(t.id == 0)
    parentInternal[0] = -1;
}

// full-sweep SAH over the frontier entries [a, b) of `e` (reordered in place); returns the reference of the subtree's root
struct TopBuilder
{
  std::vector<TopEntry>& e;
  const std::vector<int>& ids;  // ids for the new internal nodes; ids[0] = 0 goes to the root
  std::vector<TopNode>& out;
  size_t nextId = 0;
  std::vector<float> rightArea;
  std::vector<int> rightCount;

  static float area(const float* b)
  {
    const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    return dx * dy + dy * dz + dz * dx;
  }
  static void grow(float* b, const float* o)
  {
    for(int q = 0; q < 3; q++)
    {
      b[q] = std::min(b[q], o[q]);
      b[3 + q] = std::max(b[3 + q], o[3 + q]);
    }
  }
  // node over [a, b) split at `split` (entries already partitioned): box, count, children
  int finishNode(size_t a, size_t b, size_t split, TopNode node)
  {
    float bx[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int cnt = 0;
    for(size_t k = a; k < b; k++)
    {
      grow(bx, e[k].box);
      cnt += e[k].count;
    }
    for(int q = 0; q < 6; q++) node.box[q] = bx[q];
    node.count = cnt;
    const size_t slot = out.size();
    out.push_back(node);
    const int l = build(a, split), r = build(split, b);
    out[slot].left = l;
    out[slot].right = r;
    return node.id;
  }
  // large ranges: 32 bins per axis over the centroid bounds (O(n) per node instead of three sorts)
  int buildBinned(size_t a, size_t b, TopNode node)
  {
    const int kBins = 32;
    float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for(size_t k = a; k < b; k++)
      for(int q = 0; q < 3; q++)
      {
        const float c = e[k].box[q] + e[k].box[3 + q];
        clo[q] = std::min(clo[q], c);
        chi[q] = std::max(chi[q], c);
      }
    float best = INFINITY;
    int bestAxis = -1, bestBin = 0;
    for(int axis = 0; axis < 3; axis++)
    {
      const float ext = chi[axis] - clo[axis];
      if(!(ext > 0.0f))
        continue;
      float bb[kBins][6];
      int bc[kBins];
      for(int k = 0; k < kBins; k++)
      {
        for(int q = 0; q < 3; q++) { bb[k][q] = INFINITY; bb[k][3 + q] = -INFINITY; }
        bc[k] = 0;
      }
      const float scale = (float)kBins / ext;
      for(size_t k = a; k < b; k++)
      {
        int bin = (int)(((e[k].box[axis] + e[k].box[3 + axis]) - clo[axis]) * scale);
        bin = std::min(std::max(bin, 0), kBins - 1);
        grow(bb[bin], e[k].box);
        bc[bin] += e[k].count;
      }
      float ra[kBins];
      int rc[kBins];
      float acc[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
      int cnt = 0;
      for(int k = kBins - 1; k > 0; k--)
      {
        if(bc[k]) grow(acc, bb[k]);
        cnt += bc[k];
        ra[k] = cnt ? area(acc) : 0.0f;
        rc[k] = cnt;
      }
      float left[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
      int lc = 0;
      for(int k = 1; k < kBins; k++)
      {
        if(bc[k - 1]) grow(left, bb[k - 1]);
        lc += bc[k - 1];
        if(lc == 0 || rc[k] == 0)
          continue;
        const float c = area(left) * (float)lc + ra[k] * (float)rc[k];
        if(c < best || (c == best && bestAxis == axis && std::abs(k - kBins / 2) < std::abs(bestBin - kBins / 2)))
        {
          best = c; bestAxis = axis; bestBin = k;
        }
      }
    }
    size_t split;
    if(bestAxis < 0)
      split = a + (b - a) / 2;  // all centroids coincide: any split will do
    else
    {
      const float scale = (float)kBins / (chi[bestAxis] - clo[bestAxis]), lo = clo[bestAxis];
      const int axis = bestAxis, cut = bestBin;
      auto mid = std::stable_partition(e.begin() + a, e.begin() + b, [=](const TopEntry& x) {
        int bin = (int)(((x.box[axis] + x.box[3 + axis]) - lo) * scale);
        bin = std::min(std::max(bin, 0), kBins - 1);
        return bin < cut;
      });
      split = (size_t)(mid - e.begin());
      if(split == a || split == b)
        split = a + (b - a) / 2;
    }
    return finishNode(a, b, split, node);
  }
  int build(size_t a, size_t b)
  {
    if(b - a == 1)
      return e[a].ref;
    const size_t n = b - a;
    TopNode node;
    node.id = ids[nextId++];
    if(n > 1024)
      return buildBinned(a, b, node);
    float best = INFINITY;
    int bestAxis = 0;
    size_t bestSplit = 1;
    rightArea.resize(std::max(rightArea.size(), n + 1));
    rightCount.resize(std::max(rightCount.size(), n + 1));
    for(int axis = 0; axis < 3; axis++)
    {
      std::sort(e.begin() + a, e.begin() + b, [axis](const TopEntry& x, const TopEntry& y) {
        const float cx = x.box[axis] + x.box[3 + axis], cy = y.box[axis] + y.box[3 + axis];
        return cx < cy || (cx == cy && x.ref < y.ref);
      });
      float acc[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
      int cnt = 0;
      for(size_t k = n; k-- > 1;)  // right parts [k, n)
      {
        grow(acc, e[a + k].box);
        cnt += e[a + k].count;
        rightArea[k] = area(acc);
        rightCount[k] = cnt;
      }
      float left[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
      int lc = 0;
      for(size_t k = 1; k < n; k++)  // left part [0, k)
      {
        grow(left, e[a + k - 1].box);
        lc += e[a + k - 1].count;
        const float c = area(left) * (float)lc + rightArea[k] * (float)rightCount[k];
        // equal costs (identical boxes: a pile of coincident triangles) must not peel one entry per level: prefer the middle
        const size_t offMid = k > n / 2 ? k - n / 2 : n / 2 - k, bestOff = bestSplit > n / 2 ? bestSplit - n / 2 : n / 2 - bestSplit;
        if(c < best || (c == best && offMid < bestOff))
        {
          best = c; bestAxis = axis; bestSplit = k;
        }
      }
    }
    if(bestAxis != 2)
      std::sort(e.begin() + a, e.begin() + b, [bestAxis](const TopEntry& x, const TopEntry& y) {
        const float cx = x.box[bestAxis] + x.box[3 + bestAxis], cy = y.box[bestAxis] + y.box[3 + bestAxis];
        return cx < cy || (cx == cy && x.ref < y.ref);
      });
    return finishNode(a, b, a + bestSplit, node);
  }
};

struct Temp
{
  std::vector<void*> ptrs;
  ~Temp()
  {
    for(void* p : ptrs) (void)hipFree(p);
  }
  template <typename T>
  hipError_t alloc(T** p, size_t count)
  {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, std::max<size_t>(count * sizeof(T), 16));
    if(e == hipSuccess) ptrs.push_back(q);
    *p = (T*)q;
    return e;
  }
};

}  // namespace

#define LB_TRY(expr)                                         \
  do                                                         \
  {                                                          \
    hipError_t e_ = (expr);                                  \
    if(e_ != hipSuccess)                                     \
    {                                                        \
      out.error = std::string(#expr) + ": " + hipGetErrorString(e_); \
      if(out.nodes) (void)hipFree(out.nodes);                \
      if(out.tris) (void)hipFree(out.tris);                  \
      if(out.triShade) (void)hipFree(out.triShade);          \
      out.nodes = out.tris = out.triShade = nullptr;         \
      return e_ == hipErrorOutOfMemory ? VKRT_ERR_OUT_OF_MEMORY : VKRT_ERR_HIP; \
    }                                                        \
  } while(0)

int build_lbvh_device(const DevScene& sc, uint32_t instCount, const std::vector<vkrt_prim_mesh>& pm, const std::vector<vkrt_node>& nodes,
                      hipStream_t stream, LbvhResult& out, unsigned leafSize, bool wantWide, bool ploc, bool watertight, bool dissolve, unsigned splitPercent)
{
  bool topSah = ploc;  // the FAST_TRACE device build re-builds its upper levels with SAH; the radix tree stays the pure fast build
  if(const char* e = getenv("VKRT_TOP_SAH"))  // test hook: force on / off for either builder
    topSah = atoi(e) != 0;
  // (clustered subtrees are not runs of the Morton order, so the PLOC tree keeps one triangle per leaf)
  const unsigned kLeaf = ploc ? 1u : (leafSize < 1u ? 1u : (leafSize > 8u ? 8u : leafSize));
  out = LbvhResult{};
  std::vector<uint32_t> firstGid(instCount + 1, 0), firstIndex(instCount, 0), vertexOffset(instCount, 0);
  std::vector<int32_t> material(instCount, 0);
  for(uint32_t n = 0; n < instCount; n++)
  {
    const vkrt_prim_mesh& p = pm[nodes[n].primMesh];
    firstGid[n + 1] = firstGid[n] + p.indexCount / 3;
    firstIndex[n] = p.firstIndex;
    vertexOffset[n] = p.vertexOffset;
    material[n] = p.materialIndex;
  }
  const uint32_t triangles = firstGid[instCount];
  out.triCount = triangles;
  out.uniqueTris = triangles;
  out.rootRef = VKRT_TRAV_DONE;
  if(triangles == 0)
  {
    LB_TRY(hipMalloc(&out.nodes, 64));
    LB_TRY(hipMalloc(&out.tris, 48));
    LB_TRY(hipMalloc(&out.triShade, 16));
    return VKRT_OK;
  }

  Temp tmp;
  uint32_t *dFirstGid, *dFirstIndex, *dVertexOffset;
  int32_t* dMaterial;
  LB_TRY(tmp.alloc(&dMaterial, instCount));
  LB_TRY(hipMemcpyAsync(dMaterial, material.data(), instCount * 4, hipMemcpyHostToDevice, stream));
  LB_TRY(tmp.alloc(&dFirstGid, instCount + 1));
  LB_TRY(tmp.alloc(&dFirstIndex, instCount));
  LB_TRY(tmp.alloc(&dVertexOffset, instCount));
  LB_TRY(hipMemcpyAsync(dFirstGid, firstGid.data(), (instCount + 1) * 4, hipMemcpyHostToDevice, stream));
  LB_TRY(hipMemcpyAsync(dFirstIndex, firstIndex.data(), instCount * 4, hipMemcpyHostToDevice, stream));
  LB_TRY(hipMemcpyAsync(dVertexOffset, vertexOffset.data(), instCount * 4, hipMemcpyHostToDevice, stream));

  float4* triU;
  float* triBox;
  unsigned* bounds;
  LB_TRY(tmp.alloc(&triU, (size_t)triangles * 3));
  LB_TRY(tmp.alloc(&triBox, (size_t)triangles * 6));
  LB_TRY(tmp.alloc(&bounds, 8));
  const unsigned initB[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u};
  LB_TRY(hipMemcpyAsync(bounds, initB, sizeof initB, hipMemcpyHostToDevice, stream));

  const unsigned B = 256;
  FlatArgs A{sc.positions, sc.indices, sc.instances, dFirstGid, dFirstIndex, dVertexOffset, instCount, triangles};
  hipLaunchKernelGGL(k_flatten, dim3((triangles + B - 1) / B), dim3(B), 0, stream, A, watertight ? 1 : 0, triU, triBox, bounds);
  LB_TRY(hipGetLastError());

  // ---- triangle pre-splitting: T = references (>= triangles), refTri / refBox replace the identity / triBox ------------------
  uint32_t T = triangles;
  const unsigned* refTri = nullptr;
  const float* refBox = triBox;
  const unsigned budget = (unsigned)std::min<uint64_t>((uint64_t)triangles * splitPercent / 100u, 0x3fffffffu);
  if(budget > 0u && triangles > 1u)
  {
    const unsigned G3 = (triangles + B - 1) / B;
    float *prio, *dD;
    unsigned *counts, *offsets;
    LB_TRY(tmp.alloc(&prio, triangles));
    LB_TRY(tmp.alloc(&dD, 4));
    LB_TRY(tmp.alloc(&counts, (size_t)triangles + 1));
    LB_TRY(tmp.alloc(&offsets, (size_t)triangles + 1));
    hipLaunchKernelGGL(k_split_priority, dim3(G3), dim3(B), 0, stream, triangles, (const float4*)triU, (const float*)triBox, (const unsigned*)bounds, watertight ? 1 : 0, prio);
    float dmax = INFINITY;
    if(const char* e = getenv("VKRT_SPLIT_DMAX"))  // test hook: cap of the priority scale D (splits per unit of priority)
      dmax = (float)atof(e);
    hipLaunchKernelGGL(k_split_budget, dim3(1), dim3(1024), 0, stream, triangles, (const float*)prio, budget, dmax, dD);
    LB_TRY(hipMemsetAsync(counts + triangles, 0, 4, stream));
    hipLaunchKernelGGL((k_split_refs<false>), dim3(G3), dim3(B), 0, stream, triangles, (const float4*)triU, (const float*)triBox, (const unsigned*)bounds,
                       watertight ? 1 : 0, (const float*)prio, (const float*)dD, counts, (unsigned*)nullptr, (float*)nullptr);
    LB_TRY(hipGetLastError());
    size_t scanBytes = 0;
    LB_TRY(rocprim::exclusive_scan(nullptr, scanBytes, counts, offsets, 0u, (size_t)triangles + 1, rocprim::plus<unsigned>(), stream));
    void* scanTmp;
    LB_TRY(tmp.alloc((char**)&scanTmp, scanBytes));
    LB_TRY(rocprim::exclusive_scan(scanTmp, scanBytes, counts, offsets, 0u, (size_t)triangles + 1, rocprim::plus<unsigned>(), stream));
    unsigned total = 0;
    LB_TRY(hipMemcpyAsync(&total, offsets + triangles, 4, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipStreamSynchronize(stream));
    if(total < triangles || total > triangles + budget)
    {
      out.error = "triangle pre-splitting: reference count out of range (internal error)";
      return VKRT_ERR_HIP;
    }
    if(total > triangles)
    {
      unsigned* rt;
      float* rb;
      LB_TRY(tmp.alloc(&rt, total));
      LB_TRY(tmp.alloc(&rb, (size_t)total * 6));
      hipLaunchKernelGGL((k_split_refs<true>), dim3(G3), dim3(B), 0, stream, triangles, (const float4*)triU, (const float*)triBox, (const unsigned*)bounds,
                         watertight ? 1 : 0, (const float*)prio, (const float*)dD, offsets, rt, rb);
      LB_TRY(hipGetLastError());
      T = total;
      refTri = rt;
      refBox = rb;
    }
  }
  out.triCount = T;
  LB_TRY(hipMalloc(&out.nodes, std::max<size_t>((size_t)(T > 1 ? T - 1 : 1) * 64, 64)));
  LB_TRY(hipMalloc(&out.tris, std::max<size_t>((size_t)T * 48, 48)));
  LB_TRY(hipMalloc(&out.triShade, std::max<size_t>((size_t)T * 16, 16)));
  out.nodeCount = T > 1 ? T - 1 : 0;

  unsigned long long *keysA, *keysB;
  unsigned *valsA, *valsB;
  LB_TRY(tmp.alloc(&keysA, T));
  LB_TRY(tmp.alloc(&keysB, T));
  LB_TRY(tmp.alloc(&valsA, T));
  LB_TRY(tmp.alloc(&valsB, T));
  const unsigned G = (T + B - 1) / B;
  triBox = const_cast<float*>(refBox);  // from here on "triangle" means reference -- one leaf of the tree -- and triBox its box
  hipLaunchKernelGGL(k_morton, dim3(G), dim3(B), 0, stream, T, (const float*)triBox, (const unsigned*)bounds, keysA, valsA);
  LB_TRY(hipGetLastError());

  size_t sortBytes = 0;
  LB_TRY(rocprim::radix_sort_pairs(nullptr, sortBytes, keysA, keysB, valsA, valsB, (size_t)T, 0, 63, stream));
  void* sortTmp;
  LB_TRY(tmp.alloc((char**)&sortTmp, sortBytes));
  LB_TRY(rocprim::radix_sort_pairs(sortTmp, sortBytes, keysA, keysB, valsA, valsB, (size_t)T, 0, 63, stream));
  const unsigned* order = valsB;

  hipLaunchKernelGGL(k_pack, dim3(G), dim3(B), 0, stream, T, order, (const float4*)triU, (float4*)out.tris, A, (const int*)dMaterial, (uint4*)out.triShade,
                     dissolve ? sc.materials : (const DevMaterial*)nullptr, refTri);
  LB_TRY(hipGetLastError());

  if(T <= kLeaf)
  {
    out.rootRef = (int32_t) ~((0u << 3) | (T - 1u));
    out.maxDepth = 0;
    out.nodeCount = 0;
    LB_TRY(hipStreamSynchronize(stream));
    return VKRT_OK;
  }

  int2 *children, *range;
  int *parentInternal, *parentLeaf;
  float* nodeBox;
  unsigned* arrive;
  unsigned* scalars;  // [0] maxDepth, [1] sah accum (float)
  LB_TRY(tmp.alloc(&children, T - 1));
  LB_TRY(tmp.alloc(&range, T - 1));
  LB_TRY(tmp.alloc(&parentInternal, T - 1));
  LB_TRY(tmp.alloc(&parentLeaf, T));
  LB_TRY(tmp.alloc(&nodeBox, (size_t)(T - 1) * 6));
  LB_TRY(tmp.alloc(&arrive, T - 1));
  LB_TRY(tmp.alloc(&scalars, 4));  // [0] maxDepth, [1] sah accum, [2] frontier entries, [3] nodes above the frontier
  LB_TRY(hipMemsetAsync(arrive, 0, (size_t)(T - 1) * 4, stream));
  LB_TRY(hipMemsetAsync(scalars, 0, 16, stream));
  if(ploc)
  {
    const int rcp = ploc_cluster_device(T, order, (const float*)triBox, stream, children, range, parentInternal, parentLeaf, nodeBox, nullptr, out.error);
    if(rcp != VKRT_OK)
    {
      (void)hipFree(out.nodes); (void)hipFree(out.tris); (void)hipFree(out.triShade);
      out.nodes = out.tris = out.triShade = nullptr;
      return rcp;
    }
  }
  else
  {
    hipLaunchKernelGGL(k_hierarchy, dim3(G), dim3(B), 0, stream, (int)T, (const unsigned long long*)keysB, children, range, parentInternal,
                       parentLeaf);
    hipLaunchKernelGGL(k_fit, dim3(G), dim3(B), 0, stream, (int)T, order, (const float*)triBox, (const int2*)children,
                       (const int*)parentInternal, (const int*)parentLeaf, nodeBox, arrive);
  }
  if(topSah && T > 4096u)
  {
    // re-build the levels above subtrees of <= K triangles with a full-sweep SAH over those subtrees (see k_top_select)
    // ~2 k subtrees whatever the scene size (16 k of them: twice the build time at 2 M triangles for the same ray rate, #71)
    unsigned K = std::max(16u, T / 2048u);
    if(const char* e = getenv("VKRT_TOP_SAH_LEAF"))  // test hook: triangles per frontier subtree
      K = (unsigned)std::max(8, atoi(e));
    const unsigned cap = 32768u;
    TopEntry* dFrontier;
    int* dTopIds;
    TopNode* dTop;
    LB_TRY(tmp.alloc(&dFrontier, cap));
    LB_TRY(tmp.alloc(&dTopIds, cap));
    LB_TRY(tmp.alloc(&dTop, cap));
    LB_TRY(hipMemsetAsync(&scalars[2], 0, 8, stream));
    hipLaunchKernelGGL(k_top_select, dim3((2 * T - 1 + B - 1) / B), dim3(B), 0, stream, (int)T, K, cap, order, (const float*)triBox, (const int2*)range,
                       (const int*)parentInternal, (const int*)parentLeaf, (const float*)nodeBox, dFrontier, dTopIds, &scalars[2]);
    LB_TRY(hipGetLastError());
    unsigned cnt[2];
    LB_TRY(hipMemcpyAsync(cnt, &scalars[2], 8, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipStreamSynchronize(stream));
    if(cnt[0] >= 2u && cnt[0] <= cap && cnt[1] + 1u == cnt[0])  // (a frontier too large for the host pass keeps the tree as built)
    {
      std::vector<TopEntry> fr(cnt[0]);
      std::vector<int> ids(cnt[1]);
      LB_TRY(hipMemcpyAsync(fr.data(), dFrontier, (size_t)cnt[0] * sizeof(TopEntry), hipMemcpyDeviceToHost, stream));
      LB_TRY(hipMemcpyAsync(ids.data(), dTopIds, (size_t)cnt[1] * sizeof(int), hipMemcpyDeviceToHost, stream));
      LB_TRY(hipStreamSynchronize(stream));
      std::sort(ids.begin(), ids.end());  // the atomics hand out slots in any order: fix it (ids[0] = 0, the root)
      std::sort(fr.begin(), fr.end(), [](const TopEntry& x, const TopEntry& y) { return x.ref < y.ref; });
      std::vector<TopNode> top;
      top.reserve(cnt[1]);
      TopBuilder tb{fr, ids, top};
      (void)tb.build(0, fr.size());
      LB_TRY(hipMemcpyAsync(dTop, top.data(), top.size() * sizeof(TopNode), hipMemcpyHostToDevice, stream));
      hipLaunchKernelGGL(k_top_apply, dim3(((unsigned)top.size() + B - 1) / B), dim3(B), 0, stream, (unsigned)top.size(), (const TopNode*)dTop, children, range,
                         parentInternal, parentLeaf, nodeBox);
      LB_TRY(hipGetLastError());
      LB_TRY(hipStreamSynchronize(stream));  // `top` is read by the copy until here
    }
  }
  hipLaunchKernelGGL(k_emit, dim3(G), dim3(B), 0, stream, kLeaf, (int)T, order, (const float*)triBox, (const int2*)children, (const int2*)range,
                     (const float*)nodeBox, (float4*)out.nodes, (float*)&scalars[1]);
  hipLaunchKernelGGL(k_depth, dim3(G), dim3(B), 0, stream, kLeaf, (int)T, (const int2*)range, (const int*)parentInternal, (const int*)parentLeaf,
                     &scalars[0]);
  LB_TRY(hipGetLastError());
  if(wantWide && kLeaf == 1u)
  {
    // the whole acceleration structure on the device: collapse the radix tree into 8-wide compressed nodes while the parent
    // arrays are still here (wide_collapse.hip); a tree that does not fit its level budget is reported, not guessed at
    WideCollapseIn in{T, (const float4*)out.nodes, (const int*)parentInternal, (const int*)parentLeaf, (const float4*)out.tris, (const uint4*)out.triShade};
    const int rcw = collapse_wide8_device(in, stream, out.wide, out.error);
    if(rcw != VKRT_OK)
    {
      (void)hipFree(out.nodes); (void)hipFree(out.tris); (void)hipFree(out.triShade);
      out.nodes = out.tris = out.triShade = nullptr;
      return rcw;
    }
    out.hasWide = !out.wide.overflow;
    if(out.wide.overflow)
    {
      (void)hipFree(out.wide.nodes); (void)hipFree(out.wide.tris); (void)hipFree(out.wide.triShade);
      out.wide.nodes = out.wide.tris = out.wide.triShade = nullptr;
    }
  }
  unsigned hs[4];
  float rootBox[6];
  LB_TRY(hipMemcpyAsync(hs, scalars, 16, hipMemcpyDeviceToHost, stream));
  LB_TRY(hipMemcpyAsync(rootBox, nodeBox, 24, hipMemcpyDeviceToHost, stream));
  LB_TRY(hipStreamSynchronize(stream));
  out.rootRef = 0;
  out.maxDepth = hs[0];
  float sah;
  memcpy(&sah, &hs[1], 4);
  const float ra = 2.0f * ((rootBox[3] - rootBox[0]) * (rootBox[4] - rootBox[1]) + (rootBox[4] - rootBox[1]) * (rootBox[5] - rootBox[2]) +
                           (rootBox[5] - rootBox[2]) * (rootBox[3] - rootBox[0]));
  out.sahCost = ra > 0.0f ? sah / ra : 0.0f;
  return VKRT_OK;
}

}  // namespace vkrt
