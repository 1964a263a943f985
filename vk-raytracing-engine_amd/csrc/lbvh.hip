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
__global__ void k_pack(unsigned n, const unsigned* order, const float4* triU, float4* outTris, FlatArgs A, const int* instMaterial,
                       uint4* outShade, const DevMaterial* materials)
{
  const unsigned s = blockIdx.x * blockDim.x + threadIdx.x;
  if(s >= n)
    return;
  const unsigned g = order[s];
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
                      hipStream_t stream, LbvhResult& out, unsigned leafSize, bool wantWide, bool ploc, bool watertight, bool dissolve)
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
  const uint32_t T = firstGid[instCount];
  out.triCount = T;
  out.rootRef = VKRT_TRAV_DONE;
  LB_TRY(hipMalloc(&out.nodes, std::max<size_t>((size_t)(T > 1 ? T - 1 : 1) * 64, 64)));
  LB_TRY(hipMalloc(&out.tris, std::max<size_t>((size_t)T * 48, 48)));
  LB_TRY(hipMalloc(&out.triShade, std::max<size_t>((size_t)T * 16, 16)));
  out.nodeCount = T > 1 ? T - 1 : 0;
  if(T == 0)
    return VKRT_OK;

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
  unsigned long long *keysA, *keysB;
  unsigned *valsA, *valsB;
  LB_TRY(tmp.alloc(&triU, (size_t)T * 3));
  LB_TRY(tmp.alloc(&triBox, (size_t)T * 6));
  LB_TRY(tmp.alloc(&bounds, 8));
  LB_TRY(tmp.alloc(&keysA, T));
  LB_TRY(tmp.alloc(&keysB, T));
  LB_TRY(tmp.alloc(&valsA, T));
  LB_TRY(tmp.alloc(&valsB, T));
  const unsigned initB[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u};
  LB_TRY(hipMemcpyAsync(bounds, initB, sizeof initB, hipMemcpyHostToDevice, stream));

  const unsigned B = 256, G = (T + B - 1) / B;
  FlatArgs A{sc.positions, sc.indices, sc.instances, dFirstGid, dFirstIndex, dVertexOffset, instCount, T};
  hipLaunchKernelGGL(k_flatten, dim3(G), dim3(B), 0, stream, A, watertight ? 1 : 0, triU, triBox, bounds);
  hipLaunchKernelGGL(k_morton, dim3(G), dim3(B), 0, stream, T, (const float*)triBox, (const unsigned*)bounds, keysA, valsA);
  LB_TRY(hipGetLastError());

  size_t sortBytes = 0;
  LB_TRY(rocprim::radix_sort_pairs(nullptr, sortBytes, keysA, keysB, valsA, valsB, (size_t)T, 0, 63, stream));
  void* sortTmp;
  LB_TRY(tmp.alloc((char**)&sortTmp, sortBytes));
  LB_TRY(rocprim::radix_sort_pairs(sortTmp, sortBytes, keysA, keysB, valsA, valsB, (size_t)T, 0, 63, stream));
  const unsigned* order = valsB;

  hipLaunchKernelGGL(k_pack, dim3(G), dim3(B), 0, stream, T, order, (const float4*)triU, (float4*)out.tris, A, (const int*)dMaterial, (uint4*)out.triShade, dissolve ? sc.materials : (const DevMaterial*)nullptr);
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
