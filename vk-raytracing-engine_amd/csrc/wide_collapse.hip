// wide_collapse.hip -- device-side collapse of the LBVH's binary radix tree into the 8-wide compressed layout the traversal
// kernels walk (bvh_host.h "wide8"), so that VKRT_BUILD_LBVH_GPU builds the whole acceleration structure on the GPU like the
// driver build it replaces (hello_vulkan.cpp:1001-1047: buildBlas / buildTlas with PREFER_FAST_TRACE run on the device).
//
// Same algorithm as the host collapse (bvh_host.cpp collapse_wide8): a dynamic programme over the binary tree gives, for
// every node n and every slot budget i = 1..7, the cheapest way C(n,i) to represent n's subtree with at most i children
// of one wide node (SAH: node visit = area, leaf = area x triangles, leaves hold <= 3 triangles); the wide tree is then read
// off top-down.  On the device:
//   k_w8_dp      bottom-up over the radix tree, one thread per triangle climbing with per-node arrival counters (the k_fit
//                pattern of lbvh.hip: agent-scope release / acquire around the counter)
//   per level    k_w8_kids (children of every wide node of the level + how many are internal / how many triangles),
//                k_w8_scan (one workgroup: exclusive prefix sums -> child bases, triangle bases, next level's extent),
//                k_w8_write (slot assignment by octant, 8-bit quantisation verified in double, node words, triangle order,
//                the next level's work items)
//   k_w8_pack    triangle + shading records in wide-tree order
// Wide nodes are numbered breadth-first, so the build is deterministic (no allocation atomics).  No host round trip inside;
// the caller reads back four scalars (node count, depth, SAH cost, overflow flag) at its final synchronisation.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "device_math.h"
#include "lbvh.h"
#include "wide_node.h"

namespace vkrt {

namespace {

constexpr float kNodeCost = 1.0f;  // same constants as bvh_host.cpp
constexpr float kPrimCost = 1.0f;  // (0.3 / 0.6 / 1.5 measured and rejected, profiles/r03_experiments.md #101)
constexpr int kMaxLevels = 32;

struct W8Dp  // 48 bytes per binary node
{
  float c[8];        // c[0] = area, c[i] = C(n,i) for i = 1..7
  uint32_t prims;
  uint32_t head;     // bit 0: C(n,1) chose the leaf form; bits 4..7: k of D(n,8)
  uint32_t splits;   // 4 bits per i = 2..7 (bits 4(i-2)..): k of D(n,i) if C(n,i) == D(n,i), 0 = "use C(n,i-1)"
  uint32_t pad;
};

struct Kid
{
  float lo[3], hi[3];
  int ref;  // BVH2 ref: >= 0 internal node, < 0 leaf code ~((pos << 3) | (count - 1))
};

VKRT_DEV float areaOf(const float* lo, const float* hi)
{
  const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
  return 2.f * (dx * dy + dy * dz + dz * dx);
}

VKRT_DEV void children2(const float4* __restrict__ nodes2, int node, Kid c[2])
{
  const float4 q0 = nodes2[4 * (size_t)node + 0], q1 = nodes2[4 * (size_t)node + 1], q2 = nodes2[4 * (size_t)node + 2], q3 = nodes2[4 * (size_t)node + 3];
  c[0].lo[0] = q0.x; c[0].lo[1] = q0.y; c[0].lo[2] = q0.z; c[0].hi[0] = q0.w; c[0].hi[1] = q1.x; c[0].hi[2] = q1.y;
  c[1].lo[0] = q1.z; c[1].lo[1] = q1.w; c[1].lo[2] = q2.x; c[1].hi[0] = q2.y; c[1].hi[1] = q2.z; c[1].hi[2] = q2.w;
  c[0].ref = __float_as_int(q3.x);
  c[1].ref = __float_as_int(q3.y);
}
VKRT_DEV uint32_t leafPrims(int ref) { return ((~(uint32_t)ref) & 7u) + 1u; }

// dp entries written by other CUs: agent-scope loads (this CU's L1 is never refreshed by their stores)
VKRT_DEV float dpC(const W8Dp* dp, int node, int i) { return __hip_atomic_load(&dp[node].c[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
VKRT_DEV uint32_t dpPrims(const W8Dp* dp, int node) { return __hip_atomic_load(&dp[node].prims, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ void k_w8_dp(int n, const float4* __restrict__ nodes2, const int* __restrict__ parentInternal, const int* __restrict__ parentLeaf, W8Dp* dp,
                        unsigned* arrive)
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
      return;  // the sibling subtree is not finished; its last thread takes this node
    __threadfence();
    Kid c[2];
    children2(nodes2, node, c);
    float cost[2][8];
    uint32_t prims = 0;
    for(int s = 0; s < 2; s++)
    {
      if(c[s].ref < 0)
      {
        const float v = areaOf(c[s].lo, c[s].hi) * (float)leafPrims(c[s].ref) * kPrimCost;
        for(int i = 1; i <= 7; i++) cost[s][i] = v;
        prims += leafPrims(c[s].ref);
      }
      else
      {
        for(int i = 1; i <= 7; i++) cost[s][i] = dpC(dp, c[s].ref, i);
        prims += dpPrims(dp, c[s].ref);
      }
    }
    float lo[3], hi[3];
    for(int q = 0; q < 3; q++) { lo[q] = fminf(c[0].lo[q], c[1].lo[q]); hi[q] = fmaxf(c[0].hi[q], c[1].hi[q]); }
    const float area = areaOf(lo, hi);
    auto distribute = [&](int j, uint32_t& bestK) {
      float best = INFINITY;
      bestK = 1;
      for(int kk = 1; kk < j; kk++)
      {
        const int kl = kk < 7 ? kk : 7, kr = (j - kk) < 7 ? (j - kk) : 7;
        const float v = cost[0][kl] + cost[1][kr];
        if(v < best) { best = v; bestK = (uint32_t)kk; }
      }
      return best;
    };
    W8Dp e;
    uint32_t split8;
    const float d8 = distribute(8, split8);
    const float cInternal = area * kNodeCost + d8;
    const float cLeaf = prims <= 3u ? area * (float)prims * kPrimCost : INFINITY;
    const bool asLeaf = cLeaf <= cInternal;
    e.c[0] = area;
    e.c[1] = asLeaf ? cLeaf : cInternal;
    e.prims = prims;
    e.head = (asLeaf ? 1u : 0u) | (split8 << 4);
    e.splits = 0;
    e.pad = 0;
    for(int i = 2; i <= 7; i++)
    {
      uint32_t kk;
      const float dI = distribute(i, kk);
      if(dI < e.c[i - 1]) { e.c[i] = dI; e.splits |= kk << (4 * (i - 2)); }
      else e.c[i] = e.c[i - 1];
    }
    for(int i = 0; i < 8; i++) __hip_atomic_store(&dp[node].c[i], e.c[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&dp[node].prims, e.prims, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&dp[node].head, e.head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&dp[node].splits, e.splits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    node = parentInternal[node];
  }
}

// ---- top-down emission ---------------------------------------------------------------------------------------------
struct Emit
{
  const float4* nodes2;
  const W8Dp* dp;
  int* item;          // binary root of wide node i (breadth-first numbering)
  Kid* kids;          // 8 per wide node
  uint32_t* kidInfo;  // per wide node: count | leafMask << 8
  uint32_t* cntI;     // internal children per wide node
  uint32_t* cntT;     // triangles in leaf children per wide node
  uint32_t* baseI;    // index of the first internal child's wide node
  uint32_t* baseT;    // first slot in the triangle order
  uint32_t* level;    // [0] start of the current level, [1] its node count, [2] triangles emitted so far, [3] levels done,
                      // [4] overflow flag; [8 + 2 L], [9 + 2 L]: start / count of level L (for the caller's statistics)
  uint32_t* triOrder; // wide-tree slot -> position in the LBVH's sorted triangle arrays
  uint4* outNodes;    // VKRT_WNODE_QUADS x uint4 per wide node
  float* nodeCost;    // SAH contribution per wide node
  uint32_t capacity;  // wide nodes the arrays can hold
  uint32_t triCount;
};

VKRT_DEV bool dpIsLeaf(const W8Dp* dp, int ref) { return ref < 0 || (dp[ref].head & 1u) != 0u; }

// children of a wide node: subtree c gets i slots (host: W8Ctx::gather)
VKRT_DEV void gatherKids(const Emit& E, const Kid& c0, int i0, const Kid& c1, int i1, Kid* out, int& n)
{
  Kid stk[10];
  int bud[10];
  int sp = 0;
  stk[sp] = c1; bud[sp++] = i1;
  stk[sp] = c0; bud[sp++] = i0;
  while(sp > 0)
  {
    sp--;
    const Kid c = stk[sp];
    int i = bud[sp];
    if(n >= 8) break;  // (budgets sum to <= 8; belt and braces)
    if(c.ref < 0) { out[n++] = c; continue; }
    const W8Dp& e = E.dp[c.ref];
    while(i > 1 && ((e.splits >> (4 * (i - 2))) & 15u) == 0u) i--;
    if(i == 1) { out[n++] = c; continue; }
    Kid ch[2];
    children2(E.nodes2, c.ref, ch);
    const int k = (int)((e.splits >> (4 * (i - 2))) & 15u);
    stk[sp] = ch[1]; bud[sp++] = (i - k) < 7 ? (i - k) : 7;
    stk[sp] = ch[0]; bud[sp++] = k < 7 ? k : 7;
  }
}

__global__ void k_w8_kids(Emit E)
{
  const uint32_t start = E.level[0], count = E.level[1];
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if(t >= count)
    return;
  const uint32_t me = start + t;
  const int root = E.item[me];
  Kid kids[8];
  int n = 0;
  Kid c[2];
  children2(E.nodes2, root, c);
  if(me == 0 && (E.dp[root].head & 1u))
  {  // whole scene <= 3 triangles: a root with one leaf child covering the binary root
    Kid r;
    for(int q = 0; q < 3; q++) { r.lo[q] = fminf(c[0].lo[q], c[1].lo[q]); r.hi[q] = fmaxf(c[0].hi[q], c[1].hi[q]); }
    r.ref = root;
    kids[n++] = r;
  }
  else
  {
    const int ks = (int)((E.dp[root].head >> 4) & 15u);
    gatherKids(E, c[0], ks < 7 ? ks : 7, c[1], (8 - ks) < 7 ? (8 - ks) : 7, kids, n);
  }
  uint32_t leafMask = 0, nI = 0, nT = 0;
  for(int k = 0; k < n; k++)
  {
    E.kids[(size_t)me * 8 + k] = kids[k];
    if(dpIsLeaf(E.dp, kids[k].ref))
    {
      leafMask |= 1u << k;
      nT += kids[k].ref < 0 ? leafPrims(kids[k].ref) : E.dp[kids[k].ref].prims;
    }
    else
      nI++;
  }
  E.kidInfo[me] = (uint32_t)n | (leafMask << 8);
  E.cntI[me] = nI;
  E.cntT[me] = nT;
}

// one workgroup: exclusive prefix sums over the level, next level's extent
__global__ __launch_bounds__(1024) void k_w8_scan(Emit E, int lvl)
{
  __shared__ uint32_t sI[1024], sT[1024];
  const uint32_t start = E.level[0], count = E.level[1], triBase0 = E.level[2];
  const uint32_t per = (count + 1023u) / 1024u, a = threadIdx.x * per, b = min(count, a + per);
  uint32_t tI = 0, tT = 0;
  for(uint32_t k = a; k < b; k++) { tI += E.cntI[start + k]; tT += E.cntT[start + k]; }
  sI[threadIdx.x] = tI; sT[threadIdx.x] = tT;
  __syncthreads();
  for(uint32_t off = 1; off < 1024; off <<= 1)
  {
    uint32_t vI = 0, vT = 0;
    if(threadIdx.x >= off) { vI = sI[threadIdx.x - off]; vT = sT[threadIdx.x - off]; }
    __syncthreads();
    sI[threadIdx.x] += vI; sT[threadIdx.x] += vT;
    __syncthreads();
  }
  uint32_t pI = sI[threadIdx.x] - tI, pT = sT[threadIdx.x] - tT;  // exclusive prefix of this thread's chunk
  const uint32_t nextStart = start + count;
  for(uint32_t k = a; k < b; k++)
  {
    E.baseI[start + k] = nextStart + pI;
    E.baseT[start + k] = triBase0 + pT;
    pI += E.cntI[start + k];
    pT += E.cntT[start + k];
  }
  __syncthreads();
  if(threadIdx.x == 0)
  {
    const uint32_t nextCount = sI[1023];
    E.level[8 + 2 * lvl] = start;
    E.level[9 + 2 * lvl] = count;
    if(count > 0) E.level[3] = (uint32_t)lvl + 1u;
    if(nextStart + nextCount > E.capacity || (lvl == kMaxLevels - 1 && nextCount > 0))
    {
      E.level[4] = 1u;  // does not fit / deeper than kMaxLevels: the caller falls back to the host collapse
      E.level[1] = 0u;
    }
    else
    {
      E.level[0] = nextStart;
      E.level[1] = nextCount;
    }
    E.level[2] = triBase0 + sT[1023];
  }
}

VKRT_DEV void collectTris(const Emit& E, const Kid& c, uint32_t* pos, int& n)
{
  int stk[8];
  int sp = 0;
  stk[sp++] = c.ref;
  while(sp > 0)
  {
    const int ref = stk[--sp];
    if(ref < 0)
    {
      const uint32_t code = ~(uint32_t)ref, first = code >> 3, cnt = (code & 7u) + 1u;
      for(uint32_t k = 0; k < cnt && n < 3; k++) pos[n++] = first + k;
      continue;
    }
    Kid ch[2];
    children2(E.nodes2, ref, ch);
    if(sp + 2 <= 8) { stk[sp++] = ch[1].ref; stk[sp++] = ch[0].ref; }
  }
}

__global__ void k_w8_write(Emit E, uint32_t lvl)
{
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  // (level[0] / level[1] already describe the NEXT level here; this level's extent is kept in level[8 + 2 L])
  const uint32_t start = E.level[8 + 2 * lvl], count = E.level[9 + 2 * lvl];
  if(t >= count)
    return;
  const uint32_t me = start + t;
  const uint32_t info = E.kidInfo[me];
  const int n = (int)(info & 0xffu);
  const uint32_t leafMask = info >> 8;
  Kid kids[8];
  for(int k = 0; k < n; k++) kids[k] = E.kids[(size_t)me * 8 + k];
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for(int k = 0; k < n; k++)
    for(int q = 0; q < 3; q++) { lo[q] = fminf(lo[q], kids[k].lo[q]); hi[q] = fmaxf(hi[q], kids[k].hi[q]); }
  // octant-ordered slots: a child on the + side of axis q wants a slot with bit q set (greedy assignment, as on the host)
  int slotOf[8], childAt[8];
  for(int k = 0; k < 8; k++) { slotOf[k] = -1; childAt[k] = -1; }
  const float cen[3] = {0.5f * (lo[0] + hi[0]), 0.5f * (lo[1] + hi[1]), 0.5f * (lo[2] + hi[2])};
  for(int round = 0; round < n; round++)
  {
    float bestC = -INFINITY;
    int bc = -1, bs = -1;
    for(int c = 0; c < n; c++)
    {
      if(slotOf[c] >= 0) continue;
      const float v[3] = {0.5f * (kids[c].lo[0] + kids[c].hi[0]) - cen[0], 0.5f * (kids[c].lo[1] + kids[c].hi[1]) - cen[1],
                          0.5f * (kids[c].lo[2] + kids[c].hi[2]) - cen[2]};
      for(int s = 0; s < 8; s++)
      {
        if(childAt[s] >= 0) continue;
        const float cost = ((s & 1) ? v[0] : -v[0]) + ((s & 2) ? v[1] : -v[1]) + ((s & 4) ? v[2] : -v[2]);
        if(cost > bestC) { bestC = cost; bc = c; bs = s; }
      }
    }
    if(bc < 0)
    {  // every comparison failed (NaN / inf boxes from non-finite geometry): keep the assignment total -- first free child, first free slot
      for(int c = 0; c < n && bc < 0; c++) if(slotOf[c] < 0) bc = c;
      for(int s = 0; s < 8 && bs < 0; s++) if(childAt[s] < 0) bs = s;
    }
    slotOf[bc] = bs;
    childAt[bs] = bc;
  }
  // grid: origin = lo, per-axis power-of-two cell so that the extent fits QMAX cells
  const int QMAX = VKRT_WNODE_QMAX;
  uint32_t eb[3];
  for(int q = 0; q < 3; q++)
  {
    const double ext = (double)hi[q] - (double)lo[q];
    int e = -126;
    if(ext > 0)
    {
      int ex;
      const double m = frexp(ext / (double)QMAX, &ex);  // ext / QMAX = m 2^ex, m in [0.5, 1): ceil(log2) = ex, or ex - 1 for an exact power of two
      e = m == 0.5 ? ex - 1 : ex;
    }
    e = e < -126 ? -126 : (e > 126 ? 126 : e);
    for(;;)
    {  // make sure every child's hi really fits (ceil may need one more cell)
      const double sc = ldexp(1.0, e);
      bool ok = true;
      for(int k = 0; k < n; k++)
        if(ceil(((double)kids[k].hi[q] - (double)lo[q]) / sc) > (double)QMAX) ok = false;
      if(ok || e >= 126) break;
      e++;
    }
    eb[q] = (uint32_t)(e + 127);
  }
  uint32_t imask = 0, meta[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint16_t qlo[3][8], qhi[3][8];
  for(int q = 0; q < 3; q++)
    for(int s = 0; s < 8; s++) { qlo[q][s] = 0; qhi[q][s] = 0; }
  const uint32_t triBase = E.baseT[me], childBase = E.baseI[me];
  uint32_t triOff = 0, nInternal = 0;
  double sah = (double)areaOf(lo, hi) * kNodeCost;
  for(int s = 0; s < 8; s++)
  {
    const int c = childAt[s];
    if(c < 0)
      continue;
    const Kid& ch = kids[c];
    for(int q = 0; q < 3; q++)
    {
      const double sc = ldexp(1.0, (int)eb[q] - 127), o = (double)lo[q];
      int ql = (int)floor(((double)ch.lo[q] - o) / sc);
      ql = ql < 0 ? 0 : (ql > QMAX ? QMAX : ql);
      while(ql > 0 && o + ql * sc > (double)ch.lo[q]) ql--;
      int qh = (int)ceil(((double)ch.hi[q] - o) / sc);
      qh = qh < 0 ? 0 : (qh > QMAX ? QMAX : qh);
      while(qh < QMAX && o + qh * sc < (double)ch.hi[q]) qh++;
      qlo[q][s] = (uint16_t)ql;
      qhi[q][s] = (uint16_t)qh;
    }
    if(!((leafMask >> c) & 1u))
    {
      imask |= 1u << s;
      meta[s] = 0x20u | (24u + (uint32_t)s);
      // internal children are stored consecutively from childBase in SLOT order (the traversal indexes them by popcount of imask)
      nInternal++;
    }
    else
    {
      uint32_t pos[3];
      int cnt = 0;
      collectTris(E, ch, pos, cnt);
      for(int k = 0; k < cnt; k++)
        if(triBase + triOff + (uint32_t)k < E.triCount) E.triOrder[triBase + triOff + (uint32_t)k] = pos[k];
      meta[s] = (((1u << cnt) - 1u) << 5) | triOff;
      triOff += (uint32_t)cnt;
      sah += (double)areaOf(ch.lo, ch.hi) * cnt * kPrimCost;
    }
  }
  // work items of the next level, in slot order
  uint32_t k2 = 0;
  for(int s = 0; s < 8; s++)
    if((imask >> s) & 1u)
    {
      if(childBase + k2 < E.capacity) E.item[childBase + k2] = kids[childAt[s]].ref;  // (out of range only after an overflow was flagged)
      k2++;
    }
  (void)nInternal;
  uint32_t w[VKRT_WNODE_DWORDS];
  w[0] = __float_as_uint(lo[0]); w[1] = __float_as_uint(lo[1]); w[2] = __float_as_uint(lo[2]); w[3] = eb[0] | (eb[1] << 8) | (eb[2] << 16) | (imask << 24);
  w[4] = childBase; w[5] = triBase;
  w[6] = meta[0] | (meta[1] << 8) | (meta[2] << 16) | (meta[3] << 24);
  w[7] = meta[4] | (meta[5] << 8) | (meta[6] << 16) | (meta[7] << 24);
  vkrt_wnode_store_planes(w, qlo, qhi);
  uint4* nd = E.outNodes + (size_t)me * VKRT_WNODE_QUADS;
  for(int k = 0; k < VKRT_WNODE_QUADS; k++) nd[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
  E.nodeCost[me] = (float)sah;
}

// statistics for the caller: [0] wide node count, [1] depth (levels - 1), [2] SAH cost as float bits, [3] overflow flag
__global__ __launch_bounds__(1024) void k_w8_finish(Emit E, const float4* __restrict__ nodes2, uint32_t* stats)
{
  __shared__ double part[1024];
  const uint32_t total = E.level[0];  // start of the (empty) level after the last one = number of wide nodes
  double s = 0;
  for(uint32_t k = threadIdx.x; k < total; k += 1024) s += (double)E.nodeCost[k];
  part[threadIdx.x] = s;
  __syncthreads();
  for(uint32_t off = 512; off > 0; off >>= 1)
  {
    if(threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
    __syncthreads();
  }
  if(threadIdx.x == 0)
  {
    Kid c[2];
    children2(nodes2, 0, c);
    float lo[3], hi[3];
    for(int q = 0; q < 3; q++) { lo[q] = fminf(c[0].lo[q], c[1].lo[q]); hi[q] = fmaxf(c[0].hi[q], c[1].hi[q]); }
    const float ra = fmaxf(areaOf(lo, hi), 1e-30f);
    stats[0] = total;
    stats[1] = E.level[3] > 0 ? E.level[3] - 1u : 0u;
    stats[2] = __float_as_uint((float)(part[0] / (double)ra));
    stats[3] = E.level[4];
  }
}

__global__ void k_w8_pack(uint32_t n, const uint32_t* __restrict__ triOrder, const float4* __restrict__ tris, const uint4* __restrict__ shade,
                          float4* outTris, uint4* outShade)
{
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if(s >= n)
    return;
  const uint32_t p = triOrder[s];
  outTris[3 * (size_t)s + 0] = tris[3 * (size_t)p + 0];
  outTris[3 * (size_t)s + 1] = tris[3 * (size_t)p + 1];
  outTris[3 * (size_t)s + 2] = tris[3 * (size_t)p + 2];
  outShade[s] = shade[p];
}

}  // namespace

#define WC_TRY(expr)                                                \
  do                                                                \
  {                                                                 \
    hipError_t e_ = (expr);                                         \
    if(e_ != hipSuccess)                                            \
    {                                                               \
      err = std::string(#expr) + ": " + hipGetErrorString(e_);      \
      for(void* p_ : tmp) (void)hipFree(p_);                        \
      if(out.nodes) (void)hipFree(out.nodes);                       \
      if(out.tris) (void)hipFree(out.tris);                         \
      if(out.triShade) (void)hipFree(out.triShade);                 \
      out.nodes = out.tris = out.triShade = nullptr;                \
      return e_ == hipErrorOutOfMemory ? VKRT_ERR_OUT_OF_MEMORY : VKRT_ERR_HIP; \
    }                                                               \
  } while(0)

int collapse_wide8_device(const WideCollapseIn& in, hipStream_t stream, WideCollapseOut& out, std::string& err)
{
  out = WideCollapseOut{};
  std::vector<void*> tmp;
  auto alloc = [&](void** p, size_t bytes) {
    hipError_t e = hipMalloc(p, std::max<size_t>(bytes, 16));
    if(e == hipSuccess) tmp.push_back(*p);
    return e;
  };
  const uint32_t T = in.triCount;
  if(T < 2)
  {
    err = "collapse_wide8_device needs a binary tree (>= 2 triangles)";
    return VKRT_ERR_INVALID_ARGUMENT;
  }
  const uint32_t cap = T - 1;  // every wide node is rooted at a distinct binary internal node
  Emit E{};
  W8Dp* dp;
  unsigned* arrive;
  uint32_t* stats;
  WC_TRY(alloc((void**)&dp, (size_t)cap * sizeof(W8Dp)));
  WC_TRY(alloc((void**)&arrive, (size_t)cap * 4));
  WC_TRY(alloc((void**)&E.item, (size_t)cap * 4));
  WC_TRY(alloc((void**)&E.kids, (size_t)cap * 8 * sizeof(Kid)));
  WC_TRY(alloc((void**)&E.kidInfo, (size_t)cap * 4));
  WC_TRY(alloc((void**)&E.cntI, (size_t)cap * 4));
  WC_TRY(alloc((void**)&E.cntT, (size_t)cap * 4));
  WC_TRY(alloc((void**)&E.baseI, (size_t)cap * 4));
  WC_TRY(alloc((void**)&E.baseT, (size_t)cap * 4));
  WC_TRY(alloc((void**)&E.level, (8 + 2 * kMaxLevels) * 4));
  WC_TRY(alloc((void**)&E.triOrder, (size_t)T * 4));
  WC_TRY(alloc((void**)&E.nodeCost, (size_t)cap * 4));
  WC_TRY(alloc((void**)&stats, 16));
  WC_TRY(hipMalloc(&out.nodes, std::max<size_t>((size_t)cap * VKRT_WNODE_BYTES, VKRT_WNODE_MIN_ALLOC)));
  WC_TRY(hipMalloc(&out.tris, (size_t)T * 48));
  WC_TRY(hipMalloc(&out.triShade, (size_t)T * 16));
  E.nodes2 = in.nodes2;
  E.dp = dp;
  E.outNodes = (uint4*)out.nodes;
  E.capacity = cap;
  E.triCount = T;
  WC_TRY(hipMemsetAsync(arrive, 0, (size_t)cap * 4, stream));
  WC_TRY(hipMemsetAsync(E.level, 0, (8 + 2 * kMaxLevels) * 4, stream));
  WC_TRY(hipMemsetAsync(E.item, 0, 4, stream));  // wide node 0 is rooted at binary node 0
  const uint32_t one = 1;
  WC_TRY(hipMemcpyAsync(&E.level[1], &one, 4, hipMemcpyHostToDevice, stream));
  const unsigned B = 256;
  hipLaunchKernelGGL(k_w8_dp, dim3((T + B - 1) / B), dim3(B), 0, stream, (int)T, in.nodes2, in.parentInternal, in.parentLeaf, dp, arrive);
  uint64_t width = 1;
  for(int lvl = 0; lvl < kMaxLevels; lvl++)
  {
    const uint32_t bound = (uint32_t)std::min<uint64_t>(width, cap);  // a level holds at most 8^L nodes
    const dim3 g((bound + B - 1) / B);
    hipLaunchKernelGGL(k_w8_kids, g, dim3(B), 0, stream, E);
    hipLaunchKernelGGL(k_w8_scan, dim3(1), dim3(1024), 0, stream, E, lvl);
    hipLaunchKernelGGL(k_w8_write, g, dim3(B), 0, stream, E, (uint32_t)lvl);
    width = std::min<uint64_t>(width * 8, cap);
  }
  hipLaunchKernelGGL(k_w8_finish, dim3(1), dim3(1024), 0, stream, E, in.nodes2, stats);
  hipLaunchKernelGGL(k_w8_pack, dim3((T + B - 1) / B), dim3(B), 0, stream, T, (const uint32_t*)E.triOrder, in.tris, in.triShade, (float4*)out.tris,
                     (uint4*)out.triShade);
  WC_TRY(hipGetLastError());
  uint32_t hs[4];
  WC_TRY(hipMemcpyAsync(hs, stats, 16, hipMemcpyDeviceToHost, stream));  // four scalars, not the tree
  WC_TRY(hipStreamSynchronize(stream));
  for(void* p : tmp) (void)hipFree(p);
  out.nodeCount = hs[0];
  out.maxDepth = hs[1];
  memcpy(&out.sahCost, &hs[2], 4);
  out.overflow = hs[3] != 0u;
  return VKRT_OK;
}

}  // namespace vkrt
