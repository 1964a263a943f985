// ploc.hip -- binary hierarchy by parallel locally-ordered clustering (VKRT_BUILD_PLOC_GPU), the trace-quality device build.
//
// The reference asks its driver for a FAST_TRACE acceleration structure built on the device (hello_vulkan.cpp:1010, :1046).  The
// Morton radix tree of lbvh.hip is the fast-build end of that trade: it splits where the codes' bits say, not where the surface
// area says, and traces ~5 % slower than the host's binned-SAH tree on the bench scene.  This builder keeps everything on the
// device and closes that gap: agglomerative clustering restricted to a window of the Morton order (Meister & Bittner 2018,
// "Parallel Locally-Ordered Clustering for Bounding Volume Hierarchy Construction", restated from the paper's description).
//
//   clusters = the triangles in Morton order (cluster = node reference + box + triangle count)
//   repeat until one cluster is left:
//     k_ploc_nn      every cluster i picks, among the clusters within RADIUS positions of it, the one it is cheapest to merge
//                    with (boxes of a workgroup's window staged in LDS).  Cost of a merge = what it adds to the tree's SAH
//                    sum: area(union) * (n_i + n_j) - area_i * n_i - area_j * n_j (n = triangles below).  The paper's
//                    distance, area(union) alone, builds trees that trace 1 % slower here (profiles/r02_experiments.md #67)
//     k_ploc_count   mutual picks merge: the lower position of a pair becomes the new node, the higher one disappears;
//                    per-workgroup counts of surviving clusters and of merges
//     k_ploc_scan    one workgroup: exclusive scan of the per-workgroup counts
//     k_ploc_apply   writes the merged nodes (children, box, parents, triangle count) and the compacted cluster array
//
// Ties: a cluster prefers position i ^ 1, then the lowest position.  With equal distances everywhere (coincident triangles) all
// even / odd neighbours pair up in one pass; in general position the rule is irrelevant.  Some mutual pair always exists
// (an i / i ^ 1 edge among the globally closest pairs is mutual by the first rule; otherwise the lowest position among them
// and its pick choose each other), so every pass merges at least once and the loop ends.
// Node ids are handed out from T - 2 downwards, pass by pass, position by position: the last merge is node 0, the root the
// rest of the pipeline expects, and the tree is the same on every run.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <string>

#include "device_math.h"
#include "lbvh.h"

namespace vkrt {
namespace {

#define PLOC_BLOCK 256
#define PLOC_MAX_RADIUS 32

struct PlocCluster  // 32 B
{
  float lo[3];
  int ref;    // >= 0 internal node, < 0 ~(leaf position)
  float hi[3];
  int count;  // triangles below
};

__global__ void k_ploc_leaves(unsigned n, const unsigned* __restrict__ order, const float* __restrict__ triBox, PlocCluster* c)
{
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  const float* b = &triBox[6 * (size_t)order[i]];
  PlocCluster k;
  k.lo[0] = b[0]; k.lo[1] = b[1]; k.lo[2] = b[2]; k.hi[0] = b[3]; k.hi[1] = b[4]; k.hi[2] = b[5];
  k.ref = ~(int)i;
  k.count = 1;
  c[i] = k;
}

VKRT_DEV float unionArea(const float* a, const float* b)  // a, b: lo[3] hi[3]
{
  const float dx = fmaxf(a[3], b[3]) - fminf(a[0], b[0]);
  const float dy = fmaxf(a[4], b[4]) - fminf(a[1], b[1]);
  const float dz = fmaxf(a[5], b[5]) - fminf(a[2], b[2]);
  return dx * dy + dy * dz + dz * dx;
}

// metric < 0: no search -- every cluster picks position i ^ 1 (all pairs are mutual).  The pass after one that merged nothing:
// the cost above is symmetric in exact arithmetic only; cancellation (areas of 1e6 next to 1e-6, huge counts) can leave a
// window without any mutual pair, and the loop must still shrink the array.
__global__ __launch_bounds__(PLOC_BLOCK) void k_ploc_nn(int nc, int radius, int metric, const PlocCluster* __restrict__ c, int* __restrict__ nn)
{
  if(metric < 0 || metric == 98)
  {
    const int i = (int)(blockIdx.x * PLOC_BLOCK + threadIdx.x);
    if(i < nc)
      nn[i] = metric == 98 ? (i + 1 < nc ? i + 1 : -1)  // test hook (VKRT_PLOC_METRIC=98): picks that are never mutual
                           : ((i ^ 1) < nc ? (i ^ 1) : -1);
    return;
  }
  __shared__ float box[(PLOC_BLOCK + 2 * PLOC_MAX_RADIUS) * 6];
  __shared__ float own[(PLOC_BLOCK + 2 * PLOC_MAX_RADIUS) * 2];  // own area, triangle count
  const int first = (int)(blockIdx.x * PLOC_BLOCK) - radius;  // position of box[0]
  for(int k = (int)threadIdx.x; k < PLOC_BLOCK + 2 * radius; k += PLOC_BLOCK)
  {
    const int p = first + k;
    if(p >= 0 && p < nc)
    {
      const PlocCluster q = c[p];
      box[6 * k + 0] = q.lo[0]; box[6 * k + 1] = q.lo[1]; box[6 * k + 2] = q.lo[2];
      box[6 * k + 3] = q.hi[0]; box[6 * k + 4] = q.hi[1]; box[6 * k + 5] = q.hi[2];
      own[2 * k] = unionArea(&box[6 * k], &box[6 * k]);
      own[2 * k + 1] = (float)q.count;
    }
  }
  __syncthreads();
  const int i = (int)(blockIdx.x * PLOC_BLOCK + threadIdx.x);
  if(i >= nc)
    return;
  const float* mine = &box[6 * (i - first)];
  const float myCount = own[2 * (i - first) + 1];
  // (the terms that only depend on i are the same for all its candidates and left out; the full cost is symmetric in i and j,
  //  which the mutual-pair argument above needs)
  auto dist = [&](int j) {
    const float u = unionArea(mine, &box[6 * (j - first)]);
    if(metric == 0) return u;                              // Meister & Bittner: area of the union
    if(metric == 1) return u - own[2 * (j - first)];       // area increase
    return u * (myCount + own[2 * (j - first) + 1]) - own[2 * (j - first)] * own[2 * (j - first) + 1];  // SAH increase (default)
  };
  float best = INFINITY;
  int pick = -1;
  const int partner = i ^ 1;
  if(partner < nc)
  {
    best = dist(partner);
    pick = partner;
  }
  const int j0 = max(0, i - radius), j1 = min(nc - 1, i + radius);
  for(int j = j0; j <= j1; j++)
  {
    if(j == i || j == partner)
      continue;
    const float d = dist(j);
    if(d < best || pick < 0)
    {
      best = d;
      pick = j;
    }
  }
  nn[i] = pick;
}

// role of cluster i in this pass: 0 survives unchanged, 1 absorbs its partner (becomes a new node), 2 is absorbed
VKRT_DEV int plocRole(int nc, const int* __restrict__ nn, int i)
{
  const int j = nn[i];
  if(j < 0 || nn[j] != i)
    return 0;
  return i < j ? 1 : 2;
}

__global__ __launch_bounds__(PLOC_BLOCK) void k_ploc_count(int nc, const int* __restrict__ nn, uint2* __restrict__ blockCounts)
{
  const int i = (int)(blockIdx.x * PLOC_BLOCK + threadIdx.x);
  const int role = i < nc ? plocRole(nc, nn, i) : 2;
  const unsigned long long keep = __ballot(role != 2), merge = __ballot(role == 1);
  __shared__ unsigned wk[PLOC_BLOCK / 64], wm[PLOC_BLOCK / 64];
  if((threadIdx.x & 63u) == 0u)
  {
    wk[threadIdx.x >> 6] = (unsigned)__popcll(keep);
    wm[threadIdx.x >> 6] = (unsigned)__popcll(merge);
  }
  __syncthreads();
  if(threadIdx.x == 0)
  {
    unsigned k = 0, m = 0;
    for(int w = 0; w < PLOC_BLOCK / 64; w++) { k += wk[w]; m += wm[w]; }
    blockCounts[blockIdx.x] = make_uint2(k, m);
  }
}

// exclusive scan of the per-workgroup (kept, merged) counts by one workgroup; totals -> totals[0..1]
__global__ __launch_bounds__(1024) void k_ploc_scan(unsigned blocks, uint2* __restrict__ blockCounts, unsigned* __restrict__ totals)
{
  __shared__ uint2 part[1024];
  __shared__ uint2 carry;
  if(threadIdx.x == 0)
    carry = make_uint2(0u, 0u);
  __syncthreads();
  for(unsigned base = 0; base < blocks; base += 1024u)
  {
    const unsigned b = base + threadIdx.x;
    const uint2 v = b < blocks ? blockCounts[b] : make_uint2(0u, 0u);
    part[threadIdx.x] = v;
    __syncthreads();
    for(unsigned off = 1; off < 1024u; off <<= 1)  // Hillis-Steele inclusive scan
    {
      uint2 add = make_uint2(0u, 0u);
      if(threadIdx.x >= off)
        add = part[threadIdx.x - off];
      __syncthreads();
      part[threadIdx.x].x += add.x;
      part[threadIdx.x].y += add.y;
      __syncthreads();
    }
    const uint2 incl = part[threadIdx.x], c0 = carry;
    if(b < blocks)
      blockCounts[b] = make_uint2(c0.x + incl.x - v.x, c0.y + incl.y - v.y);
    __syncthreads();
    if(threadIdx.x == 1023u)
      carry = make_uint2(c0.x + incl.x, c0.y + incl.y);
    __syncthreads();
  }
  if(threadIdx.x == 0)
  {
    totals[0] = carry.x;
    totals[1] = carry.y;
  }
}

__global__ __launch_bounds__(PLOC_BLOCK) void k_ploc_apply(int nc, int nextId, const int* __restrict__ nn, const PlocCluster* __restrict__ c,
                                                           const uint2* __restrict__ blockOffsets, PlocCluster* __restrict__ out,
                                                           int2* __restrict__ children, int2* __restrict__ range, int* __restrict__ parentInternal,
                                                           int* __restrict__ parentLeaf, float* __restrict__ nodeBox)
{
  const int i = (int)(blockIdx.x * PLOC_BLOCK + threadIdx.x);
  const int role = i < nc ? plocRole(nc, nn, i) : 2;
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const unsigned long long keep = __ballot(role != 2), merge = __ballot(role == 1), below = (1ull << lane) - 1ull;
  __shared__ unsigned wk[PLOC_BLOCK / 64], wm[PLOC_BLOCK / 64];
  if(lane == 0u)
  {
    wk[wave] = (unsigned)__popcll(keep);
    wm[wave] = (unsigned)__popcll(merge);
  }
  __syncthreads();
  if(role == 2)
    return;
  unsigned kBefore = 0, mBefore = 0;
  for(unsigned w = 0; w < wave; w++) { kBefore += wk[w]; mBefore += wm[w]; }
  const uint2 off = blockOffsets[blockIdx.x];
  const unsigned pos = off.x + kBefore + (unsigned)__popcll(keep & below);
  PlocCluster a = c[i];
  if(role == 1)
  {
    const PlocCluster b = c[nn[i]];
    const int id = nextId - (int)(off.y + mBefore + (unsigned)__popcll(merge & below));
    children[id] = make_int2(a.ref, b.ref);
    if(a.ref >= 0) parentInternal[a.ref] = id; else parentLeaf[~a.ref] = id;
    if(b.ref >= 0) parentInternal[b.ref] = id; else parentLeaf[~b.ref] = id;
#pragma unroll
    for(int q = 0; q < 3; q++)
    {
      a.lo[q] = fminf(a.lo[q], b.lo[q]);
      a.hi[q] = fmaxf(a.hi[q], b.hi[q]);
      nodeBox[6 * (size_t)id + q] = a.lo[q];
      nodeBox[6 * (size_t)id + 3 + q] = a.hi[q];
    }
    a.count += b.count;
    a.ref = id;
    range[id] = make_int2(0, a.count - 1);  // (k_emit / k_depth only use the triangle count of a node)
    if(id == 0)
      parentInternal[0] = -1;
  }
  out[pos] = a;
}

struct Scratch
{
  void* p = nullptr;
  ~Scratch() { if(p) (void)hipFree(p); }
};

}  // namespace

int ploc_cluster_device(uint32_t T, const unsigned* order, const float* triBox, hipStream_t stream, int2* children, int2* range, int* parentInternal,
                        int* parentLeaf, float* nodeBox, unsigned* passes, std::string& err)
{
  int radius = 16;
  if(const char* e = getenv("VKRT_PLOC_RADIUS"))  // test hook: search window of the clustering (1..32)
    radius = std::max(1, std::min(PLOC_MAX_RADIUS, atoi(e)));
  int metric = 2;
  if(const char* e = getenv("VKRT_PLOC_METRIC"))  // test hook: merge cost (see k_ploc_nn)
    metric = atoi(e);
  const unsigned maxBlocks = (T + PLOC_BLOCK - 1) / PLOC_BLOCK;
  const size_t clusterBytes = (size_t)T * sizeof(PlocCluster);
  const size_t bytes = 2 * clusterBytes + (size_t)T * 4 + (size_t)maxBlocks * 8 + 64;
  Scratch s;
#define PLOC_TRY(expr)                                                  \
  do                                                                    \
  {                                                                     \
    hipError_t e_ = (expr);                                             \
    if(e_ != hipSuccess)                                                \
    {                                                                   \
      err = std::string(#expr) + ": " + hipGetErrorString(e_);          \
      return e_ == hipErrorOutOfMemory ? VKRT_ERR_OUT_OF_MEMORY : VKRT_ERR_HIP; \
    }                                                                   \
  } while(0)
  PLOC_TRY(hipMalloc(&s.p, bytes));
  PlocCluster* cl[2] = {(PlocCluster*)s.p, (PlocCluster*)((char*)s.p + clusterBytes)};
  int* nn = (int*)((char*)s.p + 2 * clusterBytes);
  uint2* blockCounts = (uint2*)((char*)nn + (size_t)T * 4);
  unsigned* totals = (unsigned*)((char*)blockCounts + (size_t)maxBlocks * 8);
  hipLaunchKernelGGL(k_ploc_leaves, dim3(maxBlocks), dim3(PLOC_BLOCK), 0, stream, T, order, triBox, cl[0]);
  uint32_t nc = T;
  int nextId = (int)T - 2;
  int cur = 0;
  unsigned pass = 0;
  bool forcePairs = false;
  while(nc > 1)
  {
    const unsigned blocks = (nc + PLOC_BLOCK - 1) / PLOC_BLOCK;
    hipLaunchKernelGGL(k_ploc_nn, dim3(blocks), dim3(PLOC_BLOCK), 0, stream, (int)nc, radius, forcePairs ? -1 : metric, (const PlocCluster*)cl[cur], nn);
    hipLaunchKernelGGL(k_ploc_count, dim3(blocks), dim3(PLOC_BLOCK), 0, stream, (int)nc, (const int*)nn, blockCounts);
    hipLaunchKernelGGL(k_ploc_scan, dim3(1), dim3(1024), 0, stream, blocks, blockCounts, totals);
    hipLaunchKernelGGL(k_ploc_apply, dim3(blocks), dim3(PLOC_BLOCK), 0, stream, (int)nc, nextId, (const int*)nn, (const PlocCluster*)cl[cur],
                       (const uint2*)blockCounts, cl[cur ^ 1], children, range, parentInternal, parentLeaf, nodeBox);
    PLOC_TRY(hipGetLastError());
    unsigned h[2];
    PLOC_TRY(hipMemcpyAsync(h, totals, 8, hipMemcpyDeviceToHost, stream));
    PLOC_TRY(hipStreamSynchronize(stream));
    if((h[1] == 0u && forcePairs) || h[0] != nc - h[1] || (int)h[1] > nextId + 1)
    {
      err = "PLOC pass without progress (internal error)";
      return VKRT_ERR_HIP;
    }
    forcePairs = h[1] == 0u;  // nothing merged (no mutual pair under rounding): pair neighbours in the next pass
    nc = h[0];
    nextId -= (int)h[1];
    cur ^= 1;
    pass++;
  }
#undef PLOC_TRY
  if(passes)
    *passes = pass;
  return nextId == -1 ? VKRT_OK : VKRT_ERR_HIP;
}

}  // namespace vkrt
