// bvh_host.cpp -- host-side binned-SAH BVH2 builder (VKRT_BUILD_SAH_HOST).
//
// Replaces the driver's PREFER_FAST_TRACE acceleration-structure build that the reference requests
// through nvvk::RaytracingBuilderKHR (hello_vulkan.cpp:1001-1011 buildBlas, :1031-1047 buildTlas).
// All TLAS instances are flattened to world space first (bvh_host.h: flatten_instances), so one
// BVH covers the whole scene.  Output is the 64-byte node / 48-byte triangle layout of
// device_scene.h, nodes in depth-first order (a node's first child is usually the next line).
#include "bvh_host.h"
#include "tri_prep.h"
#include "wide_node.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace vkrt {

namespace {

struct Box
{
  float lo[3], hi[3];
  void reset()
  {
    for(int k = 0; k < 3; k++) { lo[k] = INFINITY; hi[k] = -INFINITY; }
  }
  void grow(const Box& b)
  {
    for(int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); }
  }
  void growPt(const float* p)
  {
    for(int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); }
  }
  float area() const
  {
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if(!(dx >= 0.f) || !(dy >= 0.f) || !(dz >= 0.f))
      return 0.f;
    return 2.f * (dx * dy + dy * dz + dz * dx);
  }
};

constexpr int kBins = 32;
constexpr float kTraversalCost = 1.0f;
constexpr float kIntersectCost = 1.0f;
constexpr uint32_t kMaxDepth = 60;

struct Ctx
{
  const std::vector<FlatTri>& tris;
  std::vector<Box> tb;
  std::vector<float> cen;  // 3 per tri
  std::vector<uint32_t> order;
  BuiltBvh& out;
  uint32_t maxLeaf;
  double sah = 0;

  Ctx(const std::vector<FlatTri>& t, BuiltBvh& o, uint32_t ml) : tris(t), out(o), maxLeaf(ml) {}

  int32_t emitLeaf(uint32_t first, uint32_t count)
  {
    const uint32_t slot0 = (uint32_t)out.triOrder.size();
    for(uint32_t k = 0; k < count; k++)
      out.triOrder.push_back(order[first + k]);
    const uint32_t code = (slot0 << 3) | (count - 1u);
    return (int32_t)~code;
  }

  // returns child ref; cost = SAH cost of the subtree relative to its own box
  int32_t build(uint32_t first, uint32_t count, const Box& box, uint32_t depth, double& cost)
  {
    out.maxDepth = std::max(out.maxDepth, depth);
    if(count <= 1u || (count <= maxLeaf && depth >= kMaxDepth))
    {
      cost = kIntersectCost * count;
      return emitLeaf(first, count);
    }
    // centroid bounds
    Box cb;
    cb.reset();
    for(uint32_t k = 0; k < count; k++)
      cb.growPt(&cen[3 * order[first + k]]);
    const float parentArea = box.area();
    float bestCost = INFINITY;
    int bestAxis = -1, bestBin = -1;
    if(depth < kMaxDepth)
    {
      for(int axis = 0; axis < 3; axis++)
      {
        const float ext = cb.hi[axis] - cb.lo[axis];
        if(!(ext > 0.f))
          continue;
        Box bins[kBins];
        uint32_t cnt[kBins];
        for(int b = 0; b < kBins; b++) { bins[b].reset(); cnt[b] = 0; }
        const float scale = (float)kBins / ext;
        for(uint32_t k = 0; k < count; k++)
        {
          const uint32_t t = order[first + k];
          int b = (int)((cen[3 * t + axis] - cb.lo[axis]) * scale);
          b = std::min(std::max(b, 0), kBins - 1);
          bins[b].grow(tb[t]);
          cnt[b]++;
        }
        float rightArea[kBins];
        uint32_t rightCnt[kBins];
        Box acc;
        acc.reset();
        uint32_t c = 0;
        for(int b = kBins - 1; b > 0; b--)
        {
          acc.grow(bins[b]);
          c += cnt[b];
          rightArea[b] = acc.area();
          rightCnt[b] = c;
        }
        acc.reset();
        c = 0;
        for(int b = 1; b < kBins; b++)
        {
          acc.grow(bins[b - 1]);
          c += cnt[b - 1];
          if(c == 0 || rightCnt[b] == 0)
            continue;
          const float cst = acc.area() * (float)c + rightArea[b] * (float)rightCnt[b];
          if(cst < bestCost) { bestCost = cst; bestAxis = axis; bestBin = b; }
        }
      }
    }
    const float leafCost = kIntersectCost * (float)count;
    const float splitCost = (bestAxis >= 0 && parentArea > 0.f) ? kTraversalCost + kIntersectCost * bestCost / parentArea : INFINITY;
    if(count <= maxLeaf && !(splitCost < leafCost))
    {
      cost = leafCost;
      return emitLeaf(first, count);
    }
    uint32_t mid;
    uint32_t* ord = &order[first];
    if(bestAxis >= 0)
    {
      const float ext = cb.hi[bestAxis] - cb.lo[bestAxis];
      const float scale = (float)kBins / ext;
      const float lo = cb.lo[bestAxis];
      const int axis = bestAxis, bin = bestBin;
      uint32_t* m = std::partition(ord, ord + count, [&](uint32_t t) {
        int b = (int)((cen[3 * t + axis] - lo) * scale);
        b = std::min(std::max(b, 0), kBins - 1);
        return b < bin;
      });
      mid = (uint32_t)(m - ord);
    }
    else
      mid = 0;
    if(mid == 0 || mid == count)
    {  // degenerate (coincident centroids / depth cap): object median on the widest axis
      int axis = 0;
      for(int k = 1; k < 3; k++)
        if(cb.hi[k] - cb.lo[k] > cb.hi[axis] - cb.lo[axis]) axis = k;
      mid = count / 2;
      std::nth_element(ord, ord + mid, ord + count, [&](uint32_t a, uint32_t b) {
        const float ca = cen[3 * a + axis], cb2 = cen[3 * b + axis];
        return ca < cb2 || (ca == cb2 && a < b);
      });
    }
    Box b0, b1;
    b0.reset(); b1.reset();
    for(uint32_t k = 0; k < mid; k++) b0.grow(tb[ord[k]]);
    for(uint32_t k = mid; k < count; k++) b1.grow(tb[ord[k]]);

    const uint32_t me = (uint32_t)(out.nodes.size() / 16);
    out.nodes.resize(out.nodes.size() + 16);
    double c0 = 0, c1 = 0;
    const int32_t ch0 = build(first, mid, b0, depth + 1, c0);
    const int32_t ch1 = build(first + mid, count - mid, b1, depth + 1, c1);
    float* n = &out.nodes[(size_t)me * 16];
    n[0] = b0.lo[0]; n[1] = b0.lo[1]; n[2] = b0.lo[2]; n[3] = b0.hi[0];
    n[4] = b0.hi[1]; n[5] = b0.hi[2]; n[6] = b1.lo[0]; n[7] = b1.lo[1];
    n[8] = b1.lo[2]; n[9] = b1.hi[0]; n[10] = b1.hi[1]; n[11] = b1.hi[2];
    memcpy(&n[12], &ch0, 4);
    memcpy(&n[13], &ch1, 4);
    n[14] = 0.f; n[15] = 0.f;
    const double pa = parentArea > 0.f ? parentArea : 1.0;
    cost = kTraversalCost + (b0.area() * c0 + b1.area() * c1) / pa;
    return (int32_t)me;
  }
};

}  // namespace

void invert3x3_rows(const float o2w[12], float w2o[9])
{
  // cofactor inverse of the upper-left 3x3 in double, fixed operation order
  const double a = o2w[0], b = o2w[1], c = o2w[2];
  const double d = o2w[4], e = o2w[5], f = o2w[6];
  const double g = o2w[8], h = o2w[9], i = o2w[10];
  const double A = e * i - f * h;
  const double B = -(d * i - f * g);
  const double C = d * h - e * g;
  const double det = a * A + b * B + c * C;
  const double inv = 1.0 / det;
  w2o[0] = (float)(A * inv);
  w2o[1] = (float)(-(b * i - c * h) * inv);
  w2o[2] = (float)((b * f - c * e) * inv);
  w2o[3] = (float)(B * inv);
  w2o[4] = (float)((a * i - c * g) * inv);
  w2o[5] = (float)(-(a * f - c * d) * inv);
  w2o[6] = (float)(C * inv);
  w2o[7] = (float)(-(a * h - b * g) * inv);
  w2o[8] = (float)((a * e - b * d) * inv);
}

static inline void xformPoint(const float m[12], const float* p, float* r)
{
  r[0] = ((m[0] * p[0] + m[1] * p[1]) + m[2] * p[2]) + m[3];
  r[1] = ((m[4] * p[0] + m[5] * p[1]) + m[6] * p[2]) + m[7];
  r[2] = ((m[8] * p[0] + m[9] * p[1]) + m[10] * p[2]) + m[11];
}

void flatten_instances(const float* positions, const uint32_t* indices, const vkrt_prim_mesh* pm, const vkrt_node* nodes,
                       uint32_t nodeCount, std::vector<FlatTri>& out)
{
  out.clear();
  uint32_t gid = 0;
  for(uint32_t n = 0; n < nodeCount; n++)
  {
    float m[12];
    for(int r = 0; r < 3; r++)
      for(int c = 0; c < 4; c++)
        m[r * 4 + c] = nodes[n].worldMatrix[c * 4 + r];
    const vkrt_prim_mesh& p = pm[nodes[n].primMesh];
    for(uint32_t t = 0; t < p.indexCount / 3; t++)
    {
      const uint32_t i0 = indices[p.firstIndex + 3 * t + 0] + p.vertexOffset;
      const uint32_t i1 = indices[p.firstIndex + 3 * t + 1] + p.vertexOffset;
      const uint32_t i2 = indices[p.firstIndex + 3 * t + 2] + p.vertexOffset;
      float a[3], b[3], c[3];
      xformPoint(m, positions + 3 * (size_t)i0, a);
      xformPoint(m, positions + 3 * (size_t)i1, b);
      xformPoint(m, positions + 3 * (size_t)i2, c);
      FlatTri ft;
      for(int k = 0; k < 3; k++) { ft.v0[k] = a[k]; ft.e1[k] = b[k] - a[k]; ft.e2[k] = c[k] - a[k]; ft.p1[k] = b[k]; ft.p2[k] = c[k]; }
      ft.gid = gid++; ft.inst = n; ft.prim = t;
      out.push_back(ft);
    }
  }
}

void build_sah_host(const std::vector<FlatTri>& tris, uint32_t maxLeaf, BuiltBvh& out, bool watertight)
{
  out.nodes.clear(); out.triOrder.clear(); out.maxDepth = 0; out.sahCost = 0; out.rootRef = (int32_t)0x80000000;
  const uint32_t n = (uint32_t)tris.size();
  if(n == 0)
    return;
  Ctx cx(tris, out, std::min<uint32_t>(std::max<uint32_t>(maxLeaf, 1u), 8u));
  cx.tb.resize(n); cx.cen.resize(3 * (size_t)n); cx.order.resize(n);
  out.triOrder.reserve(n);
  out.nodes.reserve((size_t)n * 16);
  Box all;
  all.reset();
  for(uint32_t i = 0; i < n; i++)
  {
    const FlatTri& t = tris[i];
    Box b;
    // boxes must contain what the ray/triangle test sees (both forms of the vertices) plus its reach (tri_prep.h)
    vkrt_tri_bounds(t.v0, t.p1, t.p2, t.e1, t.e2, watertight ? 1 : 0, b.lo, b.hi);
    for(int k = 0; k < 3; k++)
      cx.cen[3 * (size_t)i + k] = 0.5f * (b.lo[k] + b.hi[k]);
    cx.tb[i] = b;
    cx.order[i] = i;
    all.grow(b);
  }
  double cost = 0;
  out.rootRef = cx.build(0, n, all, 0, cost);
  out.sahCost = (float)cost;
}

void pack_tri_shade(const std::vector<FlatTri>& tris, const std::vector<uint32_t>& order, const uint32_t* indices, const vkrt_prim_mesh* pm,
                    const vkrt_node* nodes, std::vector<uint32_t>& out)
{
  out.resize(order.size() * 4);
  for(size_t s = 0; s < order.size(); s++)
  {
    const FlatTri& t = tris[order[s]];
    const vkrt_prim_mesh& p = pm[nodes[t.inst].primMesh];
    const uint32_t base = p.firstIndex + 3u * t.prim;
    out[4 * s + 0] = indices[base + 0] + p.vertexOffset;
    out[4 * s + 1] = indices[base + 1] + p.vertexOffset;
    out[4 * s + 2] = indices[base + 2] + p.vertexOffset;
    out[4 * s + 3] = (uint32_t)std::max(0, p.materialIndex);
  }
}

void pack_triangles(const std::vector<FlatTri>& tris, const std::vector<uint32_t>& order, std::vector<float>& out, bool watertight,
                    const std::vector<uint8_t>* instDissolves)
{
  out.resize(order.size() * 12);
  for(size_t s = 0; s < order.size(); s++)
  {
    const FlatTri& t = tris[order[s]];
    float* o = &out[s * 12];
    const float* r1 = watertight ? t.p1 : t.e1;
    const float* r2 = watertight ? t.p2 : t.e2;
    o[0] = t.v0[0]; o[1] = t.v0[1]; o[2] = t.v0[2]; o[3] = r1[0];
    o[4] = r1[1]; o[5] = r1[2]; o[6] = r2[0]; o[7] = r2[1];
    o[8] = r2[2];
    // any-hit stage (VKRT_OPT_ANYHIT_DISSOLVE): bit 31 of the id word flags a triangle of a non-opaque material
    const uint32_t idWord = t.gid | ((instDissolves && t.inst < instDissolves->size() && (*instDissolves)[t.inst]) ? 0x80000000u : 0u);
    memcpy(&o[9], &idWord, 4);
    memcpy(&o[10], &t.inst, 4);
    memcpy(&o[11], &t.prim, 4);
  }
}

}  // namespace vkrt

// =========================================================================================================
// wide8: collapse the binary SAH tree into 8-wide compressed nodes (see bvh_host.h)
//
// The collapse is the SAH-optimal one for the given binary topology (dynamic programme of Ylitie, Karras,
// Laine 2017, section 3): for every binary node n and every i = 1..7,
//   C(n,1) = min( leaf(n) = A_n * P_n * c_prim          (only if P_n <= 3 triangles),
//                 internal(n) = A_n * c_node + D(n,8) )
//   D(n,j) = min_{0<k<j} C(left,k) + C(right,j-k)       (n dissolved, children spread over j slots)
//   C(n,i) = min( D(n,i), C(n,i-1) )                     (i > 1)
// and the tree is then rebuilt from the arg-mins.  A greedy largest-area-first widening filled only ~3.5 of
// the 8 slots on average (many tiny bottom nodes); the DP fills them and merges 1-triangle binary leaves into
// <=3-triangle leaf children.  c_node : c_prim = 1 : 1: a triangle step issues ~70 instructions against ~250 of a node
// step, but runs with ~21 % of the lanes busy against ~68 % (r01_experiments.md #38), so per useful lane they cost the same.
// =========================================================================================================
namespace vkrt {
namespace {

constexpr float kNodeCost = 1.0f;
constexpr float kPrimCost = 1.0f;  // (0.3 / 0.6 / 1.5 measured and rejected, profiles/r03_experiments.md #101)

struct W8Child
{
  float lo[3], hi[3];
  int32_t ref;  // BVH2 ref: >=0 internal node, <0 leaf code
};

inline float boxAreaF(const float* lo, const float* hi)
{
  const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
  return 2.f * (dx * dy + dy * dz + dz * dx);
}

struct DpEntry
{
  float c[8];        // c[i] = C(n,i), i = 1..7
  float area;
  uint32_t prims;
  uint8_t asLeaf;    // C(n,1) chose the leaf form
  uint8_t split8;    // k of D(n,8) (children of the wide node rooted here)
  uint8_t split[8];  // split[i] = k of D(n,i) if C(n,i) == D(n,i), 0 = "use C(n,i-1)"
};

struct W8Ctx
{
  const BuiltBvh& b2;
  BuiltWide8& out;
  std::vector<DpEntry> dp;  // per binary internal node
  double sahSum = 0;
  float rootArea = 1.f;

  void children2(int32_t node, W8Child c[2]) const
  {
    const float* n = &b2.nodes[(size_t)node * 16];
    c[0].lo[0] = n[0]; c[0].lo[1] = n[1]; c[0].lo[2] = n[2]; c[0].hi[0] = n[3]; c[0].hi[1] = n[4]; c[0].hi[2] = n[5];
    c[1].lo[0] = n[6]; c[1].lo[1] = n[7]; c[1].lo[2] = n[8]; c[1].hi[0] = n[9]; c[1].hi[1] = n[10]; c[1].hi[2] = n[11];
    memcpy(&c[0].ref, &n[12], 4);
    memcpy(&c[1].ref, &n[13], 4);
  }
  static uint32_t leafPrims(int32_t ref) { return ((~(uint32_t)ref) & 7u) + 1u; }

  // cost of subtree `c` occupying i child slots
  float costOf(const W8Child& c, int i) const
  {
    if(c.ref < 0)
      return boxAreaF(c.lo, c.hi) * (float)leafPrims(c.ref) * kPrimCost;
    return dp[(size_t)c.ref].c[i];
  }
  uint32_t primsOf(const W8Child& c) const { return c.ref < 0 ? leafPrims(c.ref) : dp[(size_t)c.ref].prims; }

  // post-order DP (explicit stack: binary trees can be deep)
  void solve(int32_t root)
  {
    std::vector<std::pair<int32_t, int>> st;
    st.emplace_back(root, 0);
    while(!st.empty())
    {
      auto [n, phase] = st.back();
      st.pop_back();
      W8Child c[2];
      children2(n, c);
      if(phase == 0)
      {
        st.emplace_back(n, 1);
        if(c[0].ref >= 0) st.emplace_back(c[0].ref, 0);
        if(c[1].ref >= 0) st.emplace_back(c[1].ref, 0);
        continue;
      }
      DpEntry& e = dp[(size_t)n];
      float lo[3], hi[3];
      for(int k = 0; k < 3; k++) { lo[k] = std::min(c[0].lo[k], c[1].lo[k]); hi[k] = std::max(c[0].hi[k], c[1].hi[k]); }
      e.area = boxAreaF(lo, hi);
      e.prims = primsOf(c[0]) + primsOf(c[1]);
      auto distribute = [&](int j, uint8_t& bestK) {
        float best = INFINITY;
        bestK = 1;
        for(int k = 1; k < j; k++)
        {
          const int kl = std::min(k, 7), kr = std::min(j - k, 7);
          const float v = costOf(c[0], kl) + costOf(c[1], kr);
          if(v < best) { best = v; bestK = (uint8_t)k; }
        }
        return best;
      };
      const float d8 = distribute(8, e.split8);
      const float cInternal = e.area * kNodeCost + d8;
      const float cLeaf = e.prims <= 3u ? e.area * (float)e.prims * kPrimCost : INFINITY;
      e.asLeaf = cLeaf <= cInternal;
      e.c[0] = INFINITY;
      e.c[1] = e.asLeaf ? cLeaf : cInternal;
      e.split[0] = e.split[1] = 0;
      for(int i = 2; i <= 7; i++)
      {
        uint8_t k;
        const float dI = distribute(i, k);
        if(dI < e.c[i - 1]) { e.c[i] = dI; e.split[i] = k; }
        else { e.c[i] = e.c[i - 1]; e.split[i] = 0; }
      }
    }
  }

  // collect the children of a wide node: subtree c gets i slots
  void gather(const W8Child& c, int i, std::vector<W8Child>& kids) const
  {
    if(c.ref < 0) { kids.push_back(c); return; }
    const DpEntry& e = dp[(size_t)c.ref];
    while(i > 1 && e.split[i] == 0) i--;
    if(i == 1) { kids.push_back(c); return; }
    W8Child ch[2];
    children2(c.ref, ch);
    const int k = e.split[i];
    gather(ch[0], std::min(k, 7), kids);
    gather(ch[1], std::min(i - k, 7), kids);
  }
  void collectTris(const W8Child& c, std::vector<uint32_t>& t) const
  {
    if(c.ref < 0)
    {
      const uint32_t code = ~(uint32_t)c.ref, first = code >> 3, cnt = (code & 7u) + 1u;
      for(uint32_t k = 0; k < cnt; k++) t.push_back(b2.triOrder[first + k]);
      return;
    }
    W8Child ch[2];
    children2(c.ref, ch);
    collectTris(ch[0], t);
    collectTris(ch[1], t);
  }
  bool isLeafChild(const W8Child& c) const { return c.ref < 0 || dp[(size_t)c.ref].asLeaf; }

  // fills wide node `me` whose children are `kids`
  void emit(uint32_t me, const std::vector<W8Child>& kids, uint32_t depth)
  {
    out.maxDepth = std::max(out.maxDepth, depth);
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for(const W8Child& c : kids)
      for(int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], c.lo[k]); hi[k] = std::max(hi[k], c.hi[k]); }
    // octant-ordered slots: child on the +side of axis k wants a slot with bit k set (greedy assignment)
    int slotOf[8], childAt[8];
    for(int k = 0; k < 8; k++) { slotOf[k] = -1; childAt[k] = -1; }
    const float cen[3] = {0.5f * (lo[0] + hi[0]), 0.5f * (lo[1] + hi[1]), 0.5f * (lo[2] + hi[2])};
    for(size_t round = 0; round < kids.size(); round++)
    {
      float bestC = -INFINITY;
      int bc = -1, bs = -1;
      for(size_t c = 0; c < kids.size(); c++)
      {
        if(slotOf[c] >= 0) continue;
        const float v[3] = {0.5f * (kids[c].lo[0] + kids[c].hi[0]) - cen[0], 0.5f * (kids[c].lo[1] + kids[c].hi[1]) - cen[1],
                            0.5f * (kids[c].lo[2] + kids[c].hi[2]) - cen[2]};
        for(int s = 0; s < 8; s++)
        {
          if(childAt[s] >= 0) continue;
          const float cost = ((s & 1) ? v[0] : -v[0]) + ((s & 2) ? v[1] : -v[1]) + ((s & 4) ? v[2] : -v[2]);
          if(cost > bestC) { bestC = cost; bc = (int)c; bs = s; }
        }
      }
      if(bc < 0)
      {  // every comparison failed (NaN / inf boxes): keep the assignment total -- first free child, first free slot
        for(size_t k = 0; k < kids.size() && bc < 0; k++) if(slotOf[k] < 0) bc = (int)k;
        for(int s = 0; s < 8 && bs < 0; s++) if(childAt[s] < 0) bs = s;
      }
      slotOf[bc] = bs;
      childAt[bs] = bc;
    }
    // grid: origin = lo, per-axis power-of-two cell so that the extent fits QMAX cells
    const int QMAX = VKRT_WNODE_QMAX;
    uint32_t eb[3];
    for(int k = 0; k < 3; k++)
    {
      const double ext = (double)hi[k] - (double)lo[k];
      int e = ext > 0 ? (int)std::ceil(std::log2(ext / (double)QMAX)) : -126;
      e = std::min(std::max(e, -126), 126);
      for(;;)
      {  // make sure every child's hi really fits (ceil may need one more cell)
        const double sc = std::ldexp(1.0, e);
        bool ok = true;
        for(const W8Child& c : kids)
          if(std::ceil(((double)c.hi[k] - (double)lo[k]) / sc) > (double)QMAX) ok = false;
        if(ok || e >= 126) break;
        e++;
      }
      eb[k] = (uint32_t)(e + 127);
    }
    uint32_t imask = 0, meta[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint16_t qlo[3][8], qhi[3][8];
    memset(qlo, 0, sizeof qlo);
    memset(qhi, 0, sizeof qhi);
    const uint32_t triBase = (uint32_t)out.triOrder.size();
    uint32_t triOff = 0;
    std::vector<std::pair<int, int>> internalSlots;  // (slot, child)
    for(int s = 0; s < 8; s++)
    {
      const int c = childAt[s];
      if(c < 0)
        continue;
      const W8Child& ch = kids[(size_t)c];
      for(int k = 0; k < 3; k++)
      {
        const double sc = std::ldexp(1.0, (int)eb[k] - 127), o = (double)lo[k];
        int ql = (int)std::floor(((double)ch.lo[k] - o) / sc);
        ql = std::min(std::max(ql, 0), QMAX);
        while(ql > 0 && o + ql * sc > (double)ch.lo[k]) ql--;
        int qh = (int)std::ceil(((double)ch.hi[k] - o) / sc);
        qh = std::min(std::max(qh, 0), QMAX);
        while(qh < QMAX && o + qh * sc < (double)ch.hi[k]) qh++;
        qlo[k][s] = (uint16_t)ql;
        qhi[k][s] = (uint16_t)qh;
      }
      if(!isLeafChild(ch))
      {
        imask |= 1u << s;
        meta[s] = 0x20u | (24u + (uint32_t)s);
        internalSlots.emplace_back(s, c);
      }
      else
      {
        std::vector<uint32_t> t;
        collectTris(ch, t);  // <= 3 by construction
        for(uint32_t x : t) out.triOrder.push_back(x);
        const uint32_t cnt = (uint32_t)t.size();
        meta[s] = (((1u << cnt) - 1u) << 5) | triOff;
        triOff += cnt;
        sahSum += (double)boxAreaF(ch.lo, ch.hi) * cnt * kPrimCost;
      }
    }
    sahSum += (double)boxAreaF(lo, hi) * kNodeCost;  // one node visit
    const uint32_t childBase = (uint32_t)(out.nodes.size() / VKRT_WNODE_DWORDS);
    out.nodes.resize(out.nodes.size() + VKRT_WNODE_DWORDS * internalSlots.size());
    uint32_t* n = &out.nodes[(size_t)me * VKRT_WNODE_DWORDS];
    memcpy(&n[0], &lo[0], 4); memcpy(&n[1], &lo[1], 4); memcpy(&n[2], &lo[2], 4);
    n[3] = eb[0] | (eb[1] << 8) | (eb[2] << 16) | (imask << 24);
    n[4] = childBase;
    n[5] = triBase;
    n[6] = meta[0] | (meta[1] << 8) | (meta[2] << 16) | (meta[3] << 24);
    n[7] = meta[4] | (meta[5] << 8) | (meta[6] << 16) | (meta[7] << 24);
    vkrt_wnode_store_planes(n, qlo, qhi);
    // recurse (internalSlots is in increasing slot order = storage order)
    for(size_t k = 0; k < internalSlots.size(); k++)
    {
      const W8Child& ch = kids[(size_t)internalSlots[k].second];
      W8Child c2[2];
      children2(ch.ref, c2);
      const int ks = dp[(size_t)ch.ref].split8;
      std::vector<W8Child> sub;
      gather(c2[0], std::min(ks, 7), sub);
      gather(c2[1], std::min(8 - ks, 7), sub);
      emit(childBase + (uint32_t)k, sub, depth + 1);
    }
  }
};

}  // namespace

void build_wide8_host(const std::vector<FlatTri>& tris, BuiltWide8& out, bool watertight)
{
  out = BuiltWide8{};
  if(tris.empty())
    return;
  BuiltBvh b2;
  build_sah_host(tris, 1, b2, watertight);  // one triangle per binary leaf; the DP forms the <=3-triangle leaf children
  collapse_wide8(b2, tris, out, watertight);
}

void collapse_wide8(const BuiltBvh& b2, const std::vector<FlatTri>& tris, BuiltWide8& out, bool watertight)
{
  out = BuiltWide8{};
  if(tris.empty())
    return;
  W8Ctx cx{b2, out};
  out.triOrder.reserve(tris.size());
  out.nodes.resize(VKRT_WNODE_DWORDS);
  std::vector<W8Child> kids;
  if(b2.rootRef < 0)
  {  // single triangle: a root with one leaf child
    W8Child c;
    const FlatTri& t = tris[b2.triOrder.empty() ? 0 : b2.triOrder[0]];
    vkrt_tri_bounds(t.v0, t.p1, t.p2, t.e1, t.e2, watertight ? 1 : 0, c.lo, c.hi);
    c.ref = b2.rootRef;
    kids.push_back(c);
  }
  else
  {
    cx.dp.resize(b2.nodes.size() / 16);
    cx.solve(b2.rootRef);
    W8Child c[2];
    cx.children2(b2.rootRef, c);
    if(cx.dp[(size_t)b2.rootRef].asLeaf)
    {  // whole scene <= 3 triangles: root with one leaf child covering the binary root
      W8Child r;
      for(int k = 0; k < 3; k++) { r.lo[k] = std::min(c[0].lo[k], c[1].lo[k]); r.hi[k] = std::max(c[0].hi[k], c[1].hi[k]); }
      r.ref = b2.rootRef;
      kids.push_back(r);
    }
    else
    {
      const int ks = cx.dp[(size_t)b2.rootRef].split8;
      cx.gather(c[0], std::min(ks, 7), kids);
      cx.gather(c[1], std::min(8 - ks, 7), kids);
    }
  }
  {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for(const W8Child& c : kids)
      for(int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], c.lo[k]); hi[k] = std::max(hi[k], c.hi[k]); }
    cx.rootArea = std::max(boxAreaF(lo, hi), 1e-30f);
  }
  cx.emit(0, kids, 0);
  out.nodeCount = (uint32_t)(out.nodes.size() / VKRT_WNODE_DWORDS);
  out.sahCost = (float)(cx.sahSum / cx.rootArea);
}

}  // namespace vkrt
