/*
 * oracle.cpp -- scalar CPU restatement of the reference's path-tracing path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and
 * only as the checker / reported CPU baseline.  The product (libvkrt.so) never links,
 * loads or calls it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden images or known-answer vectors
 * for this path and cannot be built or run here (needs Vulkan,
 * glslang, nvpro_core, NRD/NRI and an RT GPU).  This restatement follows the GLSL
 * sources line by line (citations below are relative to the reference tree); it is
 * pinned only by (i) the integer PRNG known answers in tests/golden/prng_kat.json,
 * (ii) an independent numpy-float32 restatement of the shading functions
 * (oracle/np_shading.py), (iii) brute-force vs BVH agreement, (iv) analytic images.
 *
 * Arithmetic profile ("vkrt math profile") -- what the GLSL leaves
 * implementation-defined is fixed as follows, identically in the HIP kernels:
 *   - IEEE-754 binary32, round-to-nearest-even, subnormals kept, no contraction
 *     (-ffp-contract=off); + - * / sqrt correctly rounded.
 *   - shader math is evaluated in source order, left to right, without FMA;
 *     dot(a,b) = (a.x*b.x + a.y*b.y) + a.z*b.z.
 *   - normalize(v) = v * (1/sqrt(dot(v,v))); length(v) = sqrt(dot(v,v)).
 *   - min(x,y) = y<x ? y : x ; max(x,y) = x<y ? y : x (GLSL 4.60 section 8.3).
 *   - sin/cos: Cody-Waite reduction by pi/4 + degree-7/8 minimax polynomials with
 *     explicit fmaf (vk_sincos below); pow(x,5) = ((x*x)*(x*x))*x.
 *   - ray/triangle and ray/box tests (driver-defined in the reference, raytrace.rgen:64)
 *     use explicit fmaf as written in isect_* below; closest hit = smallest t, ties
 *     broken by the smallest flattened triangle id (instance-major, then primitive).
 */
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/vkrt.h"

namespace {

// ------------------------------------------------------------------------------------------
// vec3 helpers (source-order float arithmetic)
// ------------------------------------------------------------------------------------------
struct V3 { float x, y, z; };
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 v3(float s) { return V3{s, s, s}; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
inline V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 cross(V3 a, V3 b)
{
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline float length(V3 a) { return sqrtf(dot(a, a)); }
inline V3 normalize(V3 a)
{
  float inv = 1.0f / sqrtf(dot(a, a));
  return a * inv;
}
inline float glsl_min(float x, float y) { return (y < x) ? y : x; }
inline float glsl_max(float x, float y) { return (x < y) ? y : x; }
inline float glsl_clamp(float x, float lo, float hi) { return glsl_min(glsl_max(x, lo), hi); }
inline V3 glsl_mix(V3 a, V3 b, float t) { return a * (1.0f - t) + b * t; }
// reflect(I,N) = I - 2*dot(N,I)*N  (GLSL 4.60 section 8.5)
inline V3 glsl_reflect(V3 I, V3 N) { return I - (2.0f * dot(N, I)) * N; }

// globals.glsl:4-5
const float M_PI_F = 3.14159265f;
const float M_INV_PI_F = 1.0f / M_PI_F;

// ------------------------------------------------------------------------------------------
// math profile: sin / cos / pow5
// ------------------------------------------------------------------------------------------
inline void vk_sincos(float x, float* s_out, float* c_out)
{
  const float FOPI = 1.27323954473516f;  // 4/pi
  const float DP1 = 0.78515625f, DP2 = 2.4187564849853515625e-4f, DP3 = 3.77489497744594108e-8f;
  float ax = fabsf(x);
  int j = (int)(ax * FOPI);
  j = (j + 1) & ~1;
  float y = (float)j;
  float r = fmaf(y, -DP1, ax);
  r = fmaf(y, -DP2, r);
  r = fmaf(y, -DP3, r);
  float z = r * r;
  float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
  float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f),
                  z * z, fmaf(-0.5f, z, 1.0f));
  int q = (j >> 1) & 3;
  float s, c;
  if(q == 0) { s = ps; c = pc; }
  else if(q == 1) { s = pc; c = -ps; }
  else if(q == 2) { s = -ps; c = -pc; }
  else { s = -pc; c = ps; }
  if(x < 0.0f) s = -s;
  *s_out = s;
  *c_out = c;
}
inline float vk_sin(float x) { float s, c; vk_sincos(x, &s, &c); return s; }
inline float vk_cos(float x) { float s, c; vk_sincos(x, &s, &c); return c; }
inline float vk_pow5(float x) { float x2 = x * x; return (x2 * x2) * x; }

// ------------------------------------------------------------------------------------------
// random.glsl:6-33  (integer exact)
// ------------------------------------------------------------------------------------------
inline uint32_t tea(uint32_t val0, uint32_t val1)
{
  uint32_t v0 = val0, v1 = val1, s0 = 0u;
  for(uint32_t n = 0; n < 16u; n++)
  {
    s0 += 0x9e3779b9u;
    v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
    v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
  }
  return v0;
}
inline uint32_t lcg(uint32_t& prev)
{
  prev = 1664525u * prev + 1013904223u;
  return prev & 0x00FFFFFFu;
}
inline float rnd(uint32_t& prev) { return (float)lcg(prev) / (float)0x01000000; }

// random.glsl:35-45
inline V3 samplingHemisphere(uint32_t& seed, V3 x, V3 y, V3 z)
{
  float r1 = rnd(seed);
  float r2 = rnd(seed);
  float sq = sqrtf(r1);
  float sn, cs;
  vk_sincos(2 * M_PI_F * r2, &sn, &cs);
  V3 direction = v3(cs * sq, sn * sq, sqrtf(1 - r1));
  direction = direction.x * x + direction.y * y + direction.z * z;
  return direction;
}
// random.glsl:47-54
inline void createCoordinateSystem(V3 N, V3& Nt, V3& Nb)
{
  if(fabsf(N.x) > fabsf(N.y))
    Nt = v3(N.z, 0, -N.x) / sqrtf(N.x * N.x + N.z * N.z);
  else
    Nt = v3(0, -N.z, N.y) / sqrtf(N.y * N.y + N.z * N.z);
  Nb = cross(N, Nt);
}
// random.glsl:56-70
inline V3 samplingNDF_GGXTR(uint32_t& seed, float alpha2)
{
  float r1 = rnd(seed);
  float r2 = rnd(seed);
  float cosTheta = sqrtf((1.0f - r2) / ((alpha2 - 1.0f) * r2 + 1.0f));
  float sinTheta = glsl_clamp(sqrtf(1.0f - cosTheta * cosTheta), 0.0f, 1.0f);
  float phi = r1 * 2.0f * M_PI_F;
  float sinPhi, cosPhi;
  vk_sincos(phi, &sinPhi, &cosPhi);
  return v3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
}

// ------------------------------------------------------------------------------------------
// Scene (flat arrays of include/vkrt.h copied; instances flattened to world space)
// ------------------------------------------------------------------------------------------
struct Instance
{
  float o2w[3][4];  // row-major 3x4 of the column-major worldMatrix
  float w2o[3][3];  // inverse of the upper-left 3x3
  int32_t primMesh;
};
struct Tri  // world-space triangle for traversal
{
  V3 v0, e1, e2;
  V3 p1, p2;  // exact world-space vertices 1 and 2 (v0 + e1 rounds away from p1): what the watertight test works on
  uint32_t gid, inst, prim;
};
struct TexLevel
{
  uint32_t w, h;
  std::vector<uint8_t> rgba;
};
struct Tex
{
  uint32_t w, h;
  std::vector<uint8_t> rgba;
  bool srgb;
  std::vector<TexLevel> mips;  // mips[0] unused (level 0 is rgba); levels 1.. as nvvk::cmdGenerateMipmaps blits them (hello_vulkan.cpp:499)
};
struct Aabb
{
  float lo[3], hi[3];
};
struct BvhNode  // internal node: two child boxes + refs (canonical 64-byte node of SURVEY 8d)
{
  Aabb box[2];
  int32_t child[2];  // >=0 internal node index, <0 : ~leafIndex
};
struct BvhLeaf
{
  uint32_t first, count;  // into triOrder
};

struct Counters
{
  uint64_t rays_closest = 0, rays_shadow = 0, hits = 0, diffuse_hits = 0, tex_taps = 0, pixels = 0,
           nodes_visited = 0, tris_tested = 0;
  void add(const Counters& o)
  {
    rays_closest += o.rays_closest; rays_shadow += o.rays_shadow; hits += o.hits;
    diffuse_hits += o.diffuse_hits; tex_taps += o.tex_taps; pixels += o.pixels;
    nodes_visited += o.nodes_visited; tris_tested += o.tris_tested;
  }
};

}  // namespace

struct orc_scene
{
  std::vector<V3> pos, nrm;
  std::vector<float> tan4;  // vec4
  std::vector<float> uv2;   // vec2
  std::vector<uint32_t> idx;
  std::vector<vkrt_prim_mesh> pm;
  std::vector<GltfPBRMaterial> mats;
  std::vector<GltfLight> lights;
  std::vector<Instance> inst;
  std::vector<Tex> tex;
  float srgb_lut[256];
  int gbufferMips = 1;  // hybrid G-buffer samples with implicit LOD (fragment shader semantics); 0 = LOD 0
  int watertight = 0;   // ray/triangle test: 0 = Moeller-Trumbore (default), 1 = Woop-Benthin-Wald watertight (VKRT_OPT_WATERTIGHT)
  int dissolve = 0;     // any-hit alpha / dissolve (raytrace_rahit_todo.glsl): 0 = every geometry opaque (what the reference runs), 1 = on
  std::vector<Tri> tris;  // flattened, gid order
  // BVH
  std::vector<BvhNode> nodes;
  std::vector<BvhLeaf> leaves;
  std::vector<uint32_t> triOrder;
  bool rootIsLeaf = false;
  uint32_t maxDepth = 0;
  double sahCost = 0;
};

namespace {

// Inverse of the upper-left 3x3 in double, cofactor form (fixed operation order; the
// product's host code computes W2O the same way so both sides hold identical floats).
void invert3x3(const float m[3][4], float out[3][3])
{
  double a = m[0][0], b = m[0][1], c = m[0][2];
  double d = m[1][0], e = m[1][1], f = m[1][2];
  double g = m[2][0], h = m[2][1], i = m[2][2];
  double A = e * i - f * h;
  double B = -(d * i - f * g);
  double C = d * h - e * g;
  double det = a * A + b * B + c * C;
  double inv = 1.0 / det;
  out[0][0] = (float)(A * inv);
  out[0][1] = (float)(-(b * i - c * h) * inv);
  out[0][2] = (float)((b * f - c * e) * inv);
  out[1][0] = (float)(B * inv);
  out[1][1] = (float)((a * i - c * g) * inv);
  out[1][2] = (float)(-(a * f - c * d) * inv);
  out[2][0] = (float)(C * inv);
  out[2][1] = (float)(-(a * h - b * g) * inv);
  out[2][2] = (float)((a * e - b * d) * inv);
}

// gl_ObjectToWorldEXT * vec4(p,1): column0*x + column1*y + column2*z + column3*1
inline V3 xformPoint(const Instance& in, V3 p)
{
  V3 r;
  r.x = ((in.o2w[0][0] * p.x + in.o2w[0][1] * p.y) + in.o2w[0][2] * p.z) + in.o2w[0][3];
  r.y = ((in.o2w[1][0] * p.x + in.o2w[1][1] * p.y) + in.o2w[1][2] * p.z) + in.o2w[1][3];
  r.z = ((in.o2w[2][0] * p.x + in.o2w[2][1] * p.y) + in.o2w[2][2] * p.z) + in.o2w[2][3];
  return r;
}
// vec3(n * gl_WorldToObjectEXT): component j = dot(n, column j of W2O)
inline V3 xformNormal(const Instance& in, V3 n)
{
  V3 r;
  r.x = (n.x * in.w2o[0][0] + n.y * in.w2o[1][0]) + n.z * in.w2o[2][0];
  r.y = (n.x * in.w2o[0][1] + n.y * in.w2o[1][1]) + n.z * in.w2o[2][1];
  r.z = (n.x * in.w2o[0][2] + n.y * in.w2o[1][2]) + n.z * in.w2o[2][2];
  return r;
}

// ------------------------------------------------------------------------------------------
// Ray / triangle and ray / box (the part the reference leaves to the Vulkan driver,
// raytrace.rgen:64-75 traceRayEXT).  Explicit fmaf; see DESIGN.md section 3.
// ------------------------------------------------------------------------------------------
inline float fdot(V3 a, V3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
inline V3 fcross(V3 a, V3 b)
{
  return v3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
// Returns true when the ray hits the triangle's interior; t,u,v by one IEEE division.
inline bool isect_tri(V3 o, V3 d, const Tri& tr, float& t, float& u, float& v)
{
  V3 pvec = fcross(d, tr.e2);
  float det = fdot(tr.e1, pvec);
  V3 tvec = o - tr.v0;
  float U = fdot(tvec, pvec);
  V3 qvec = fcross(tvec, tr.e1);
  float V = fdot(d, qvec);
  float T = fdot(tr.e2, qvec);
  bool ok;
  if(det > 0.0f)
    ok = (U >= 0.0f) && (V >= 0.0f) && (U + V <= det);
  else if(det < 0.0f)
    ok = (U <= 0.0f) && (V <= 0.0f) && (U + V >= det);
  else
    ok = false;
  if(!ok)
    return false;
  float inv = 1.0f / det;
  t = T * inv;
  u = U * inv;
  v = V * inv;
  return true;
}

// Watertight alternative (include/vkrt.h VKRT_OPT_WATERTIGHT): Woop, Benthin, Wald, "Watertight Ray/Triangle Intersection", JCGT 2013,
// on the exact vertices; what the Vulkan specification demands of traceRayEXT (raytrace.rgen:64-75).  Translate to the ray origin,
// permute so that the dominant direction axis is z, shear, three 2D edge functions (products and differences only: a shared edge
// gets the exact negative in the neighbouring triangle), exact zeros re-evaluated in double, no culling.  Same operation order as
// csrc/traverse.h tri_test_wt.
struct WtRay
{
  int kz;
  float Sx, Sy, Sz;
};
inline WtRay wt_prepare(V3 d)
{
  WtRay R;
  const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
  R.kz = (ax > ay) ? (ax > az ? 0 : 2) : (ay > az ? 1 : 2);
  const float dz = R.kz == 0 ? d.x : (R.kz == 1 ? d.y : d.z);
  const float dx = R.kz == 0 ? d.y : (R.kz == 1 ? d.z : d.x);
  const float dy = R.kz == 0 ? d.z : (R.kz == 1 ? d.x : d.y);
  R.Sx = dx / dz; R.Sy = dy / dz; R.Sz = 1.0f / dz;
  return R;
}
inline V3 wt_permute(int kz, V3 v) { return kz == 0 ? v3(v.y, v.z, v.x) : (kz == 1 ? v3(v.z, v.x, v.y) : v); }
inline bool isect_tri_wt(const WtRay& R, V3 o, const Tri& tr, float& t, float& u, float& v)
{
  const V3 A = wt_permute(R.kz, tr.v0 - o), B = wt_permute(R.kz, tr.p1 - o), C = wt_permute(R.kz, tr.p2 - o);
  const float Ax = A.x - R.Sx * A.z, Ay = A.y - R.Sy * A.z;
  const float Bx = B.x - R.Sx * B.z, By = B.y - R.Sy * B.z;
  const float Cx = C.x - R.Sx * C.z, Cy = C.y - R.Sy * C.z;
  float U = Cx * By - Cy * Bx, V = Ax * Cy - Ay * Cx, W = Bx * Ay - By * Ax;
  if(U == 0.0f || V == 0.0f || W == 0.0f)
  {
    U = (float)((double)Cx * (double)By - (double)Cy * (double)Bx);
    V = (float)((double)Ax * (double)Cy - (double)Ay * (double)Cx);
    W = (float)((double)Bx * (double)Ay - (double)By * (double)Ax);
  }
  if((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f))
    return false;
  const float det = (U + V) + W;
  if(det == 0.0f)
    return false;
  // distance from the triangle's plane (not the paper's barycentric average of the sheared depths, whose error grows with the
  // triangle's depth range: seed 31004365 of the round-3 campaign), on the translated, permuted vertices: N.d' = d'z (Nx Sx + Ny Sy
  // + Nz), 1 / d'z = Sz; see csrc/traverse.h tri_test_wt
  const V3 N = cross(B - A, C - A);
  const float den = (N.x * R.Sx + N.y * R.Sy) + N.z;
  if(den == 0.0f)
    return false;
  const float inv = 1.0f / det;
  t = (dot(N, A) * R.Sz) / den;
  u = V * inv;
  v = W * inv;
  return true;
}
// Any-hit stage (include/vkrt.h VKRT_OPT_ANYHIT_DISSOLVE; reference raytrace_rahit_todo.glsl:23-37, never compiled there and written
// against the dead OBJ pipeline's WaveFrontMaterial): a candidate hit on a non-opaque material is ignored when the material's
// dissolve is 0, and otherwise with probability 1 - dissolve.  Mapped onto the glTF material the live pipeline has:
// dissolve = pbrBaseColorFactor.a, "illum == 4" = dissolve < 1.  The GLSL draws rnd(prd.seed) per invocation; Vulkan leaves both
// the order and the number of any-hit invocations to the implementation, so no sequence of draws can be THE reference's.  The
// decision is therefore a pure function of the ray and the triangle -- rnd(tea(gid, seed of the payload when the ray is traced))
// -- which makes the result a property of the triangle set like everything else here, and prd.seed is left as it is.
inline bool dissolveIgnores(float alpha, uint32_t gid, uint32_t raySeed)
{
  if(alpha == 0.0f)
    return true;
  uint32_t st = tea(gid, raySeed);
  return rnd(st) > alpha;
}
// the scene's ray/triangle test (+ any-hit stage) for one ray
struct TriTester
{
  const orc_scene& sc;
  bool wt;
  WtRay R;
  uint32_t seed;
  bool stage;  // false: this query has no any-hit stage (the G-buffer's primary rays stand for a raster pass)
  TriTester(const orc_scene& s, V3 d, uint32_t raySeed, bool anyHitStage)
      : sc(s), wt(s.watertight != 0), R(wt ? wt_prepare(d) : WtRay{2, 0.0f, 0.0f, 0.0f}), seed(raySeed), stage(anyHitStage && s.dissolve != 0)
  {
  }
  bool operator()(V3 o, V3 d, const Tri& tr, float& t, float& u, float& v) const
  {
    return wt ? isect_tri_wt(R, o, tr, t, u, v) : isect_tri(o, d, tr, t, u, v);
  }
  // true: the candidate hit on `tr` is ignored (ignoreIntersectionEXT)
  bool ignores(const Tri& tr) const
  {
    if(!stage)
      return false;
    const float alpha = sc.mats[(size_t)std::max(0, sc.pm[sc.inst[tr.inst].primMesh].materialIndex)].pbrBaseColorFactor[3];
    return alpha < 1.0f && dissolveIgnores(alpha, tr.gid, seed);
  }
};

struct RayInv
{
  V3 o, id;  // origin, 1/direction with |d| clamped away from zero
};
inline float safe_inv(float d)
{
  const float tiny = 1e-20f;
  float dd = (fabsf(d) < tiny) ? copysignf(tiny, d) : d;
  return 1.0f / dd;
}
// Conservative slab test: true if [tnear,tfar] overlaps [tmin,tmax].  Slabs widened by 2e-5 * max(|t0|,|t1|) per axis and the
// far side by a relative 4e-5 -- the margin covers the rounding of the slab arithmetic (Ize 2013) AND of the triangle test,
// which can accept a ray just outside the exact triangle; with these margins BVH and brute force agree (see test_oracle.py).
// Round 2 (found by tools/fuzz_parity.py on a scene with a 500-unit triangle): the triangle test's error in t grows with the
// distance |o - v0| in ALL axes (times 1 / cos of the incidence), not with the distance along the slab's own axis, and it can
// carry a hit across tmin / tmax that the box test had clamped away.  So every axis' pad also covers 2e-5 x the largest
// distance from the origin to the box in any axis (times that axis' |1/d|, capped for near-parallel axes whose slabs are
// decided by their own huge t), and the [tmin, tmax] clamp is relaxed by the largest pad.
inline bool isect_box(const RayInv& r, const Aabb& b, float tmin, float tmax, float& tnear)
{
  float t0x = (b.lo[0] - r.o.x) * r.id.x, t1x = (b.hi[0] - r.o.x) * r.id.x;
  float t0y = (b.lo[1] - r.o.y) * r.id.y, t1y = (b.hi[1] - r.o.y) * r.id.y;
  float t0z = (b.lo[2] - r.o.z) * r.id.z, t1z = (b.hi[2] - r.o.z) * r.id.z;
  const float m = fmaxf(fmaxf(fmaxf(fabsf(b.lo[0] - r.o.x), fabsf(b.hi[0] - r.o.x)), fmaxf(fabsf(b.lo[1] - r.o.y), fabsf(b.hi[1] - r.o.y))),
                        fmaxf(fabsf(b.lo[2] - r.o.z), fabsf(b.hi[2] - r.o.z)));
  const float cx = m * fminf(fabsf(r.id.x), 1.0e4f), cy = m * fminf(fabsf(r.id.y), 1.0e4f), cz = m * fminf(fabsf(r.id.z), 1.0e4f);
  float px = 2.0e-5f * fmaxf(fmaxf(fabsf(t0x), fabsf(t1x)), cx), py = 2.0e-5f * fmaxf(fmaxf(fabsf(t0y), fabsf(t1y)), cy),
        pz = 2.0e-5f * fmaxf(fmaxf(fabsf(t0z), fabsf(t1z)), cz);
  const float pc = 2.0e-5f * fmaxf(cx, fmaxf(cy, cz));  // slack of the [tmin, tmax] clamp
  float tn = fmaxf(fmaxf(fminf(t0x, t1x) - px, fminf(t0y, t1y) - py), fmaxf(fminf(t0z, t1z) - pz, tmin - pc));
  float tf = fminf(fminf(fmaxf(t0x, t1x) + px, fmaxf(t0y, t1y) + py), fminf(fmaxf(t0z, t1z) + pz, tmax + pc));
  tnear = tn;
  return tn <= tf * 1.00004f;
}

struct Hit
{
  float t, u, v;
  int32_t tri;  // index into scene.tris (== gid), -1 = miss
};

// accept rule shared by brute force and BVH: smallest t in (tmin, tmax); ties -> smallest gid
inline void consider(const TriTester& test, const Tri& tr, V3 o, V3 d, float tmin, Hit& best, Counters& c)
{
  float t, u, v;
  c.tris_tested++;
  if(!test(o, d, tr, t, u, v))
    return;
  if(!(t > tmin))
    return;
  if(!(t < best.t || (t == best.t && (int32_t)tr.gid < best.tri)))
    return;
  if(test.ignores(tr))
    return;
  if(true)
  {
    best.t = t; best.u = u; best.v = v; best.tri = (int32_t)tr.gid;
  }
}

// Optional tap (debug / parity localisation): every ray handed to the four queries below is appended as 9 floats
// {o.xyz, d.xyz, tmin, tmax, any ? 1 : 0} while a thread has set it (orc_hybrid_pixel_rays).
thread_local std::vector<float>* g_rayTap = nullptr;
inline void tapRay(V3 o, V3 d, float tmin, float tmax, bool any)
{
  if(g_rayTap)
  {
    const float r[9] = {o.x, o.y, o.z, d.x, d.y, d.z, tmin, tmax, any ? 1.0f : 0.0f};
    g_rayTap->insert(g_rayTap->end(), r, r + 9);
  }
}

Hit closest_brute(const orc_scene& s, V3 o, V3 d, float tmin, float tmax, Counters& c, uint32_t raySeed = 0, bool anyHitStage = true)
{
  tapRay(o, d, tmin, tmax, false);
  Hit best{tmax, 0, 0, -1};
  const TriTester test(s, d, raySeed, anyHitStage);
  // tmax exclusive: a hit needs t < tmax; emulate by starting best.t = tmax with tri = -1
  // (tie rule "gid < -1" never holds, so t == tmax is rejected).
  for(const Tri& tr : s.tris)
    consider(test, tr, o, d, tmin, best, c);
  return best;
}
bool any_brute(const orc_scene& s, V3 o, V3 d, float tmin, float tmax, Counters& c, uint32_t raySeed = 0, bool anyHitStage = true)
{
  tapRay(o, d, tmin, tmax, true);
  const TriTester test(s, d, raySeed, anyHitStage);
  for(const Tri& tr : s.tris)
  {
    float t, u, v;
    c.tris_tested++;
    if(test(o, d, tr, t, u, v) && t > tmin && t < tmax && !test.ignores(tr))
      return true;
  }
  return false;
}

Hit closest_bvh(const orc_scene& s, V3 o, V3 d, float tmin, float tmax, Counters& c, uint32_t raySeed = 0, bool anyHitStage = true)
{
  tapRay(o, d, tmin, tmax, false);
  Hit best{tmax, 0, 0, -1};
  if(s.tris.empty())
    return best;
  RayInv r{o, v3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z))};
  const TriTester test(s, d, raySeed, anyHitStage);
  int32_t stack[256];
  int sp = 0;
  int32_t cur = s.rootIsLeaf ? ~0 : 0;
  for(;;)
  {
    if(cur < 0)
    {
      const BvhLeaf& lf = s.leaves[~cur];
      for(uint32_t k = 0; k < lf.count; k++)
        consider(test, s.tris[s.triOrder[lf.first + k]], o, d, tmin, best, c);
    }
    else
    {
      const BvhNode& n = s.nodes[cur];
      c.nodes_visited++;
      float tn0, tn1;
      bool h0 = isect_box(r, n.box[0], tmin, best.t, tn0);
      bool h1 = isect_box(r, n.box[1], tmin, best.t, tn1);
      if(h0 && h1)
      {
        int nearI = (tn1 < tn0) ? 1 : 0;
        stack[sp++] = n.child[1 - nearI];
        cur = n.child[nearI];
        continue;
      }
      if(h0) { cur = n.child[0]; continue; }
      if(h1) { cur = n.child[1]; continue; }
    }
    if(sp == 0)
      break;
    cur = stack[--sp];
  }
  return best;
}
bool any_bvh(const orc_scene& s, V3 o, V3 d, float tmin, float tmax, Counters& c, uint32_t raySeed = 0, bool anyHitStage = true)
{
  tapRay(o, d, tmin, tmax, true);
  if(s.tris.empty())
    return false;
  RayInv r{o, v3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z))};
  const TriTester test(s, d, raySeed, anyHitStage);
  int32_t stack[256];
  int sp = 0;
  int32_t cur = s.rootIsLeaf ? ~0 : 0;
  for(;;)
  {
    if(cur < 0)
    {
      const BvhLeaf& lf = s.leaves[~cur];
      for(uint32_t k = 0; k < lf.count; k++)
      {
        float t, u, v;
        c.tris_tested++;
        if(test(o, d, s.tris[s.triOrder[lf.first + k]], t, u, v) && t > tmin && t < tmax && !test.ignores(s.tris[s.triOrder[lf.first + k]]))
          return true;
      }
    }
    else
    {
      const BvhNode& n = s.nodes[cur];
      c.nodes_visited++;
      float tn0, tn1;
      bool h0 = isect_box(r, n.box[0], tmin, tmax, tn0);
      bool h1 = isect_box(r, n.box[1], tmin, tmax, tn1);
      if(h0 && h1)
      {
        int nearI = (tn1 < tn0) ? 1 : 0;
        stack[sp++] = n.child[1 - nearI];
        cur = n.child[nearI];
        continue;
      }
      if(h0) { cur = n.child[0]; continue; }
      if(h1) { cur = n.child[1]; continue; }
    }
    if(sp == 0)
      break;
    cur = stack[--sp];
  }
  return false;
}

// How far outside the exact triangle the binary32 Moeller-Trumbore test above can accept a point, for ray origins within one
// triangle length: its u, v carry an absolute error ~ eps |o - v0| / (sin(phi) cos(theta)) (phi = angle between e1 and e2) in
// units of the edge length.  Negligible for ordinary triangles, 0.06 units for a 750-unit needle enclosing 7e-4 rad (round 3,
// tools/fuzz_parity.py seed 1301004260: a one-triangle leaf box pruned a "hit" that the loop over all triangles finds).  The
// closest-hit / any-hit result is DEFINED by the triangle test over all triangles, so the tree's boxes have to cover its reach:
// the box of every triangle with sin(phi) < 1/8 is widened by 16 eps max(|e1|, |e2|) / sin(phi), capped at the triangle's length.  Same rule, same
// operation order as the product's builders (csrc/tri_prep.h).
inline float triSlop(V3 e1, V3 e2)
{
  const float l1 = (e1.x * e1.x + e1.y * e1.y) + e1.z * e1.z;
  const float l2 = (e2.x * e2.x + e2.y * e2.y) + e2.z * e2.z;
  const float cx = e1.y * e2.z - e1.z * e2.y, cy = e1.z * e2.x - e1.x * e2.z, cz = e1.x * e2.y - e1.y * e2.x;
  const float a2 = (cx * cx + cy * cy) + cz * cz;
  if(!(a2 > 0.0f))
    return 0.0f;
  const float len = sqrtf(l1 > l2 ? l1 : l2);
  const float invSin = sqrtf((l1 * l2) / a2);
  if(!(invSin > 8.0f))
    return 0.0f;  // corners wider than ~7 degrees: inside the box tests' own margins
  const float sl = (9.5367431640625e-07f * len) * invSin;
  return sl < len ? sl : len;
}

// ------------------------------------------------------------------------------------------
// Full-sweep SAH BVH2, <= maxLeaf triangles per leaf (SURVEY 8d canonical tree).
// ------------------------------------------------------------------------------------------
struct Builder
{
  orc_scene& s;
  uint32_t maxLeaf;
  std::vector<Aabb> tb;       // per-triangle bounds
  std::vector<V3> cen;        // per-triangle centroid
  std::vector<float> rightArea;
  explicit Builder(orc_scene& sc, uint32_t ml) : s(sc), maxLeaf(ml) {}

  static void grow(Aabb& a, const Aabb& b)
  {
    for(int k = 0; k < 3; k++) { a.lo[k] = fminf(a.lo[k], b.lo[k]); a.hi[k] = fmaxf(a.hi[k], b.hi[k]); }
  }
  static Aabb empty()
  {
    Aabb a;
    for(int k = 0; k < 3; k++) { a.lo[k] = INFINITY; a.hi[k] = -INFINITY; }
    return a;
  }
  static float area(const Aabb& a)
  {
    float dx = a.hi[0] - a.lo[0], dy = a.hi[1] - a.lo[1], dz = a.hi[2] - a.lo[2];
    if(!(dx >= 0) || !(dy >= 0) || !(dz >= 0))
      return 0.f;
    return 2.f * (dx * dy + dy * dz + dz * dx);
  }
  Aabb boundsOf(uint32_t first, uint32_t count)
  {
    Aabb a = empty();
    for(uint32_t k = 0; k < count; k++) grow(a, tb[s.triOrder[first + k]]);
    return a;
  }
  // returns child ref; depth tracking for stats
  int32_t build(uint32_t first, uint32_t count, const Aabb& box, uint32_t depth, double& cost)
  {
    s.maxDepth = std::max(s.maxDepth, depth);
    float pa = area(box);
    auto makeLeaf = [&]() {
      s.leaves.push_back(BvhLeaf{first, count});
      cost = (double)count;  // intersect cost 1 per triangle
      return (int32_t) ~(int32_t)(s.leaves.size() - 1);
    };
    if(count == 1)
      return makeLeaf();
    // full sweep over the three axes
    float bestCost = INFINITY;
    int bestAxis = -1;
    uint32_t bestSplit = 0;
    uint32_t* ord = &s.triOrder[first];
    for(int axis = 0; axis < 3; axis++)
    {
      std::sort(ord, ord + count, [&](uint32_t a, uint32_t b) {
        float ca = (&cen[a].x)[axis], cb = (&cen[b].x)[axis];
        return ca < cb || (ca == cb && a < b);
      });
      Aabb acc = empty();
      for(uint32_t k = count - 1; k > 0; k--)
      {
        grow(acc, tb[ord[k]]);
        rightArea[k] = area(acc);
      }
      acc = empty();
      for(uint32_t k = 1; k < count; k++)
      {
        grow(acc, tb[ord[k - 1]]);
        float cst = area(acc) * (float)k + rightArea[k] * (float)(count - k);
        if(cst < bestCost) { bestCost = cst; bestAxis = axis; bestSplit = k; }
      }
    }
    float splitCost = (pa > 0.f) ? 1.0f + bestCost / pa : INFINITY;
    if(count <= maxLeaf && !(splitCost < (float)count))
      return makeLeaf();
    if(bestAxis < 0 || !(bestCost < INFINITY)) { bestAxis = 0; bestSplit = count / 2; }
    std::sort(ord, ord + count, [&](uint32_t a, uint32_t b) {
      float ca = (&cen[a].x)[bestAxis], cb = (&cen[b].x)[bestAxis];
      return ca < cb || (ca == cb && a < b);
    });
    int32_t me = (int32_t)s.nodes.size();
    s.nodes.push_back(BvhNode{});
    Aabb b0 = boundsOf(first, bestSplit);
    Aabb b1 = boundsOf(first + bestSplit, count - bestSplit);
    double c0 = 0, c1 = 0;
    int32_t ch0 = build(first, bestSplit, b0, depth + 1, c0);
    int32_t ch1 = build(first + bestSplit, count - bestSplit, b1, depth + 1, c1);
    BvhNode& n = s.nodes[me];
    n.box[0] = b0; n.box[1] = b1; n.child[0] = ch0; n.child[1] = ch1;
    double p = pa > 0 ? pa : 1.0;
    cost = 1.0 + (area(b0) * c0 + area(b1) * c1) / p;
    return me;
  }
  void run()
  {
    uint32_t n = (uint32_t)s.tris.size();
    s.nodes.clear(); s.leaves.clear(); s.triOrder.resize(n); s.maxDepth = 0;
    tb.resize(n); cen.resize(n); rightArea.resize(n + 1);
    Aabb all = empty();
    for(uint32_t i = 0; i < n; i++)
    {
      const Tri& t = s.tris[i];
      V3 a = t.v0, b = t.v0 + t.e1, c = t.v0 + t.e2;
      Aabb bb;
      bb.lo[0] = fminf(fminf(a.x, fminf(b.x, c.x)), fminf(t.p1.x, t.p2.x)); bb.hi[0] = fmaxf(fmaxf(a.x, fmaxf(b.x, c.x)), fmaxf(t.p1.x, t.p2.x));
      bb.lo[1] = fminf(fminf(a.y, fminf(b.y, c.y)), fminf(t.p1.y, t.p2.y)); bb.hi[1] = fmaxf(fmaxf(a.y, fmaxf(b.y, c.y)), fmaxf(t.p1.y, t.p2.y));
      bb.lo[2] = fminf(fminf(a.z, fminf(b.z, c.z)), fminf(t.p1.z, t.p2.z)); bb.hi[2] = fmaxf(fmaxf(a.z, fmaxf(b.z, c.z)), fmaxf(t.p1.z, t.p2.z));
      const float slop = triSlop(t.e1, t.e2);  // the reach of the binary32 triangle test outside the triangle (needles; see triSlop)
      for(int k = 0; k < 3; k++) { bb.lo[k] -= slop; bb.hi[k] += slop; }
      tb[i] = bb;
      cen[i] = v3(0.5f * (bb.lo[0] + bb.hi[0]), 0.5f * (bb.lo[1] + bb.hi[1]), 0.5f * (bb.lo[2] + bb.hi[2]));
      s.triOrder[i] = i;
      grow(all, bb);
    }
    s.rootIsLeaf = false;
    if(n == 0)
      return;
    double cost = 0;
    int32_t root = build(0, n, all, 0, cost);
    s.sahCost = cost;
    if(root < 0)
      s.rootIsLeaf = true;  // whole scene in leaf 0
  }
};

// ------------------------------------------------------------------------------------------
// Texture fetch: bilinear, REPEAT, LOD 0 (hello_vulkan.cpp:448-454; rchit has no derivatives)
// ------------------------------------------------------------------------------------------
struct V4 { float x, y, z, w; };
inline int wrapi(int i, int n)
{
  int m = i % n;
  return m < 0 ? m + n : m;
}
inline V4 texelOf(const orc_scene& s, bool srgb, const uint8_t* rgba, uint32_t w, int x, int y)
{
  const uint8_t* p = &rgba[((size_t)y * w + x) * 4];
  V4 r;
  if(srgb) { r.x = s.srgb_lut[p[0]]; r.y = s.srgb_lut[p[1]]; r.z = s.srgb_lut[p[2]]; }
  else { r.x = (float)p[0] / 255.0f; r.y = (float)p[1] / 255.0f; r.z = (float)p[2] / 255.0f; }
  r.w = (float)p[3] / 255.0f;
  return r;
}
// one bilinear REPEAT tap in one level of a texture
V4 sampleLevel(const orc_scene& s, const Tex& tx, uint32_t level, float u, float v)
{
  const uint32_t w = level ? tx.mips[level].w : tx.w, h = level ? tx.mips[level].h : tx.h;
  const uint8_t* data = level ? tx.mips[level].rgba.data() : tx.rgba.data();
  float fx = u * (float)w - 0.5f;
  float fy = v * (float)h - 0.5f;
  if(!(fabsf(fx) < 1.0e9f)) fx = 0.0f;
  if(!(fabsf(fy) < 1.0e9f)) fy = 0.0f;
  float flx = floorf(fx), fly = floorf(fy);
  float ax = fx - flx, ay = fy - fly;
  int x0 = wrapi((int)flx, (int)w), x1 = wrapi((int)flx + 1, (int)w);
  int y0 = wrapi((int)fly, (int)h), y1 = wrapi((int)fly + 1, (int)h);
  V4 t00 = texelOf(s, tx.srgb, data, w, x0, y0), t10 = texelOf(s, tx.srgb, data, w, x1, y0);
  V4 t01 = texelOf(s, tx.srgb, data, w, x0, y1), t11 = texelOf(s, tx.srgb, data, w, x1, y1);
  float bx = 1.0f - ax, by = 1.0f - ay;
  V4 r;
  r.x = (t00.x * bx + t10.x * ax) * by + (t01.x * bx + t11.x * ax) * ay;
  r.y = (t00.y * bx + t10.y * ax) * by + (t01.y * bx + t11.y * ax) * ay;
  r.z = (t00.z * bx + t10.z * ax) * by + (t01.z * bx + t11.z * ax) * ay;
  r.w = (t00.w * bx + t10.w * ax) * by + (t01.w * bx + t11.w * ax) * ay;
  return r;
}
V4 sampleTex(const orc_scene& s, int texIndex, float u, float v, Counters& c)
{
  c.tex_taps++;
  if(s.tex.empty() || texIndex < 0 || texIndex >= (int)s.tex.size())
    return V4{1, 1, 1, 1};  // 1x1 white dummy (hello_vulkan.cpp:468-472)
  return sampleLevel(s, s.tex[texIndex], 0, u, v);
}

// ---- implicit-LOD texture() of the fragment shader (hybrid G-buffer only) ------------------------------------------------
// The sampler of hello_vulkan.cpp:448-454: linear min / mag / mip filters over the whole chain (maxLod = FLT_MAX), anisotropy
// enabled with maxAnisotropy 4.  Restated from the Vulkan 1.3 specification, chapter "Image Operations" (Scale Factor
// Operation, LOD Operation, Texel Anisotropic Filtering): the exact LOD and tap placement of a GPU are implementation
// defined inside the bounds the spec gives, so this pins OUR definition (shared with the HIP sampler), not NVIDIA's.
struct TexGrad
{
  float dudx, dvdx, dudy, dvdy;  // dFdx / dFdy of the normalised texture coordinates
};
// log2 by exponent extraction and the atanh series (same operation sequence as the device: no libm call on either side)
inline float lodLog2(float x)
{
  uint32_t b;
  memcpy(&b, &x, 4);
  int e = (int)((b >> 23) & 255u) - 127;
  const uint32_t mb = (b & 0x007fffffu) | 0x3f800000u;
  float m;
  memcpy(&m, &mb, 4);
  if(m > 1.41421356f)
  {
    m = m * 0.5f;
    e += 1;
  }
  const float q = (m - 1.0f) / (m + 1.0f), q2 = q * q;
  const float series = 1.0f + q2 * (0.333333333f + q2 * (0.2f + q2 * (0.142857143f + q2 * 0.111111111f)));
  return (float)e + (2.0f * q * series) * 1.44269504f;
}
V4 sampleTexGrad(const orc_scene& s, int texIndex, float u, float v, const TexGrad& g, Counters& c)
{
  c.tex_taps++;
  if(s.tex.empty() || texIndex < 0 || texIndex >= (int)s.tex.size())
    return V4{1, 1, 1, 1};
  const Tex& tx = s.tex[texIndex];
  const uint32_t levels = (uint32_t)tx.mips.size();
  if(levels <= 1u)
    return sampleLevel(s, tx, 0, u, v);
  // scale factors in texel units of level 0
  const float fw = (float)tx.w, fh = (float)tx.h;
  const float mxu = g.dudx * fw, mxv = g.dvdx * fh, myu = g.dudy * fw, myv = g.dvdy * fh;
  const float rx2 = mxu * mxu + mxv * mxv, ry2 = myu * myu + myv * myv;
  const bool majorX = rx2 >= ry2;
  const float rmax = sqrtf(majorX ? rx2 : ry2), rmin = sqrtf(majorX ? ry2 : rx2);
  float eta = 1.0f;  // anisotropy ratio, at most maxAnisotropy = 4
  if(rmin > 0.0f)
    eta = glsl_min(rmax / rmin, 4.0f);
  else if(rmax > 0.0f)
    eta = 4.0f;
  if(!(eta >= 1.0f))
    eta = 1.0f;
  const int N = (int)ceilf(eta);
  const float scale = rmax / eta;
  float lambda = 0.0f;
  if(scale > 1.0f && scale < 3.0e38f)
    lambda = lodLog2(scale);
  else if(!(scale <= 1.0f))
    lambda = (float)(levels - 1u);
  lambda = glsl_clamp(lambda, 0.0f, (float)(levels - 1u));
  const float fl = floorf(lambda), delta = lambda - fl;
  const uint32_t hi = (uint32_t)fl, lo = std::min(hi + 1u, levels - 1u);
  const float du = majorX ? g.dudx : g.dudy, dv = majorX ? g.dvdx : g.dvdy;
  V4 r{0, 0, 0, 0};
  for(int i = 1; i <= N; i++)
  {
    const float o = (float)i / (float)(N + 1) - 0.5f;
    float tu = u + o * du, tv = v + o * dv;
    if(!(fabsf(tu) < 1.0e9f) || !(fabsf(tv) < 1.0e9f))
    {
      tu = u;
      tv = v;
    }
    const V4 a = sampleLevel(s, tx, hi, tu, tv), b = sampleLevel(s, tx, lo, tu, tv);
    r.x = r.x + (a.x * (1.0f - delta) + b.x * delta);
    r.y = r.y + (a.y * (1.0f - delta) + b.y * delta);
    r.z = r.z + (a.z * (1.0f - delta) + b.z * delta);
    r.w = r.w + (a.w * (1.0f - delta) + b.w * delta);
  }
  const float inv = 1.0f / (float)N;
  r.x = r.x * inv; r.y = r.y * inv; r.z = r.z * inv; r.w = r.w * inv;
  return r;
}

// vkCmdBlitImage with VK_FILTER_LINEAR from level L - 1 to level L (what nvvk::cmdGenerateMipmaps records per level): the
// destination texel centre maps to ((x + 0.5) * srcW / dstW, ...) in the source, bilinear around it with clamp-to-edge; an sRGB
// image is filtered on its decoded values and re-encoded.  For even sizes this is the 2x2 box filter.
inline float srgbEncode(float c) { return c <= 0.0031308f ? 12.92f * c : 1.055f * powf(c, 1.0f / 2.4f) - 0.055f; }
void buildMipChain(const orc_scene& s, Tex& tx)
{
  tx.mips.clear();
  tx.mips.push_back(TexLevel{tx.w, tx.h, {}});
  uint32_t w = tx.w, h = tx.h;
  const uint8_t* src = tx.rgba.data();
  while(w > 1 || h > 1)
  {
    TexLevel L;
    L.w = std::max(1u, w / 2); L.h = std::max(1u, h / 2);
    L.rgba.resize((size_t)L.w * L.h * 4);
    for(uint32_t y = 0; y < L.h; y++)
      for(uint32_t x = 0; x < L.w; x++)
      {
        const float fu = ((float)x + 0.5f) * ((float)w / (float)L.w) - 0.5f, fv = ((float)y + 0.5f) * ((float)h / (float)L.h) - 0.5f;
        const float flx = floorf(fu), fly = floorf(fv), ax = fu - flx, ay = fv - fly;
        const int x0 = std::clamp((int)flx, 0, (int)w - 1), x1 = std::clamp((int)flx + 1, 0, (int)w - 1);
        const int y0 = std::clamp((int)fly, 0, (int)h - 1), y1 = std::clamp((int)fly + 1, 0, (int)h - 1);
        const V4 t00 = texelOf(s, tx.srgb, src, w, x0, y0), t10 = texelOf(s, tx.srgb, src, w, x1, y0);
        const V4 t01 = texelOf(s, tx.srgb, src, w, x0, y1), t11 = texelOf(s, tx.srgb, src, w, x1, y1);
        const float c00[4] = {t00.x, t00.y, t00.z, t00.w}, c10[4] = {t10.x, t10.y, t10.z, t10.w};
        const float c01[4] = {t01.x, t01.y, t01.z, t01.w}, c11[4] = {t11.x, t11.y, t11.z, t11.w};
        for(int k = 0; k < 4; k++)
        {
          float val = (c00[k] * (1.0f - ax) + c10[k] * ax) * (1.0f - ay) + (c01[k] * (1.0f - ax) + c11[k] * ax) * ay;
          if(tx.srgb && k < 3)
            val = srgbEncode(val);
          val = val < 0.0f ? 0.0f : (val > 1.0f ? 1.0f : val);
          L.rgba[((size_t)y * L.w + x) * 4 + k] = (uint8_t)(val * 255.0f + 0.5f);
        }
      }
    w = L.w; h = L.h;
    tx.mips.push_back(std::move(L));
    src = tx.mips.back().rgba.data();
  }
}

// ------------------------------------------------------------------------------------------
// gltf.glsl:26-154
// ------------------------------------------------------------------------------------------
struct ShadeCtx
{
  const orc_scene& s;
  Counters& c;
  const TexGrad* grad = nullptr;  // non-NULL: fragment-shader texture() with these derivatives (hybrid G-buffer); NULL: LOD 0 (ray tracing stages)
};
inline V4 sampleTex(ShadeCtx& cx, int texIndex, float u, float v)
{
  return cx.grad ? sampleTexGrad(cx.s, texIndex, u, v, *cx.grad, cx.c) : sampleTex(cx.s, texIndex, u, v, cx.c);
}
inline V3 matBase(const GltfPBRMaterial& m) { return v3(m.pbrBaseColorFactor[0], m.pbrBaseColorFactor[1], m.pbrBaseColorFactor[2]); }
// gltf.glsl:26-32
V3 pbrGetBaseColor(ShadeCtx& cx, const GltfPBRMaterial& mat, float tu, float tv)
{
  V3 color = matBase(mat);
  if(mat.pbrBaseColorTexture > -1)
  {
    V4 t = sampleTex(cx, mat.pbrBaseColorTexture, tu, tv);
    color = color * v3(t.x, t.y, t.z);
  }
  return color;
}
// gltf.glsl:34-45
void pbrGetMetallicRoughness(ShadeCtx& cx, const GltfPBRMaterial& mat, float tu, float tv, float& metallic, float& roughness)
{
  metallic = mat.metallicFactor;
  roughness = mat.roughnessFactor;
  if(mat.metallicRoughnessTexture > -1)
  {
    V4 t = sampleTex(cx, mat.metallicRoughnessTexture, tu, tv);
    roughness *= t.y;
    metallic *= t.z;
  }
}
// gltf.glsl:55-66
float getNDF_GGXTR(V3 N, V3 H, float alpha)
{
  float a2 = alpha * alpha;
  float NH = dot(N, H);
  if(NH <= 0.0f)
    return 0.0f;
  float NH2 = NH * NH;
  float d = NH2 * (a2 - 1.0f) + 1.0f;
  return a2 * M_INV_PI_F / (d * d + 1e-4f);
}
// gltf.glsl:68-71
float getG_SchlickGGX(float NV, float k) { return NV / (NV * (1.0f - k) + k); }
// gltf.glsl:73-78
float getG_Smith(V3 N, V3 V, V3 L, float k)
{
  float NV = fabsf(dot(N, V));
  float NL = fabsf(dot(N, L));
  return getG_SchlickGGX(NV, k) * getG_SchlickGGX(NL, k);
}
// gltf.glsl:80-83
V3 getF_Schlick(V3 H, V3 V, V3 F0)
{
  return F0 + (v3(1.0f) - F0) * vk_pow5(1.0f - fabsf(dot(H, V)));
}
// gltf.glsl:85-96
V3 getSpecularBRDF_Cook_Torrance(V3 N, V3 H, V3 V, V3 L, V3 F0, float roughness)
{
  float alpha = roughness * roughness;
  float k = (roughness + 1.0f) * (roughness + 1.0f) / 8.0f;
  float D = getNDF_GGXTR(N, H, alpha);
  float G = getG_Smith(N, V, L, k);
  V3 F = getF_Schlick(H, V, F0);
  float down = 4.0f * fabsf(dot(V, N)) * fabsf(dot(L, N)) + 1e-4f;
  return D * F * G / down;
}
// gltf.glsl:98-109
V3 getSpecularBRDF_over_pdf_Cook_Torrance(V3 N, V3 H, V3 V, V3 L, V3 F0, float roughness, float ratio)
{
  float k = (roughness + 1.0f) * (roughness + 1.0f) / 8.0f;
  float pdf = (1.0f - ratio) * dot(N, H) / (4.0f * dot(L, H) + 1e-4f);
  float G = getG_Smith(N, V, L, k);
  V3 F = getF_Schlick(H, V, F0);
  float down = 4.0f * fabsf(dot(V, N)) * fabsf(dot(L, N)) + 1e-4f;
  return (F * G / down) / pdf;
}
// gltf.glsl:111-134 (pbrGetEmissive at :116 is dead: its result is unused)
V3 computePBR_BRDF(ShadeCtx& cx, V3 N, V3 V, V3 L, V3 H, const GltfPBRMaterial& mat, float tu, float tv)
{
  V3 baseColor = pbrGetBaseColor(cx, mat, tu, tv);
  float metalness, roughness;
  pbrGetMetallicRoughness(cx, mat, tu, tv, metalness, roughness);
  V3 F0 = v3(0.04f);
  F0 = glsl_mix(F0, baseColor, metalness);
  V3 F = getF_Schlick(H, V, F0);
  V3 f_cook_torrance = getSpecularBRDF_Cook_Torrance(N, H, V, L, F0, roughness);
  V3 kD = v3(1.0f) - F;
  kD = kD * (1.0f - metalness);
  V3 f_lambert = baseColor * M_INV_PI_F;
  V3 diffuse = kD * f_lambert;
  return diffuse + f_cook_torrance;
}
// gltf.glsl:136-154.  Non-point lights: returns 0 and leaves Li/cosTheta undefined in the
// GLSL; here Li = 0, cosTheta = 0.
V3 directLight(ShadeCtx& cx, const GltfLight& light, V3 P, V3 N, V3 V, const GltfPBRMaterial& mat, float tu, float tv, V3& Li, float& cosTheta)
{
  Li = v3(0.0f);
  cosTheta = 0.0f;
  if(light.type == 0)
  {
    V3 Ldir = v3(light.position[0], light.position[1], light.position[2]) - P;
    float d = length(Ldir);
    V3 L = Ldir / d;
    V3 H = normalize(L + V);
    float attenuation = d * d;
    Li = v3(light.color[0], light.color[1], light.color[2]) * light.intensity / attenuation;
    cosTheta = glsl_max(dot(L, N), 0.0f);
    if(cosTheta > 0.0f)
      return computePBR_BRDF(cx, N, V, L, H, mat, tu, tv);
  }
  return v3(0.0f);
}

// raycommon.glsl:8-19
struct Payload
{
  V3 hitValue;
  uint32_t seed;
  uint32_t depth;
  V3 rayOrigin, rayDirection, weight;
  bool isSpecular;
  float lightDist;
  V3 shadowRayDir;
};

// Surface attributes after interpolation/transform (raytrace.rchit:68-79).
struct Surface
{
  V3 worldPos, worldNrm, worldTag, worldBin;
  float tu, tv;
};

// raytrace.rchit:81-219, everything after the attribute fetch.
void shadeSurface(ShadeCtx& cx, const PushConstantRay& pc, const GltfPBRMaterial& mat, const Surface& sf, V3 worldRayDir, Payload& prd)
{
  cx.c.hits++;
  V3 emittance = v3(0.0f);
  if(prd.depth == 0 || prd.isSpecular)
  {
    emittance = v3(mat.emissiveFactor[0], mat.emissiveFactor[1], mat.emissiveFactor[2]);
    if(mat.emissiveTexture > -1)
    {
      V4 t = sampleTex(cx.s, mat.emissiveTexture, sf.tu, sf.tv, cx.c);
      emittance = emittance * v3(t.x, t.y, t.z);
    }
  }
  V3 tangent = sf.worldTag, binormal = sf.worldBin;
  V3 texNormal = sf.worldNrm;
  // rchit:98 ffnormal is computed and never used.
  if(mat.normalTexture > -1)
  {
    V4 t = sampleTex(cx.s, mat.normalTexture, sf.tu, sf.tv, cx.c);
    texNormal = normalize(v3(t.x, t.y, t.z) * 2.0f - v3(1.0f));
    // TBN * n, TBN = mat3(tangent, binormal, worldNrm) (rchit:99,103)
    texNormal = normalize(tangent * texNormal.x + binormal * texNormal.y + sf.worldNrm * texNormal.z);
    createCoordinateSystem(texNormal, tangent, binormal);
  }
  V3 baseColor = pbrGetBaseColor(cx, mat, sf.tu, sf.tv);
  float metalness, roughness;
  pbrGetMetallicRoughness(cx, mat, sf.tu, sf.tv, metalness, roughness);

  V3 rayOrigin = sf.worldPos;
  V3 rayDirection;
  float pdf;
  V3 BRDF;
  V3 V = normalize(-worldRayDir);
  V3 N = texNormal;

  float ratio = 0.5f * (1.0f - metalness);
  roughness = glsl_clamp(roughness, 0.01f, 0.99f);
  metalness = glsl_clamp(metalness, 0.01f, 0.99f);
  float r1 = rnd(prd.seed);
  if(r1 < ratio)
  {
    cx.c.diffuse_hits++;
    prd.isSpecular = false;
    int random_index = (int)(rnd(prd.seed) * (float)pc.lightsCount);
    const GltfLight& light = cx.s.lights[random_index];
    V3 lightDir = v3(light.position[0], light.position[1], light.position[2]) - sf.worldPos;
    float lightDistance = length(lightDir);
    V3 L = normalize(lightDir);
    prd.lightDist = lightDistance;
    prd.shadowRayDir = L;
    if(dot(L, texNormal) <= 0)
      emittance = emittance + v3(0.0f);
    else
    {
      V3 Li;
      float cosTheta;
      V3 brdf = directLight(cx, light, sf.worldPos, texNormal, V, mat, sf.tu, sf.tv, Li, cosTheta);
      emittance = emittance + (float)pc.lightsCount * brdf * Li * cosTheta;
    }
    rayDirection = normalize(samplingHemisphere(prd.seed, tangent, binormal, texNormal));
    pdf = ratio * dot(rayDirection, texNormal) * M_INV_PI_F;
    BRDF = (1.0f - metalness) * baseColor * M_INV_PI_F;
  }
  else
  {
    prd.isSpecular = true;
    float alpha = roughness * roughness;
    V3 h = samplingNDF_GGXTR(prd.seed, alpha * alpha);
    V3 H = normalize(tangent * h.x + binormal * h.y + texNormal * h.z);
    V3 L = normalize(glsl_reflect(-V, H));
    rayDirection = L;
    V3 F0 = v3(0.04f);
    F0 = glsl_mix(F0, baseColor, metalness);
    pdf = 1.0f;
    BRDF = getSpecularBRDF_over_pdf_Cook_Torrance(N, H, V, L, F0, roughness, ratio);
  }
  float cosTheta = dot(rayDirection, texNormal);
  prd.rayOrigin = rayOrigin;
  prd.rayDirection = rayDirection;
  prd.hitValue = emittance;
  prd.weight = BRDF * cosTheta / pdf;
}

// raytrace.rchit:34-79 attribute fetch + interpolation, then shadeSurface.
void closestHitShader(ShadeCtx& cx, const PushConstantRay& pc, const Hit& hit, V3 worldRayDir, Payload& prd)
{
  const orc_scene& s = cx.s;
  const Tri& tr = s.tris[hit.tri];
  const Instance& in = s.inst[tr.inst];
  const vkrt_prim_mesh& pinfo = s.pm[in.primMesh];
  uint32_t indexOffset = pinfo.firstIndex + 3 * tr.prim;
  uint32_t vertexOffset = pinfo.vertexOffset;
  uint32_t matIndex = (uint32_t)std::max(0, pinfo.materialIndex);
  uint32_t i0 = s.idx[indexOffset + 0] + vertexOffset;
  uint32_t i1 = s.idx[indexOffset + 1] + vertexOffset;
  uint32_t i2 = s.idx[indexOffset + 2] + vertexOffset;
  V3 b = v3(1.0f - hit.u - hit.v, hit.u, hit.v);
  V3 pos = s.pos[i0] * b.x + s.pos[i1] * b.y + s.pos[i2] * b.z;
  Surface sf;
  sf.worldPos = xformPoint(in, pos);
  V3 nrm = normalize(s.nrm[i0] * b.x + s.nrm[i1] * b.y + s.nrm[i2] * b.z);
  sf.worldNrm = normalize(xformNormal(in, nrm));
  V3 tg0 = v3(s.tan4[4 * i0], s.tan4[4 * i0 + 1], s.tan4[4 * i0 + 2]);
  V3 tg1 = v3(s.tan4[4 * i1], s.tan4[4 * i1 + 1], s.tan4[4 * i1 + 2]);
  V3 tg2 = v3(s.tan4[4 * i2], s.tan4[4 * i2 + 1], s.tan4[4 * i2 + 2]);
  V3 tag = normalize(tg0 * b.x + tg1 * b.y + tg2 * b.z);
  V3 worldTag = normalize(xformNormal(in, tag));
  worldTag = normalize(worldTag - dot(worldTag, sf.worldNrm) * sf.worldNrm);
  sf.worldTag = worldTag;
  sf.worldBin = s.tan4[4 * i0 + 3] * cross(sf.worldNrm, worldTag);
  sf.tu = (s.uv2[2 * i0] * b.x + s.uv2[2 * i1] * b.y) + s.uv2[2 * i2] * b.z;
  sf.tv = (s.uv2[2 * i0 + 1] * b.x + s.uv2[2 * i1 + 1] * b.y) + s.uv2[2 * i2 + 1] * b.z;
  shadeSurface(cx, pc, s.mats[matIndex], sf, worldRayDir, prd);
}

// raytrace.rmiss:11-19
inline void missShader(const PushConstantRay& pc, Payload& prd)
{
  if(prd.depth == 0)
    prd.hitValue = v3(pc.clearColor[0], pc.clearColor[1], pc.clearColor[2]) * 0.8f;
  else
    prd.hitValue = v3(0.01f);
  prd.depth = 100;
}

inline void mat4MulVec4(const vkrt_mat4& M, const float v[4], float out[4])
{
  for(int i = 0; i < 4; i++)
    out[i] = ((M.m[0 + i] * v[0] + M.m[4 + i] * v[1]) + M.m[8 + i] * v[2]) + M.m[12 + i] * v[3];
}

struct PathLog  // optional per-segment trace of one pixel (debug / parity localisation)
{
  std::vector<float>* out;
};

// raytrace.rgen:24-146 for one pixel.  Returns the value stored to the image (rgba).
void rayGen(const orc_scene& s, const PushConstantRay& pc, const GlobalUniforms& uni, uint32_t seedArg, uint32_t flags,
            uint32_t x, uint32_t y, uint32_t W, uint32_t H, bool useBvh, float* pixel /* in: old, out: new */,
            Counters& c, PathLog* log)
{
  ShadeCtx cx{s, c};
  c.pixels++;
  Payload prd;
  memset(&prd, 0, sizeof(prd));
  uint32_t index = (flags & VKRT_TRACE_SEED_INDEX_ROW_MAJOR) ? (y * W + x) : (y * x + x);  // rgen:27
  prd.seed = tea(index, seedArg);
  V3 hitValues = v3(0.0f);
  const float o4[4] = {0, 0, 0, 1};
  float origin[4];
  mat4MulVec4(uni.viewInverse, o4, origin);
  const float tMin = 0.001f, tMax = 10000.0f;
  for(int smpl = 0; smpl < pc.samples; smpl++)
  {
    float r1 = rnd(prd.seed);
    float r2 = rnd(prd.seed);
    float jx = pc.frame == 0 ? 0.5f : r1, jy = pc.frame == 0 ? 0.5f : r2;
    float pcx = (float)x + jx, pcy = (float)y + jy;
    float inU = pcx / (float)W, inV = pcy / (float)H;
    float dx = inU * 2.0f - 1.0f, dy = inV * 2.0f - 1.0f;
    const float d4[4] = {dx, dy, 1, 1};
    float target[4];
    mat4MulVec4(uni.projInverse, d4, target);
    V3 tn = normalize(v3(target[0], target[1], target[2]));
    const float t4[4] = {tn.x, tn.y, tn.z, 0};
    float direction[4];
    mat4MulVec4(uni.viewInverse, t4, direction);

    prd.hitValue = v3(0.0f);
    prd.rayOrigin = v3(origin[0], origin[1], origin[2]);
    prd.rayDirection = v3(direction[0], direction[1], direction[2]);
    prd.depth = 0;
    prd.weight = v3(0.0f);
    V3 curWeight = v3(1.0f);
    V3 hitValue = v3(0.0f);
    for(; prd.depth < (uint32_t)pc.depth; prd.depth++)
    {
      c.rays_closest++;
      V3 rd = prd.rayDirection;
      Hit h = useBvh ? closest_bvh(s, prd.rayOrigin, rd, tMin, tMax, c, prd.seed) : closest_brute(s, prd.rayOrigin, rd, tMin, tMax, c, prd.seed);
      if(log)
      {
        float ray[8] = {-2.0f, prd.rayOrigin.x, prd.rayOrigin.y, prd.rayOrigin.z, rd.x, rd.y, rd.z, tMax};
        log->out->insert(log->out->end(), ray, ray + 8);
        float rec[8] = {(float)prd.depth, (float)h.tri, h.t, h.u, h.v, 0, 0, 0};
        log->out->insert(log->out->end(), rec, rec + 8);
      }
      if(h.tri >= 0)
        closestHitShader(cx, pc, h, rd, prd);
      else
        missShader(pc, prd);
      bool shadowHit = false;
      if(!prd.isSpecular && prd.depth != 100)  // rgen:79
      {
        c.rays_shadow++;
        float smax = prd.lightDist - 0.1f;
        shadowHit = useBvh ? any_bvh(s, prd.rayOrigin, prd.shadowRayDir, tMin, smax, c, prd.seed)
                           : any_brute(s, prd.rayOrigin, prd.shadowRayDir, tMin, smax, c, prd.seed);
        if(log)
        {
          float ray[8] = {-3.0f, prd.rayOrigin.x, prd.rayOrigin.y, prd.rayOrigin.z, prd.shadowRayDir.x, prd.shadowRayDir.y, prd.shadowRayDir.z, smax};
          log->out->insert(log->out->end(), ray, ray + 8);
        }
      }
      if(!shadowHit)  // rgen:99-102
      {
        V3 q = prd.hitValue * curWeight;
        hitValue = hitValue + v3(glsl_min(q.x, 10.0f), glsl_min(q.y, 10.0f), glsl_min(q.z, 10.0f));
      }
      if(log)
      {
        float rec[8] = {-1.0f, shadowHit ? 1.0f : 0.0f, prd.hitValue.x, prd.hitValue.y, prd.hitValue.z, prd.weight.x, prd.weight.y, prd.weight.z};
        log->out->insert(log->out->end(), rec, rec + 8);
      }
      curWeight = curWeight * prd.weight;  // rgen:115
    }
    hitValues = hitValues + hitValue;
  }
  V3 res = hitValues / (float)pc.samples;  // rgen:120
  if(pc.frame > 0)                          // rgen:136-141
  {
    float a = 1.0f / (float)(pc.frame + 1);
    V3 old = v3(pixel[0], pixel[1], pixel[2]);
    V3 m = glsl_mix(old, res, a);
    pixel[0] = m.x; pixel[1] = m.y; pixel[2] = m.z; pixel[3] = 1.0f;
  }
  else
  {
    pixel[0] = res.x; pixel[1] = res.y; pixel[2] = res.z; pixel[3] = 1.0f;
  }
}

// ------------------------------------------------------------------------------------------
// Hybrid mode (BASELINE config 5).
//   G-buffer: the reference rasterises vert_shader.vert / frag_shader.frag into 4 planes
//   (hello_vulkan.cpp:583-615; attachments :690-734); with no raster hardware path from HIP the same
//   planes come from a primary ray cast through each pixel centre evaluating the same shader math
//   (cull mode NONE hello_vulkan.cpp:180; clear values main.cpp:482-487).  Textures are sampled at LOD 0
//   (the raster path would use mips/anisotropy: documented difference).
//   Hybrid ray-gen: raytraceHybrid.rgen:50-303.
// ------------------------------------------------------------------------------------------
// rg16f store of (roughness, metalness) (raytraceHybrid.rgen:20, hello_vulkan.cpp:704): float -> half (RNE) -> float
inline float quantizeHalf(float f)
{
  uint32_t x;
  memcpy(&x, &f, 4);
  const uint32_t sign = x & 0x80000000u;
  uint32_t ax = x & 0x7fffffffu;
  if(ax >= 0x7f800000u)
    return f;  // inf / nan unchanged
  if(ax < 0x38800000u)
  {  // below the smallest normal half (2^-14): quantise to multiples of 2^-24 (half subnormals), RNE
    const uint32_t e = ax >> 23;
    if(e < 101u)  // < 2^-26 -> rounds to zero (ties at 2^-25 go to even = 0)
      ax = 0u;
    else
    {
      const uint32_t mant = (ax & 0x7fffffu) | 0x800000u;
      const uint32_t shift = 126u - e;             // 14..25
      uint32_t q = mant >> shift;                   // units of 2^-24
      const uint32_t rem = mant & ((1u << shift) - 1u), half = 1u << (shift - 1u);
      if(rem > half || (rem == half && (q & 1u))) q++;
      float r = (float)q * 5.9604644775390625e-8f;  // q * 2^-24, exact
      memcpy(&ax, &r, 4);
    }
  }
  else
  {
    const uint32_t rem = ax & 0x1fffu;
    ax &= ~0x1fffu;
    if(rem > 0x1000u || (rem == 0x1000u && (ax & 0x2000u))) ax += 0x2000u;
    if(ax >= 0x47800000u) ax = 0x7f800000u;  // overflow to infinity
  }
  const uint32_t out = sign | ax;
  float r;
  memcpy(&r, &out, 4);
  return r;
}

struct GbufPixel
{
  float color[4], position[4], normal[4], rough[2];
  float nrdNormRough[4], nrdViewZ;  // NRD front-end planes of the raster pass (frag_shader.frag:133-136); filled when viewMatrix != NULL
};

// ---- NRD / REBLUR front-end packing (gltf.glsl:156-273), written out with the profile's operation order -----------------
inline float stepf(float edge, float x) { return x < edge ? 0.0f : 1.0f; }
// gltf.glsl:157-165 _NRD_EncodeUnitVector(v, false)
inline void nrdEncodeUnitVector(V3 v, float out[2])
{
  const float n = (std::fabs(v.x) * 1.0f + std::fabs(v.y) * 1.0f) + std::fabs(v.z) * 1.0f;  // dot(abs(v), vec3(1))
  v = v / n;
  const float wx = (1.0f - std::fabs(v.y)) * (stepf(0.0f, v.x) * 2.0f - 1.0f);
  const float wy = (1.0f - std::fabs(v.x)) * (stepf(0.0f, v.y) * 2.0f - 1.0f);
  const float ex = v.z >= 0.0f ? v.x : wx, ey = v.z >= 0.0f ? v.y : wy;
  out[0] = ex * 0.5f + 0.5f;
  out[1] = ey * 0.5f + 0.5f;
}
// UNORM store of an rgb10_a2 attachment (hello_vulkan.cpp:690-741 eInNormRough): round to nearest even step
inline float quantizeUnorm(float x, float levels)
{
  const float c = glsl_clamp(x, 0.0f, 1.0f);
  return std::nearbyintf(c * levels) / levels;
}
// gltf.glsl:167-177 NRD_FrontEnd_PackNormalAndRoughness + the rgb10_a2 store
inline void nrdPackNormalRoughness(V3 N, float roughness, float materialID, float out[4])
{
  float e[2];
  nrdEncodeUnitVector(N, e);
  out[0] = quantizeUnorm(e[0], 1023.0f);
  out[1] = quantizeUnorm(e[1], 1023.0f);
  out[2] = quantizeUnorm(roughness, 1023.0f);
  out[3] = quantizeUnorm(glsl_clamp(materialID / 3.0f, 0.0f, 1.0f), 3.0f);
}

// direction of the primary ray through the centre of pixel (x, y) (raytrace.rgen:46-51 with jitter 0.5)
V3 pixelCentreDir(const GlobalUniforms& uni, uint32_t x, uint32_t y, uint32_t W, uint32_t H)
{
  const float inU = ((float)x + 0.5f) / (float)W, inV = ((float)y + 0.5f) / (float)H;
  const float d4[4] = {inU * 2.0f - 1.0f, inV * 2.0f - 1.0f, 1, 1};
  float target[4];
  mat4MulVec4(uni.projInverse, d4, target);
  const V3 tn = normalize(v3(target[0], target[1], target[2]));
  const float t4[4] = {tn.x, tn.y, tn.z, 0};
  float direction[4];
  mat4MulVec4(uni.viewInverse, t4, direction);
  return v3(direction[0], direction[1], direction[2]);
}
// fragTexCoord of a (helper) fragment at another pixel of the same primitive: the rasteriser's perspective-correct
// interpolation is the barycentric position where that pixel's view ray meets the triangle's plane.
bool texCoordOnPlane(V3 org, V3 dir, const V3 p[3], const float tcU[3], const float tcV[3], float& u, float& v)
{
  const V3 e1 = p[1] - p[0], e2 = p[2] - p[0];
  const V3 pvec = cross(dir, e2);
  const float det = dot(e1, pvec);
  if(det == 0.0f)
    return false;
  const V3 tvec = org - p[0];
  const float bu = dot(tvec, pvec) / det;
  const V3 qvec = cross(tvec, e1);
  const float bv = dot(dir, qvec) / det;
  const float b0 = 1.0f - bu - bv;
  u = tcU[0] * b0 + tcU[1] * bu + tcU[2] * bv;
  v = tcV[0] * b0 + tcV[1] * bu + tcV[2] * bv;
  return true;
}

// vert_shader.vert:60-74 per vertex, barycentric interpolation (what the rasteriser does), frag_shader.frag:122-214
void gbufferPixel(const orc_scene& s, const float clearColor[4], int lightsCount, const GlobalUniforms& uni, uint32_t x, uint32_t y, uint32_t W,
                  uint32_t H, bool useBvh, GbufPixel& out, Counters& c, const float* viewMatrix = nullptr)
{
  out.nrdNormRough[0] = out.nrdNormRough[1] = out.nrdNormRough[2] = out.nrdNormRough[3] = 0.0f;  // main.cpp:488-491 clear values
  out.nrdViewZ = 0.0f;
  ShadeCtx cx{s, c};
  for(int k = 0; k < 4; k++) out.color[k] = clearColor[k];            // main.cpp:483
  out.position[0] = out.position[1] = out.position[2] = 0.0f; out.position[3] = 1.0f;   // main.cpp:485
  out.normal[0] = out.normal[1] = out.normal[2] = 0.0f; out.normal[3] = 1.0f;           // main.cpp:486
  out.rough[0] = out.rough[1] = 0.0f;                                                   // main.cpp:487
  // primary ray through the pixel centre (same camera as raytrace.rgen:46-51 with jitter 0.5)
  const float o4[4] = {0, 0, 0, 1};
  float origin[4];
  mat4MulVec4(uni.viewInverse, o4, origin);
  const float inU = ((float)x + 0.5f) / (float)W, inV = ((float)y + 0.5f) / (float)H;
  const float d4[4] = {inU * 2.0f - 1.0f, inV * 2.0f - 1.0f, 1, 1};
  float target[4];
  mat4MulVec4(uni.projInverse, d4, target);
  const V3 tn = normalize(v3(target[0], target[1], target[2]));
  const float t4[4] = {tn.x, tn.y, tn.z, 0};
  float direction[4];
  mat4MulVec4(uni.viewInverse, t4, direction);
  const V3 org = v3(origin[0], origin[1], origin[2]), dir = v3(direction[0], direction[1], direction[2]);
  c.rays_closest++;
  // (the raster pass this ray stands for has no any-hit stage and no alpha test: every triangle is opaque here)
  const Hit h = useBvh ? closest_bvh(s, org, dir, 0.001f, 10000.0f, c, 0u, false) : closest_brute(s, org, dir, 0.001f, 10000.0f, c, 0u, false);
  if(h.tri < 0)
    return;
  const Tri& tr = s.tris[h.tri];
  const Instance& in = s.inst[tr.inst];
  const vkrt_prim_mesh& pm = s.pm[in.primMesh];
  const uint32_t base = pm.firstIndex + 3 * tr.prim;
  const uint32_t vi[3] = {s.idx[base] + pm.vertexOffset, s.idx[base + 1] + pm.vertexOffset, s.idx[base + 2] + pm.vertexOffset};
  const GltfPBRMaterial& mat = s.mats[(size_t)std::max(0, pm.materialIndex)];  // pcRaster.materialId (hello_vulkan.cpp:608)
  const float bw[3] = {1.0f - h.u - h.v, h.u, h.v};
  V3 wPos = v3(0.0f), wNrm = v3(0.0f), wTag = v3(0.0f), wBin = v3(0.0f);
  float tu = 0.0f, tv = 0.0f;
  V3 corner[3];
  float cornerU[3], cornerV[3];
  for(int k = 0; k < 3; k++)
  {
    const uint32_t i = vi[k];
    const V3 p = xformPoint(in, s.pos[i]);                       // modelMatrix * position
    corner[k] = p; cornerU[k] = s.uv2[2 * i]; cornerV[k] = s.uv2[2 * i + 1];
    const V3 n = normalize(xformNormal(in, s.nrm[i]));           // mat3(inverseTranspose) * normal
    V3 t = normalize(xformNormal(in, v3(s.tan4[4 * i], s.tan4[4 * i + 1], s.tan4[4 * i + 2])));
    t = normalize(t - dot(t, n) * n);
    const V3 b = cross(n, t) * s.tan4[4 * i + 3];
    wPos = wPos + p * bw[k]; wNrm = wNrm + n * bw[k]; wTag = wTag + t * bw[k]; wBin = wBin + b * bw[k];
    tu = tu + s.uv2[2 * i] * bw[k];
    tv = tv + s.uv2[2 * i + 1] * bw[k];
  }
  // Implicit derivatives of fragTexCoord for texture(): dFdx / dFdy are differences between the two fragments of the pixel's
  // 2x2 quad in that direction (Vulkan spec, "Derivative Operations"); the quad neighbour runs on the same primitive.
  TexGrad grad{0.0f, 0.0f, 0.0f, 0.0f};
  if(s.gbufferMips)
  {
    float nu, nv;
    if(texCoordOnPlane(org, pixelCentreDir(uni, x ^ 1u, y, W, H), corner, cornerU, cornerV, nu, nv))
    {
      const float sgn = (x & 1u) ? -1.0f : 1.0f;
      grad.dudx = (nu - tu) * sgn; grad.dvdx = (nv - tv) * sgn;
    }
    if(texCoordOnPlane(org, pixelCentreDir(uni, x, y ^ 1u, W, H), corner, cornerU, cornerV, nu, nv))
    {
      const float sgn = (y & 1u) ? -1.0f : 1.0f;
      grad.dudy = (nu - tu) * sgn; grad.dvdy = (nv - tv) * sgn;
    }
    cx.grad = &grad;
  }
  const V3 viewDir = wPos - org;
  // frag_shader.frag:96-119 getNormal
  V3 N = normalize(wNrm);
  if(mat.normalTexture > -1)
  {
    V3 T = normalize(wTag), B = normalize(wBin);
    T = normalize(T - dot(T, N) * N);
    B = normalize(B - dot(B, N) * N - dot(B, T) * T);
    const V4 tx = sampleTex(cx, mat.normalTexture, tu, tv);
    V3 nrm = v3(tx.x, tx.y, tx.z) * 2.0f - v3(1.0f);
    nrm = normalize(nrm);
    nrm = normalize(T * nrm.x + B * nrm.y + N * nrm.z);
    N = nrm;
  }
  const V3 baseColor = pbrGetBaseColor(cx, mat, tu, tv);
  float metalness, roughness;
  pbrGetMetallicRoughness(cx, mat, tu, tv, metalness, roughness);
  const V3 albedo = (1.0f - metalness) * baseColor;
  const V3 V = normalize(-viewDir);
  V3 color = v3(0.0f);
  V3 emittance = v3(mat.emissiveFactor[0], mat.emissiveFactor[1], mat.emissiveFactor[2]);
  if(mat.emissiveTexture > -1)
  {
    const V4 tx = sampleTex(cx, mat.emissiveTexture, tu, tv);
    emittance = emittance * v3(tx.x, tx.y, tx.z);
  }
  for(int i = 0; i < lightsCount; i++)  // frag_shader.frag:193-213 (every light, no shadow test)
  {
    const GltfLight& light = s.lights[(size_t)i];
    const V3 lp = v3(light.position[0], light.position[1], light.position[2]);
    V3 L = normalize(lp - wPos);
    V3 lightIntensity = v3(light.color[0], light.color[1], light.color[2]) * light.intensity;
    if(light.type == 0)
    {
      const V3 lDir = lp - wPos;
      const float d = length(lDir);
      lightIntensity = lightIntensity / (d * d);
    }
    else
      L = normalize(lp);
    const V3 Hh = normalize(L + V);
    const float cosTheta = glsl_max(dot(L, N), 0.0f);
    if(cosTheta > 0.0f)
      color = color + computePBR_BRDF(cx, N, V, L, Hh, mat, tu, tv) * lightIntensity * cosTheta;
  }
  const V3 oc = emittance + color;
  out.color[0] = oc.x; out.color[1] = oc.y; out.color[2] = oc.z; out.color[3] = albedo.x;
  out.position[0] = wPos.x; out.position[1] = wPos.y; out.position[2] = wPos.z; out.position[3] = albedo.y;
  out.normal[0] = N.x; out.normal[1] = N.y; out.normal[2] = N.z; out.normal[3] = albedo.z;
  out.rough[0] = quantizeHalf(roughness);
  out.rough[1] = quantizeHalf(metalness);
  if(viewMatrix)
  {  // frag_shader.frag:134-135: NRD normal / roughness / material plane (rgb10_a2) and view-space depth (r16f)
    nrdPackNormalRoughness(N, roughness, (float)pm.materialIndex, out.nrdNormRough);
    const float w4[4] = {wPos.x, wPos.y, wPos.z, 1.0f};
    float vz[4];
    vkrt_mat4 VM;
    memcpy(VM.m, viewMatrix, sizeof VM.m);
    mat4MulVec4(VM, w4, vz);
    out.nrdViewZ = quantizeHalf(vz[2]);
  }
}

// raytraceHybrid.rgen:50-303 for one pixel; accum = imageAccum texel (in/out)
void hybridPixel(const orc_scene& s, const PushConstantRay& pc, const GlobalUniforms& uni, uint32_t seedArg, uint32_t flags, uint32_t x, uint32_t y,
                 uint32_t W, bool useBvh, const GbufPixel& g, float* accum, Counters& c, float* nrdRadHitD = nullptr)
{
  ShadeCtx cx{s, c};
  c.pixels++;
  Payload prd;
  memset(&prd, 0, sizeof(prd));
  const uint32_t index = (flags & VKRT_TRACE_SEED_INDEX_ROW_MAJOR) ? (y * W + x) : (y * x + x);  // rgen:55
  prd.seed = tea(index, seedArg);
  float color[4] = {0.0f, 0.0f, 0.0f, 1.0f};
  const V3 worldPos = v3(g.position[0], g.position[1], g.position[2]);
  const V3 worldNrm = v3(g.normal[0], g.normal[1], g.normal[2]);
  auto accumulateFrames = [&]() {  // rgen:36-48
    if(pc.frame > 0)
    {
      const float a = 1.0f / (float)(pc.frame + 1);
      for(int k = 0; k < 4; k++) accum[k] = accum[k] * (1.0f - a) + color[k] * a;
    }
    else
      for(int k = 0; k < 4; k++) accum[k] = color[k];
  };
  if(worldPos.x == 0.0f && worldPos.y == 0.0f && worldPos.z == 0.0f && worldNrm.x == 0.0f && worldNrm.y == 0.0f && worldNrm.z == 0.0f)
  {
    accumulateFrames();
    return;
  }
  const V3 albedo = v3(g.color[3], g.position[3], g.normal[3]);
  const float roughness = g.rough[0], metalness = g.rough[1];
  auto anyHit = [&](V3 o, V3 d, float tmin, float tmax) {
    c.rays_shadow++;
    return useBvh ? any_bvh(s, o, d, tmin, tmax, c, prd.seed) : any_brute(s, o, d, tmin, tmax, c, prd.seed);
  };
  if(pc.useShadows == 1)  // rgen:81-131
  {
    float visibility = 1.0f;
    const int random_index = (int)(rnd(prd.seed) * (float)pc.lightsCount);
    const GltfLight& light = s.lights[(size_t)random_index];
    const V3 lightDir = v3(light.position[0], light.position[1], light.position[2]) - worldPos;
    const float lightDistance = length(lightDir);
    const V3 L = normalize(lightDir);
    if(dot(L, worldNrm) < 0.0f)
      visibility = 0.0f;
    else if(anyHit(worldPos, L, 0.1f, lightDistance - 0.1f))
      visibility = 0.0f;
    visibility = glsl_max(visibility, 0.01f);
    color[3] *= visibility;
  }
  if(pc.useAO == 1)  // rgen:134-169
  {
    float ao = 0.0f;
    V3 tangent, binormal;
    createCoordinateSystem(worldNrm, tangent, binormal);
    const float weightAo = 1.0f / 4;
    for(int i = 0; i < 4; i++)
    {
      const V3 rayDir = normalize(samplingHemisphere(prd.seed, tangent, binormal, worldNrm));
      if(anyHit(worldPos, rayDir, 0.1f, 2.0f))
        ao += weightAo;
    }
    color[3] *= (1.0f - ao);
  }
  if(pc.useGI == 1)  // rgen:172-282
  {
    V3 hitValues = v3(0.0f);
    float hitDists = 0.0f;
    const float tMin = 0.001f, tMax = 10000.0f;
    V3 direction, curWeight;
    const float ratio = metalness * (1.0f - roughness);
    if(ratio < 0.8f)
    {
      prd.isSpecular = false;
      V3 tangent, binormal;
      createCoordinateSystem(worldNrm, tangent, binormal);
      direction = normalize(samplingHemisphere(prd.seed, tangent, binormal, worldNrm));
      curWeight = albedo;
    }
    else
    {
      prd.isSpecular = true;
      const float o4[4] = {0, 0, 0, 1};
      float cam[4];
      mat4MulVec4(uni.viewInverse, o4, cam);
      const V3 V = normalize(v3(cam[0], cam[1], cam[2]) - worldPos);
      direction = normalize(glsl_reflect(-V, worldNrm));
      curWeight = v3(1.0f);
    }
    prd.hitValue = v3(0.0f);
    prd.rayOrigin = worldPos;
    prd.rayDirection = direction;
    prd.depth = 1;
    prd.weight = v3(0.0f);
    V3 hitValue = v3(0.0f);
    for(; prd.depth < (uint32_t)pc.depth; prd.depth++)
    {
      c.rays_closest++;
      const V3 rd = prd.rayDirection;
      const Hit h = useBvh ? closest_bvh(s, prd.rayOrigin, rd, tMin, tMax, c, prd.seed) : closest_brute(s, prd.rayOrigin, rd, tMin, tMax, c, prd.seed);
      if(h.tri >= 0)
        closestHitShader(cx, pc, h, rd, prd);
      else
        missShader(pc, prd);
      bool shadowHit = false;
      if(!prd.isSpecular && prd.depth != 100)
        shadowHit = anyHit(prd.rayOrigin, prd.shadowRayDir, tMin, prd.lightDist - 0.1f);
      if(!shadowHit)
      {
        const V3 q = prd.hitValue * curWeight;
        hitValue = hitValue + v3(glsl_min(q.x, 10.0f), glsl_min(q.y, 10.0f), glsl_min(q.z, 10.0f));
      }
      if(prd.depth == 1 && !prd.isSpecular)  // rgen:253-264
        hitDists = shadowHit ? 0.5f * prd.lightDist : prd.lightDist;
      curWeight = curWeight * prd.weight;
    }
    hitValues = hitValues + hitValue;
    color[0] = hitValues.x; color[1] = hitValues.y; color[2] = hitValues.z;
    if(nrdRadHitD)
    {  // rgen:273-281: REBLUR front end with hitDistParams (3, 1, 20, -25); o_diffRadianceHitD is rgba16f
      const float viewZ = g.nrdViewZ;
      const float t = glsl_clamp(exp2f(-25.0f * roughness * roughness), 0.0f, 1.0f);
      const float f = (3.0f + std::fabs(viewZ) * 1.0f) * (1.0f * (1.0f - t) + 20.0f * t);  // gltf.glsl:253-257
      float normHitDist = glsl_clamp(hitDists / f, 0.0f, 1.0f);                          // :259-264
      V3 rad = hitValues;                                                               // :219-238 with sanitize = true
      const bool bad = std::isnan(rad.x) || std::isnan(rad.y) || std::isnan(rad.z) || std::isinf(rad.x) || std::isinf(rad.y) || std::isinf(rad.z);
      rad = bad ? v3(0.0f) : v3(glsl_clamp(rad.x, 0.0f, 65504.0f), glsl_clamp(rad.y, 0.0f, 65504.0f), glsl_clamp(rad.z, 0.0f, 65504.0f));
      normHitDist = (std::isnan(normHitDist) || std::isinf(normHitDist)) ? 0.0f : glsl_clamp(normHitDist, 0.0f, 1.0f);
      if(normHitDist != 0.0f)
        normHitDist = glsl_max(normHitDist, 1e-7f);
      const float Y = (rad.x * 0.25f + rad.y * 0.5f) + rad.z * 0.25f;                   // :206-213 _NRD_LinearToYCoCg
      const float Co = (rad.x * 0.5f + rad.y * 0.0f) + rad.z * -0.5f;
      const float Cg = (rad.x * -0.25f + rad.y * 0.5f) + rad.z * -0.25f;
      nrdRadHitD[0] = quantizeHalf(Y); nrdRadHitD[1] = quantizeHalf(Co); nrdRadHitD[2] = quantizeHalf(Cg); nrdRadHitD[3] = quantizeHalf(normHitDist);
    }
  }
  accumulateFrames();
}

thread_local char g_err[256] = "";

}  // namespace

// =============================================================================================
// C interface for the Python test harness (ctypes)
// =============================================================================================
extern "C" {

const char* orc_last_error() { return g_err; }

uint32_t orc_tea(uint32_t a, uint32_t b) { return tea(a, b); }
uint32_t orc_lcg(uint32_t* state) { return lcg(*state); }
float orc_rnd(uint32_t* state) { return rnd(*state); }

orc_scene* orc_scene_create(const vkrt_scene_desc* d)
{
  if(!d || d->struct_size != sizeof(vkrt_scene_desc))
  {
    snprintf(g_err, sizeof g_err, "bad scene desc");
    return nullptr;
  }
  if(d->material_count == 0 || d->light_count == 0)
  {
    snprintf(g_err, sizeof g_err, "scene needs >=1 material and >=1 light");
    return nullptr;
  }
  orc_scene* s = new orc_scene();
  s->pos.resize(d->vertex_count); s->nrm.resize(d->vertex_count);
  memcpy(s->pos.data(), d->positions, sizeof(float) * 3 * d->vertex_count);
  memcpy(s->nrm.data(), d->normals, sizeof(float) * 3 * d->vertex_count);
  s->tan4.assign(d->tangents, d->tangents + 4 * (size_t)d->vertex_count);
  s->uv2.assign(d->texcoords0, d->texcoords0 + 2 * (size_t)d->vertex_count);
  s->idx.assign(d->indices, d->indices + d->index_count);
  s->pm.assign(d->prim_meshes, d->prim_meshes + d->prim_mesh_count);
  s->mats.assign(d->materials, d->materials + d->material_count);
  s->lights.assign(d->lights, d->lights + d->light_count);
  for(int i = 0; i < 256; i++)
  {
    float c = (float)i / 255.0f;
    s->srgb_lut[i] = (c <= 0.04045f) ? c / 12.92f : powf((c + 0.055f) / 1.055f, 2.4f);
  }
  for(uint32_t t = 0; t < d->texture_count; t++)
  {
    Tex tx;
    tx.w = d->textures[t].width; tx.h = d->textures[t].height; tx.srgb = d->textures[t].is_srgb != 0;
    tx.rgba.assign(d->textures[t].rgba8, d->textures[t].rgba8 + (size_t)tx.w * tx.h * 4);
    buildMipChain(*s, tx);
    s->tex.push_back(std::move(tx));
  }
  uint32_t gid = 0;
  for(uint32_t n = 0; n < d->node_count; n++)
  {
    Instance in;
    const float* m = d->nodes[n].worldMatrix;
    for(int r = 0; r < 3; r++)
      for(int c = 0; c < 4; c++)
        in.o2w[r][c] = m[c * 4 + r];
    invert3x3(in.o2w, in.w2o);
    in.primMesh = d->nodes[n].primMesh;
    if(in.primMesh < 0 || (uint32_t)in.primMesh >= d->prim_mesh_count)
    {
      snprintf(g_err, sizeof g_err, "node %u: primMesh out of range", n);
      delete s;
      return nullptr;
    }
    s->inst.push_back(in);
    const vkrt_prim_mesh& pm = s->pm[in.primMesh];
    for(uint32_t p = 0; p < pm.indexCount / 3; p++)
    {
      uint32_t i0 = s->idx[pm.firstIndex + 3 * p + 0] + pm.vertexOffset;
      uint32_t i1 = s->idx[pm.firstIndex + 3 * p + 1] + pm.vertexOffset;
      uint32_t i2 = s->idx[pm.firstIndex + 3 * p + 2] + pm.vertexOffset;
      V3 a = xformPoint(in, s->pos[i0]), b = xformPoint(in, s->pos[i1]), c = xformPoint(in, s->pos[i2]);
      Tri t;
      t.v0 = a; t.e1 = b - a; t.e2 = c - a; t.p1 = b; t.p2 = c;
      t.gid = gid++; t.inst = n; t.prim = p;
      s->tris.push_back(t);
    }
  }
  return s;
}
void orc_scene_destroy(orc_scene* s) { delete s; }
uint32_t orc_triangle_count(const orc_scene* s) { return (uint32_t)s->tris.size(); }

// Full-sweep SAH BVH2 with <= max_leaf triangles per leaf (canonical accounting tree).
int orc_build_bvh(orc_scene* s, uint32_t max_leaf)
{
  Builder b(*s, max_leaf < 1 ? 1 : max_leaf);
  b.run();
  return 0;
}
void orc_bvh_info(const orc_scene* s, uint32_t* nodes, uint32_t* leaves, uint32_t* maxDepth, double* sah)
{
  *nodes = (uint32_t)s->nodes.size(); *leaves = (uint32_t)s->leaves.size(); *maxDepth = s->maxDepth; *sah = s->sahCost;
}
// World-space triangles as (v0,e1,e2) float[9] + (gid,inst,prim) u32[3], for builder tests.
void orc_get_triangles(const orc_scene* s, float* v9, uint32_t* ids3)
{
  for(size_t i = 0; i < s->tris.size(); i++)
  {
    const Tri& t = s->tris[i];
    float* o = v9 + 9 * i;
    o[0] = t.v0.x; o[1] = t.v0.y; o[2] = t.v0.z; o[3] = t.e1.x; o[4] = t.e1.y; o[5] = t.e1.z; o[6] = t.e2.x; o[7] = t.e2.y; o[8] = t.e2.z;
    ids3[3 * i] = t.gid; ids3[3 * i + 1] = t.inst; ids3[3 * i + 2] = t.prim;
  }
}

/* Render rows[0..nrows) of a full_w x full_h launch into the compact buffer `out`
 * (nrows x full_w x rgba32f, in/out when pc->frame > 0).  use_bvh 0 = brute force.
 * threads <= 0: hardware concurrency.  counters (8 x u64, vkrt_counters order) optional. */
int orc_render_rows(const orc_scene* s, const PushConstantRay* pc, const GlobalUniforms* cam, uint32_t seed, uint32_t flags,
                    uint32_t full_w, uint32_t full_h, const uint32_t* rows, uint32_t nrows, float* out, int use_bvh, int threads,
                    uint64_t* counters)
{
  if(use_bvh && s->nodes.empty() && !s->rootIsLeaf && !s->tris.empty())
  {
    snprintf(g_err, sizeof g_err, "BVH not built");
    return 1;
  }
  if(pc->lightsCount < 0 || (uint32_t)pc->lightsCount > s->lights.size())
  {
    snprintf(g_err, sizeof g_err, "lightsCount out of range");
    return 1;
  }
  int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
  if(nt < 1) nt = 1;
  // Work is dealt dynamically in chunks of 32 pixels of a row (a row of a 1080p frame costs 0.03-20 s depending on what it sees: whole
  // rows per thread left the slowest row as the wall time of a many-thread sample).  Every thread counts into a Counters of its own on
  // its own stack and hands it over once at the end: the per-thread blocks used to sit side by side in one vector and were incremented
  // per node visit, so neighbouring threads fought over cache lines (256 threads came out slower than 16).
  const uint32_t CHUNK = 32;
  const uint32_t perRow = (full_w + CHUNK - 1) / CHUNK;
  const uint64_t nchunks = (uint64_t)nrows * perRow;
  std::atomic<uint64_t> next{0};
  std::vector<Counters> cs(nt);
  auto work = [&](int tid) {
    Counters c;
    for(;;)
    {
      const uint64_t k = next.fetch_add(1, std::memory_order_relaxed);
      if(k >= nchunks) break;
      const uint32_t r = (uint32_t)(k / perRow), x0 = (uint32_t)(k % perRow) * CHUNK, x1 = std::min(full_w, x0 + CHUNK);
      const uint32_t y = rows[r];
      for(uint32_t x = x0; x < x1; x++)
        rayGen(*s, *pc, *cam, seed, flags, x, y, full_w, full_h, use_bvh != 0, out + ((size_t)r * full_w + x) * 4, c, nullptr);
    }
    cs[tid] = c;
  };
  if(nt == 1)
    work(0);
  else
  {
    std::vector<std::thread> th;
    for(int t = 0; t < nt; t++) th.emplace_back(work, t);
    for(auto& t : th) t.join();
  }
  if(counters)
  {
    Counters tot;
    for(auto& c : cs) tot.add(c);
    counters[0] = tot.rays_closest; counters[1] = tot.rays_shadow; counters[2] = tot.hits; counters[3] = tot.diffuse_hits;
    counters[4] = tot.tex_taps; counters[5] = tot.pixels; counters[6] = tot.nodes_visited; counters[7] = tot.tris_tested;
  }
  return 0;
}

/* Per-segment log of one pixel: records of 8 floats: {-2, o.xyz, d.xyz, tmax} closest ray, {depth, gid, t, u, v, 0,0,0} its
 * hit, {-3, o.xyz, d.xyz, tmax} shadow ray (if traced), {-1, shadowHit, hitValue.xyz, weight.xyz} segment.  Returns count of floats. */
int orc_pixel_log(const orc_scene* s, const PushConstantRay* pc, const GlobalUniforms* cam, uint32_t seed, uint32_t flags,
                  uint32_t full_w, uint32_t full_h, uint32_t x, uint32_t y, int use_bvh, float* rgba_inout, float* log, int log_cap)
{
  std::vector<float> v;
  PathLog pl{&v};
  Counters c;
  rayGen(*s, *pc, *cam, seed, flags, x, y, full_w, full_h, use_bvh != 0, rgba_inout, c, &pl);
  int n = (int)std::min<size_t>(v.size(), (size_t)log_cap);
  memcpy(log, v.data(), sizeof(float) * n);
  return (int)v.size();
}

/* Closest / any hit for a batch of rays (for builder and traversal tests). */
int orc_trace_rays(const orc_scene* s, uint32_t n, const float* o, const float* d, float tmin, float tmax, int any_hit, int use_bvh,
                   float* t, float* u, float* v, int32_t* gid, uint64_t* counters)
{
  Counters c;
  for(uint32_t i = 0; i < n; i++)
  {
    V3 ro = v3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), rd = v3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    if(any_hit)
    {
      bool h = use_bvh ? any_bvh(*s, ro, rd, tmin, tmax, c) : any_brute(*s, ro, rd, tmin, tmax, c);
      gid[i] = h ? 0 : -1; t[i] = 0; u[i] = 0; v[i] = 0;
    }
    else
    {
      Hit h = use_bvh ? closest_bvh(*s, ro, rd, tmin, tmax, c) : closest_brute(*s, ro, rd, tmin, tmax, c);
      gid[i] = h.tri; t[i] = h.t; u[i] = h.u; v[i] = h.v;
    }
  }
  if(counters) { counters[6] = c.nodes_visited; counters[7] = c.tris_tested; }
  return 0;
}

/* op: 0 sin, 1 cos, 2 sqrt, 3 a/b, 4 pow5, 5 x of normalize((a,b,0)) */
void orc_eval_math(int op, uint32_t n, const float* a, const float* b, float* out)
{
  for(uint32_t i = 0; i < n; i++)
  {
    switch(op)
    {
      case 0: out[i] = vk_sin(a[i]); break;
      case 1: out[i] = vk_cos(a[i]); break;
      case 2: out[i] = sqrtf(a[i]); break;
      case 3: out[i] = a[i] / b[i]; break;
      case 4: out[i] = vk_pow5(a[i]); break;
      default: out[i] = normalize(v3(a[i], b[i], 0.0f)).x; break;
    }
  }
}

/* Shading spot-evaluation for cross-checking with oracle/np_shading.py.
 * in : per item 40 floats:
 *   [0:3] worldPos [3:6] worldNrm [6:9] tangent [9:12] binormal [12:15] worldRayDir
 *   [15:19] baseColorFactor [19] metallic [20] roughness [21:24] emissive
 *   [24:27] light.position [27:30] light.color [30] intensity [31] light.type
 *   [32] seed(bits) [33] depth(bits) [34] isSpecular(bits) [35] lightsCount(bits) [36:40] pad
 * out: per item 20 floats:
 *   [0:3] hitValue [3:6] rayOrigin [6:9] rayDirection [9:12] weight [12] isSpecular
 *   [13] lightDist [14:17] shadowRayDir [17] seed(bits) [18:20] pad
 */
void orc_eval_shade(uint32_t n, const float* in, float* out)
{
  orc_scene s;
  s.lights.resize(1);
  Counters c;
  ShadeCtx cx{s, c};
  for(uint32_t i = 0; i < n; i++)
  {
    const float* p = in + 40 * (size_t)i;
    float* o = out + 20 * (size_t)i;
    Surface sf;
    sf.worldPos = v3(p[0], p[1], p[2]); sf.worldNrm = v3(p[3], p[4], p[5]); sf.worldTag = v3(p[6], p[7], p[8]);
    sf.worldBin = v3(p[9], p[10], p[11]); sf.tu = 0; sf.tv = 0;
    V3 rd = v3(p[12], p[13], p[14]);
    GltfPBRMaterial m;
    memset(&m, 0, sizeof m);
    for(int k = 0; k < 4; k++) m.pbrBaseColorFactor[k] = p[15 + k];
    m.metallicFactor = p[19]; m.roughnessFactor = p[20];
    for(int k = 0; k < 3; k++) m.emissiveFactor[k] = p[21 + k];
    m.pbrBaseColorTexture = m.metallicRoughnessTexture = m.normalTexture = m.emissiveTexture = -1;
    GltfLight L;
    for(int k = 0; k < 3; k++) { L.position[k] = p[24 + k]; L.color[k] = p[27 + k]; }
    L.intensity = p[30];
    uint32_t bits[4];
    memcpy(bits, p + 32, 16);
    memcpy(&L.type, p + 31, 4);
    PushConstantRay pc;
    memset(&pc, 0, sizeof pc);
    pc.lightsCount = (int32_t)bits[3];
    Payload prd;
    memset(&prd, 0, sizeof prd);
    prd.seed = bits[0]; prd.depth = bits[1]; prd.isSpecular = bits[2] != 0;
    // whichever index int(rnd*lightsCount) picks, it finds this record's light
    s.lights.assign((size_t)std::max(1, pc.lightsCount), L);
    shadeSurface(cx, pc, m, sf, rd, prd);
    o[0] = prd.hitValue.x; o[1] = prd.hitValue.y; o[2] = prd.hitValue.z;
    o[3] = prd.rayOrigin.x; o[4] = prd.rayOrigin.y; o[5] = prd.rayOrigin.z;
    o[6] = prd.rayDirection.x; o[7] = prd.rayDirection.y; o[8] = prd.rayDirection.z;
    o[9] = prd.weight.x; o[10] = prd.weight.y; o[11] = prd.weight.z;
    o[12] = prd.isSpecular ? 1.0f : 0.0f;
    o[13] = prd.lightDist;
    o[14] = prd.shadowRayDir.x; o[15] = prd.shadowRayDir.y; o[16] = prd.shadowRayDir.z;
    memcpy(o + 17, &prd.seed, 4);
    o[18] = 0; o[19] = 0;
  }
}

/* Camera ray of raytrace.rgen:30,42-51 for pixel (x,y), jitter (jx,jy): out = origin3, dir3. */
void orc_camera_ray(const GlobalUniforms* uni, uint32_t x, uint32_t y, uint32_t W, uint32_t H, float jx, float jy, float* out6)
{
  const float o4[4] = {0, 0, 0, 1};
  float origin[4];
  mat4MulVec4(uni->viewInverse, o4, origin);
  float pcx = (float)x + jx, pcy = (float)y + jy;
  float inU = pcx / (float)W, inV = pcy / (float)H;
  float dx = inU * 2.0f - 1.0f, dy = inV * 2.0f - 1.0f;
  const float d4[4] = {dx, dy, 1, 1};
  float target[4];
  mat4MulVec4(uni->projInverse, d4, target);
  V3 tn = normalize(v3(target[0], target[1], target[2]));
  const float t4[4] = {tn.x, tn.y, tn.z, 0};
  float direction[4];
  mat4MulVec4(uni->viewInverse, t4, direction);
  out6[0] = origin[0]; out6[1] = origin[1]; out6[2] = origin[2];
  out6[3] = direction[0]; out6[4] = direction[1]; out6[5] = direction[2];
}

/* G-buffer of rows[0..nrows): planes color/position/normal (nrows x W x 4 floats) and rough (nrows x W x 2). */
int orc_gbuffer_rows(const orc_scene* s, const float* clearColor, int lightsCount, const GlobalUniforms* cam, uint32_t full_w, uint32_t full_h,
                     const uint32_t* rows, uint32_t nrows, float* color, float* position, float* normal, float* rough, int use_bvh, int threads)
{
  if(lightsCount < 0 || (uint32_t)lightsCount > s->lights.size())
  {
    snprintf(g_err, sizeof g_err, "lightsCount out of range");
    return 1;
  }
  int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
  if(nt < 1) nt = 1;
  std::atomic<uint32_t> next{0};
  auto work = [&]() {
    Counters c;
    for(;;)
    {
      const uint32_t r = next.fetch_add(1);
      if(r >= nrows) break;
      for(uint32_t x = 0; x < full_w; x++)
      {
        GbufPixel g;
        gbufferPixel(*s, clearColor, lightsCount, *cam, x, rows[r], full_w, full_h, use_bvh != 0, g, c);
        const size_t p = (size_t)r * full_w + x;
        memcpy(color + 4 * p, g.color, 16); memcpy(position + 4 * p, g.position, 16); memcpy(normal + 4 * p, g.normal, 16);
        memcpy(rough + 2 * p, g.rough, 8);
      }
    }
  };
  std::vector<std::thread> th;
  for(int t = 1; t < nt; t++) th.emplace_back(work);
  work();
  for(auto& t : th) t.join();
  return 0;
}

/* The same raster-pass stand-in with the three NRD front-end attachments (frag_shader.frag:133-136): normRough (rgb10_a2 values),
 * viewZ (r16f values); viewMatrix = pcRaster.viewMatrix, column-major. */
int orc_gbuffer_rows_nrd(const orc_scene* s, const float* clearColor, int lightsCount, const GlobalUniforms* cam, const float* viewMatrix,
                         uint32_t full_w, uint32_t full_h, const uint32_t* rows, uint32_t nrows, float* color, float* position, float* normal,
                         float* rough, float* normRough, float* viewZ, int use_bvh)
{
  if(lightsCount < 0 || (uint32_t)lightsCount > s->lights.size())
  {
    snprintf(g_err, sizeof g_err, "lightsCount out of range");
    return 1;
  }
  Counters c;
  for(uint32_t r = 0; r < nrows; r++)
    for(uint32_t x = 0; x < full_w; x++)
    {
      GbufPixel g;
      gbufferPixel(*s, clearColor, lightsCount, *cam, x, rows[r], full_w, full_h, use_bvh != 0, g, c, viewMatrix);
      const size_t p = (size_t)r * full_w + x;
      memcpy(color + 4 * p, g.color, 16); memcpy(position + 4 * p, g.position, 16); memcpy(normal + 4 * p, g.normal, 16);
      memcpy(rough + 2 * p, g.rough, 8); memcpy(normRough + 4 * p, g.nrdNormRough, 16);
      viewZ[p] = g.nrdViewZ;
    }
  return 0;
}

/* raytraceHybrid.rgen incl. its REBLUR front end (rgen:273-281): viewZ in, radHitD (rgba16f values) out (written where GI ran). */
int orc_hybrid_rows_nrd(const orc_scene* s, const PushConstantRay* pc, const GlobalUniforms* cam, uint32_t seed, uint32_t flags, uint32_t full_w,
                        uint32_t full_h, const uint32_t* rows, uint32_t nrows, const float* color, const float* position, const float* normal,
                        const float* rough, const float* viewZ, float* accum, float* radHitD, int use_bvh)
{
  (void)full_h;
  if(pc->lightsCount < 0 || (uint32_t)pc->lightsCount > s->lights.size())
  {
    snprintf(g_err, sizeof g_err, "lightsCount out of range");
    return 1;
  }
  Counters c;
  for(uint32_t r = 0; r < nrows; r++)
    for(uint32_t x = 0; x < full_w; x++)
    {
      const size_t p = (size_t)r * full_w + x;
      GbufPixel g;
      memcpy(g.color, color + 4 * p, 16); memcpy(g.position, position + 4 * p, 16); memcpy(g.normal, normal + 4 * p, 16);
      memcpy(g.rough, rough + 2 * p, 8);
      g.nrdViewZ = viewZ[p];
      hybridPixel(*s, *pc, *cam, seed, flags, x, rows[r], full_w, use_bvh != 0, g, accum + 4 * p, c, radHitD + 4 * p);
    }
  return 0;
}

/* Ray/triangle test of every later query on this scene: 0 = Moeller-Trumbore (default), 1 = watertight (VKRT_OPT_WATERTIGHT).  The
 * tree need not be rebuilt (its boxes bound both vertex forms). */
void orc_set_watertight(orc_scene* s, int on) { s->watertight = on ? 1 : 0; }
/* Any-hit alpha / dissolve stage of every later query (VKRT_OPT_ANYHIT_DISSOLVE): 0 = all geometry opaque (default, as the reference runs). */
void orc_set_dissolve(orc_scene* s, int on) { s->dissolve = on ? 1 : 0; }

/* The rays raytraceHybrid.rgen traces for ONE pixel, in order (9 floats each: o.xyz, d.xyz, tmin, tmax, any-hit flag); gpix = the
 * pixel's G-buffer texels (color4, position4, normal4, rough2).  Returns the number of floats the full log has. */
int orc_hybrid_pixel_rays(const orc_scene* s, const PushConstantRay* pc, const GlobalUniforms* cam, uint32_t seed, uint32_t flags, uint32_t full_w,
                          uint32_t x, uint32_t y, const float* gpix, int use_bvh, float* accum4, float* rays, int cap)
{
  GbufPixel g;
  memcpy(g.color, gpix, 16); memcpy(g.position, gpix + 4, 16); memcpy(g.normal, gpix + 8, 16); memcpy(g.rough, gpix + 12, 8);
  std::vector<float> v;
  Counters c;
  g_rayTap = &v;
  hybridPixel(*s, *pc, *cam, seed, flags, x, y, full_w, use_bvh != 0, g, accum4, c);
  g_rayTap = nullptr;
  memcpy(rays, v.data(), sizeof(float) * std::min<size_t>(v.size(), (size_t)cap));
  return (int)v.size();
}

/* raytraceHybrid.rgen over rows[0..nrows) given the G-buffer planes of those rows; accum (nrows x W x 4) is in/out. */
int orc_hybrid_rows(const orc_scene* s, const PushConstantRay* pc, const GlobalUniforms* cam, uint32_t seed, uint32_t flags, uint32_t full_w,
                    uint32_t full_h, const uint32_t* rows, uint32_t nrows, const float* color, const float* position, const float* normal,
                    const float* rough, float* accum, int use_bvh, int threads, uint64_t* counters)
{
  (void)full_h;
  if(pc->lightsCount < 0 || (uint32_t)pc->lightsCount > s->lights.size())
  {
    snprintf(g_err, sizeof g_err, "lightsCount out of range");
    return 1;
  }
  int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
  if(nt < 1) nt = 1;
  std::atomic<uint32_t> next{0};
  std::vector<Counters> cs((size_t)nt);
  auto work = [&](int tid) {
    Counters& c = cs[(size_t)tid];
    for(;;)
    {
      const uint32_t r = next.fetch_add(1);
      if(r >= nrows) break;
      for(uint32_t x = 0; x < full_w; x++)
      {
        const size_t p = (size_t)r * full_w + x;
        GbufPixel g;
        memcpy(g.color, color + 4 * p, 16); memcpy(g.position, position + 4 * p, 16); memcpy(g.normal, normal + 4 * p, 16);
        memcpy(g.rough, rough + 2 * p, 8);
        hybridPixel(*s, *pc, *cam, seed, flags, x, rows[r], full_w, use_bvh != 0, g, accum + 4 * p, c);
      }
    }
  };
  std::vector<std::thread> th;
  for(int t = 1; t < nt; t++) th.emplace_back(work, t);
  work(0);
  for(auto& t : th) t.join();
  if(counters)
  {
    Counters tot;
    for(auto& c : cs) tot.add(c);
    counters[0] = tot.rays_closest; counters[1] = tot.rays_shadow; counters[2] = tot.hits; counters[3] = tot.diffuse_hits;
    counters[4] = tot.tex_taps; counters[5] = tot.pixels; counters[6] = tot.nodes_visited; counters[7] = tot.tris_tested;
  }
  return 0;
}

/* post.frag:36-58 composite + gamma for n pixels (rtMode 0 = hybrid composite, 1 = path tracer pass-through). */
void orc_post(int rtMode, int viewAccumulated, int useGI, uint32_t n, const float* mainImg, const float* rtImg, float* out)
{
  const float gamma = 1.0f / 2.2f;
  for(uint32_t i = 0; i < n; i++)
  {
    float m[4] = {mainImg[4 * i], mainImg[4 * i + 1], mainImg[4 * i + 2], mainImg[4 * i + 3]};
    if(rtMode == 0)
    {
      const float* r = rtImg + 4 * (size_t)i;
      if(viewAccumulated == 0) { m[0] = m[0] * r[3] + r[0]; m[1] = m[1] * r[3] + r[1]; m[2] = m[2] * r[3] + r[2]; m[3] = 1.0f; }
      else if(useGI == 1) { m[0] = r[0] * r[3]; m[1] = r[1] * r[3]; m[2] = r[2] * r[3]; }
      else { m[0] = m[1] = m[2] = r[3]; }
    }
    for(int k = 0; k < 4; k++) out[4 * (size_t)i + k] = powf(m[k], gamma);
  }
}

float orc_quantize_half(float f) { return quantizeHalf(f); }

/* 1 (default): the G-buffer's texture() uses implicit LOD + anisotropy; 0: LOD 0 (what the numpy restatement implements). */
void orc_set_gbuffer_mips(orc_scene* s, int on) { s->gbufferMips = on ? 1 : 0; }

/* Mip chain introspection: number of levels; size and RGBA8 texels of one level (rgba8 may be NULL to query the size). */
uint32_t orc_texture_levels(const orc_scene* s, int texIndex) { return (uint32_t)s->tex[(size_t)texIndex].mips.size(); }
void orc_texture_level(const orc_scene* s, int texIndex, uint32_t level, uint32_t* w, uint32_t* h, uint8_t* rgba8)
{
  const Tex& tx = s->tex[(size_t)texIndex];
  const uint32_t lw = level ? tx.mips[level].w : tx.w, lh = level ? tx.mips[level].h : tx.h;
  if(w) *w = lw;
  if(h) *h = lh;
  if(rgba8)
    memcpy(rgba8, level ? tx.mips[level].rgba.data() : tx.rgba.data(), (size_t)lw * lh * 4);
}
/* Implicit-LOD sampler spot check: uv = float[2n], grads = float[4n] (dudx, dvdx, dudy, dvdy) -> rgba float[4n]. */
void orc_sample_texture_grad(const orc_scene* s, int texIndex, uint32_t n, const float* uv, const float* grads, float* rgba)
{
  Counters c;
  for(uint32_t i = 0; i < n; i++)
  {
    const TexGrad g{grads[4 * i], grads[4 * i + 1], grads[4 * i + 2], grads[4 * i + 3]};
    V4 r = sampleTexGrad(*s, texIndex, uv[2 * i], uv[2 * i + 1], g, c);
    rgba[4 * i] = r.x; rgba[4 * i + 1] = r.y; rgba[4 * i + 2] = r.z; rgba[4 * i + 3] = r.w;
  }
}

/* Bilinear sampler spot check: uv = float[2n] -> rgba float[4n] from texture texIndex. */
void orc_sample_texture(const orc_scene* s, int texIndex, uint32_t n, const float* uv, float* rgba)
{
  Counters c;
  for(uint32_t i = 0; i < n; i++)
  {
    V4 r = sampleTex(*s, texIndex, uv[2 * i], uv[2 * i + 1], c);
    rgba[4 * i] = r.x; rgba[4 * i + 1] = r.y; rgba[4 * i + 2] = r.z; rgba[4 * i + 3] = r.w;
  }
}

}  // extern "C"
