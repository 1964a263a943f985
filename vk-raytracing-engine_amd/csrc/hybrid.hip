// hybrid.hip -- hybrid mode of the reference (BASELINE config 5):
//   k_gbuffer  stands in for the raster pass HelloVulkan::rasterizeGltf (hello_vulkan.cpp:583-615) with
//              vert_shader.vert:60-74 / frag_shader.frag:122-214: there is no raster path from HIP, so the
//              four planes raytraceHybrid.rgen reads are produced by a primary ray through each pixel centre
//              evaluating the same per-vertex and per-fragment math (cull NONE :180, clear values main.cpp:482-487).
//   k_hybrid   raytraceHybrid.rgen:50-303 (1 shadow ray, 4 AO rays, optional GI path reusing the path tracer's
//              closest-hit / miss shaders), accumulating into the rgba32f accumulation image (:36-48).
//   k_post     post.frag:36-58 composite (raster.rgb * rt.a + rt.rgb) and gamma 1/2.2.
// One thread per pixel; traversal and shading are the path tracer's (traverse*.h, shade.h).  With the wide layout k_hybrid runs one
// wave per 8x8 tile in lockstep so that its rays use the work-sharing traversal (traverse_share.h).
#include <hip/hip_runtime.h>

#include "device_math.h"
#include "device_scene.h"
#include "kernels.h"
#include "rgen.h"
#include "shade.h"
#include "traverse.h"
#include "traverse_wide.h"
#include "traverse_share.h"

#define HY_BLOCK 256

struct HybridParams
{
  TraceParams T;      // scene, pc, camera, launch geometry (image pointer unused)
  float4* color;      // eOutImage  rgba32f
  float4* position;   // ePosMap    rgba32f
  float4* normal;     // eNormMap   rgba32f
  float2* rough;      // eRoughMap  (rg16f-quantised values held as floats)
  float4* accum;      // eAccumMap  rgba32f
  float clearColor[4];
  int lightsCount;
  // optional NRD / REBLUR front-end attachments (frag_shader.frag:133-136, raytraceHybrid.rgen:273-281); NULL = not requested
  float4* nrdNormRough;  // eInNormRough rgb10_a2 values
  float* nrdViewZ;       // eInViewZ     r16f values
  float4* nrdRadHitD;    // eInRadHitD   rgba16f values
  float viewMatrix[16];  // pcRaster.viewMatrix (column-major), for viewZ
  uint2* giLater;        // k_hybrid: non-NULL = GI runs afterwards on the wavefront streams; (seed, visibility bits) per pixel go here
};

// ---- NRD / REBLUR front-end packing, gltf.glsl:156-273 (same operation order as the oracle) -----------------------------
VKRT_DEV float stepf(float edge, float x) { return x < edge ? 0.0f : 1.0f; }
VKRT_DEV float quantizeUnorm(float x, float levels) { return rintf(glsl_clamp(x, 0.0f, 1.0f) * levels) / levels; }  // rgb10_a2 store
VKRT_DEV float4 nrdPackNormalRoughness(f3 N, float roughness, float materialID)  // gltf.glsl:157-177
{
  const float n = (fabsf(N.x) * 1.0f + fabsf(N.y) * 1.0f) + fabsf(N.z) * 1.0f;
  const f3 v = N / n;
  const float wx = (1.0f - fabsf(v.y)) * (stepf(0.0f, v.x) * 2.0f - 1.0f);
  const float wy = (1.0f - fabsf(v.x)) * (stepf(0.0f, v.y) * 2.0f - 1.0f);
  const float ex = v.z >= 0.0f ? v.x : wx, ey = v.z >= 0.0f ? v.y : wy;
  return make_float4(quantizeUnorm(ex * 0.5f + 0.5f, 1023.0f), quantizeUnorm(ey * 0.5f + 0.5f, 1023.0f), quantizeUnorm(roughness, 1023.0f),
                     quantizeUnorm(glsl_clamp(materialID / 3.0f, 0.0f, 1.0f), 3.0f));
}

// Primary-ray direction of pixel (x, y), raytraceHybrid.rgen / the rasteriser's view of the same pixel centre
VKRT_DEV f3 primaryDir(const TraceParams& P, uint32_t x, uint32_t y)
{
  float target[4], direction[4];
  const float inU = ((float)x + 0.5f) / (float)P.fullW, inV = ((float)y + 0.5f) / (float)P.fullH;
  mat4MulVec4(P.projInverse, inU * 2.0f - 1.0f, inV * 2.0f - 1.0f, 1.0f, 1.0f, target);
  const f3 tn = normalize3(mk3(target[0], target[1], target[2]));
  mat4MulVec4(P.viewInverse, tn.x, tn.y, tn.z, 0.0f, direction);
  return mk3(direction[0], direction[1], direction[2]);
}
// Texture coordinates where the ray (org, dir) meets the plane of the triangle (p0, p1, p2): what the rasteriser's
// perspective-correct interpolation evaluates for a (helper) fragment of the same primitive.  false: ray parallel to the plane.
VKRT_DEV bool planeTexCoord(f3 org, f3 dir, f3 p0, f3 p1, f3 p2, const float* tcu, const float* tcv, float& u, float& v)
{
  const f3 e1 = p1 - p0, e2 = p2 - p0;
  const f3 pvec = cross3(dir, e2);
  const float det = dot3(e1, pvec);
  if(det == 0.0f)
    return false;
  const f3 tvec = org - p0;
  const float bu = dot3(tvec, pvec) / det;
  const f3 qvec = cross3(tvec, e1);
  const float bv = dot3(dir, qvec) / det;
  const float b0 = 1.0f - bu - bv;
  u = tcu[0] * b0 + tcu[1] * bu + tcu[2] * bv;
  v = tcv[0] * b0 + tcv[1] * bu + tcv[2] * bv;
  return true;
}

VKRT_DEV bool pixelOf(const TraceParams& P, uint32_t& x, uint32_t& y, uint32_t& lrow)
{
  const unsigned w = blockIdx.x * blockDim.x + threadIdx.x;  // tile-major, as in the path tracer
  if(w >= P.tileCount * 64u)
    return false;
  const unsigned tile = w >> 6, inTile = w & 63u;
  x = (tile % P.tilesX) * 8u + (inTile & 7u);
  lrow = (tile / P.tilesX) * 8u + (inTile >> 3);
  if(x >= P.fullW || lrow >= P.localRows)
    return false;
  y = globalRow(P, lrow);
  return y < P.fullH;
}

// TM: VKRT_TM_WATERTIGHT and / or VKRT_TM_MASKID (a raster pass has no any-hit stage -- every triangle is opaque here -- but on a scene
// built for the stage the id words carry its flag, which the tie rule "smallest triangle id" and the exclusive tmax must not see)
template <bool WIDE, int TM>
__global__ __launch_bounds__(HY_BLOCK) void k_gbuffer(const HybridParams H)
{
  extern __shared__ int lds_stack[];
  const TraceParams& P = H.T;
  const DevScene& sc = P.sc;
  uint32_t x, y, lrow;
  unsigned nClosest = 0;
  TravCount tc;
  __shared__ float lut[512];
  ShadeStats st;
  st.hits = 0; st.diffuse = 0; st.taps = 0;
  st.lut = ldsTexelLut(P.sc, lut);
  if(pixelOf(P, x, y, lrow))
  {
    const size_t p = (size_t)lrow * P.fullW + x;
    float4 oColor = make_float4(H.clearColor[0], H.clearColor[1], H.clearColor[2], H.clearColor[3]);
    float4 oPos = make_float4(0.0f, 0.0f, 0.0f, 1.0f), oNrm = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
    float2 oRough = make_float2(0.0f, 0.0f);
    float4 oNormRough = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // main.cpp:488-491 clear values
    float oViewZ = 0.0f;
    float origin[4];
    mat4MulVec4(P.viewInverse, 0.0f, 0.0f, 0.0f, 1.0f, origin);
    const f3 org = mk3(origin[0], origin[1], origin[2]), dir = primaryDir(P, x, y);
    RayHit hit;
    nClosest = 1;
    traverse_any<false, WIDE, TM>(sc, org, dir, 0.001f, 10000.0f, false, lds_stack, (int)threadIdx.x, HY_BLOCK, hit, tc);
    if(hit.slot >= 0)
    {
      const float4 recq = sc.tris[hit.slot * VKRT_TRI_QUADS + 2];
      const DevInstance in = sc.instances[(uint32_t)__float_as_int(recq.z)];
      const uint4 ts = sc.triShade[hit.slot];
      const uint32_t vi[3] = {ts.x, ts.y, ts.z};
      const float4* mq = (const float4*)&sc.materials[ts.w];
      const float4 m0 = mq[0], m1 = mq[1], m2 = mq[2], m3 = mq[3];
      GltfPBRMaterial mat;
      mat.pbrBaseColorFactor[0] = m0.x; mat.pbrBaseColorFactor[1] = m0.y; mat.pbrBaseColorFactor[2] = m0.z; mat.pbrBaseColorFactor[3] = m0.w;
      mat.pbrBaseColorTexture = __float_as_int(m1.x); mat.metallicFactor = m1.y; mat.roughnessFactor = m1.z;
      mat.metallicRoughnessTexture = __float_as_int(m1.w);
      mat.normalTexture = __float_as_int(m2.x); mat.emissiveFactor[0] = m2.y; mat.emissiveFactor[1] = m2.z; mat.emissiveFactor[2] = m2.w;
      mat.emissiveTexture = __float_as_int(m3.x);
      const float bw[3] = {1.0f - hit.u - hit.v, hit.u, hit.v};
      f3 wPos = mk3(0.0f), wNrm = mk3(0.0f), wTag = mk3(0.0f), wBin = mk3(0.0f);
      float tu = 0.0f, tv = 0.0f;
      f3 cornerPos[3];
      float cornerU[3], cornerV[3];
#pragma unroll
      for(int k = 0; k < 3; k++)  // vert_shader.vert:60-74 per vertex, then the rasteriser's barycentric interpolation
      {
        const float4 a = sc.vertexPN[VKRT_VERTEX_QUADS * vi[k]], b = sc.vertexPN[VKRT_VERTEX_QUADS * vi[k] + 1], tq = sc.vertexPN[VKRT_VERTEX_QUADS * vi[k] + 2];
        const f3 pw = xformPoint(in, mk3(a.x, a.y, a.z));
        const f3 n = normalize3(xformNormal(in, mk3(a.w, b.x, b.y)));
        f3 t = normalize3(xformNormal(in, mk3(tq.x, tq.y, tq.z)));
        t = normalize3(t - dot3(t, n) * n);
        const f3 bn = cross3(n, t) * tq.w;
        wPos = wPos + pw * bw[k]; wNrm = wNrm + n * bw[k]; wTag = wTag + t * bw[k]; wBin = wBin + bn * bw[k];
        tu = tu + b.z * bw[k];
        tv = tv + b.w * bw[k];
        cornerPos[k] = pw; cornerU[k] = b.z; cornerV[k] = b.w;
      }
      // dFdx / dFdy of fragTexCoord: differences inside the pixel's 2x2 quad (the neighbour is the other pixel of the quad in
      // that direction, evaluated on this primitive's plane like a helper invocation), Vulkan spec "Derivative Operations"
      TexGrad grad;
      grad.on = sc.gbufferMips != 0u;
      grad.dudx = grad.dvdx = grad.dudy = grad.dvdy = 0.0f;
      if(grad.on)
      {
        float nu, nv;
        if(planeTexCoord(org, primaryDir(P, x ^ 1u, y), cornerPos[0], cornerPos[1], cornerPos[2], cornerU, cornerV, nu, nv))
        {
          const float sgn = (x & 1u) ? -1.0f : 1.0f;
          grad.dudx = (nu - tu) * sgn; grad.dvdx = (nv - tv) * sgn;
        }
        if(planeTexCoord(org, primaryDir(P, x, y ^ 1u), cornerPos[0], cornerPos[1], cornerPos[2], cornerU, cornerV, nu, nv))
        {
          const float sgn = (y & 1u) ? -1.0f : 1.0f;
          grad.dudy = (nu - tu) * sgn; grad.dvdy = (nv - tv) * sgn;
        }
      }
      const f3 viewDir = wPos - org;
      f3 N = normalize3(wNrm);  // frag_shader.frag:96-119
      if(mat.normalTexture > -1)
      {
        f3 T = normalize3(wTag), B = normalize3(wBin);
        T = normalize3(T - dot3(T, N) * N);
        B = normalize3(B - dot3(B, N) * N - dot3(B, T) * T);
        const f4 tx = sampleTex(sc, mat.normalTexture, tu, tv, grad, st);
        f3 nrm = mk3(tx.x, tx.y, tx.z) * 2.0f - mk3(1.0f);
        nrm = normalize3(nrm);
        nrm = normalize3(T * nrm.x + B * nrm.y + N * nrm.z);
        N = nrm;
      }
      const f3 baseColor = pbrGetBaseColor(sc, mat, tu, tv, grad, st);
      float metalness, roughness;
      pbrGetMetallicRoughness(sc, mat, tu, tv, grad, metalness, roughness, st);
      const f3 albedo = (1.0f - metalness) * baseColor;
      const f3 V = normalize3(-viewDir);
      f3 color = mk3(0.0f);
      f3 emittance = mk3(mat.emissiveFactor[0], mat.emissiveFactor[1], mat.emissiveFactor[2]);
      if(mat.emissiveTexture > -1)
      {
        const f4 tx = sampleTex(sc, mat.emissiveTexture, tu, tv, grad, st);
        emittance = emittance * mk3(tx.x, tx.y, tx.z);
      }
      const unsigned retap = (mat.pbrBaseColorTexture > -1 ? 1u : 0u) + (mat.metallicRoughnessTexture > -1 ? 1u : 0u);
      for(int i = 0; i < H.lightsCount; i++)  // frag_shader.frag:193-213
      {
        const float4* lq = (const float4*)&sc.lights[i];
        const float4 l0 = lq[0], l1 = lq[1];
        const f3 lp = mk3(l0.x, l0.y, l0.z);
        f3 L = normalize3(lp - wPos);
        f3 lightIntensity = mk3(l0.w, l1.x, l1.y) * l1.z;
        if(__float_as_int(l1.w) == 0)
        {
          const f3 lDir = lp - wPos;
          const float d = length3(lDir);
          lightIntensity = lightIntensity / (d * d);
        }
        else
          L = normalize3(lp);
        const f3 Hh = normalize3(L + V);
        const float cosTheta = glsl_max(dot3(L, N), 0.0f);
        if(cosTheta > 0.0f)
        {
          st.taps += retap;
          color = color + computePBR_BRDF(N, V, L, Hh, baseColor, metalness, roughness) * lightIntensity * cosTheta;
        }
      }
      const f3 oc = emittance + color;
      oColor = make_float4(oc.x, oc.y, oc.z, albedo.x);
      oPos = make_float4(wPos.x, wPos.y, wPos.z, albedo.y);
      oNrm = make_float4(N.x, N.y, N.z, albedo.z);
      oRough = make_float2(quantizeHalf(roughness), quantizeHalf(metalness));
      if(H.nrdNormRough)
      {
        oNormRough = nrdPackNormalRoughness(N, roughness, (float)ts.w);  // (materialId -1 and 0 pack alike: clamp(id / 3, 0, 1))
        float vz[4];
        mat4MulVec4(H.viewMatrix, wPos.x, wPos.y, wPos.z, 1.0f, vz);
        oViewZ = quantizeHalf(vz[2]);
      }
    }
    H.color[p] = oColor; H.position[p] = oPos; H.normal[p] = oNrm; H.rough[p] = oRough;
    if(H.nrdNormRough)
    {
      H.nrdNormRough[p] = oNormRough;
      H.nrdViewZ[p] = oViewZ;
      H.nrdRadHitD[p] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // frag_shader.frag:136 / clear value
    }
  }
  __shared__ unsigned long long red[VKRT_COUNTER_STRIDE * (HY_BLOCK / 64)];
  const unsigned vals[5] = {nClosest, 0, 0, 0, st.taps};
  blockAddCounters(&P.counters->v[blockIdx.x % VKRT_COUNTER_SLOTS][0], vals, 5, red);
}

// One ray of every lane that has one.  SHARE: the 64 lanes of the (one-wave) workgroup walk together and lanes without a ray,
// or done with theirs, take over pending subtrees of the others (traverse_share.h) -- the call must then be reached by all 64 lanes
// (workgroup-uniform control flow around it).  Otherwise every lane walks alone.
template <bool WIDE, bool SHARE, int TM>
VKRT_DEV void hyTrace(const DevScene& sc, bool valid, f3 o, f3 d, float tmin, float tmax, bool anyHit, int* lds, int* shareLds, RayHit& hit, TravCount& tc,
                      uint32_t raySeed)
{
  if(SHARE)
  {
    uint2* stk = ((uint2*)lds) + threadIdx.x;
    if(!valid)
    {
      o = mk3(0.0f); d = mk3(1.0f, 0.0f, 0.0f); tmax = 0.0f;
    }
    if(anyHit)
      traverse_wide8_share<false, true, TM>(sc, valid, o, d, tmin, tmax, stk, shareRes(shareLds), hit, tc, raySeed);
    else
      traverse_wide8_share<false, false, TM>(sc, valid, o, d, tmin, tmax, stk, shareRes(shareLds), hit, tc, raySeed);
    if(!valid)
      hit.slot = -1;
  }
  else
  {
    hit.slot = -1;
    if(valid)
      traverse_any<false, WIDE, TM>(sc, o, d, tmin, tmax, anyHit, lds, (int)threadIdx.x, (int)blockDim.x, hit, tc, raySeed);
  }
}

// SHARE (the default with the wide layout): one wave per workgroup = one 8x8 tile, every loop and branch around a trace is
// taken by the whole wave as long as any of its pixels needs it.  The per-pixel sequence of random numbers, rays and float
// operations is that of rgen either way; tests/test_hybrid.py compares the two instantiations with each other and the oracle.
template <bool WIDE, bool SHARE, int TM>
__global__ __launch_bounds__(SHARE ? 64 : HY_BLOCK) void k_hybrid(const HybridParams H)
{
  extern __shared__ int lds_stack[];
  __shared__ int shareLds[SHARE ? VKRT_SHARE_LDS_WORDS : 1];
  const TraceParams& P = H.T;
  const DevScene& sc = P.sc;
  uint32_t x = 0, y = 0, lrow = 0;
  unsigned nClosest = 0, nShadow = 0, nPixels = 0;
  TravCount tc;
  __shared__ float lut[512];
  ShadeStats st;
  st.hits = 0; st.diffuse = 0; st.taps = 0;
  st.lut = ldsTexelLut(P.sc, lut);
  const bool inImage = pixelOf(P, x, y, lrow);
  auto anyLane = [](bool c) { return SHARE ? __any(c) != 0 : c; };
  Payload prd;
  prd.seed = 0u;
  prd.isSpecular = false; prd.lightDist = 0.0f; prd.shadowRayDir = mk3(0.0f); prd.depth = 0;
  prd.hitValue = mk3(0.0f); prd.weight = mk3(0.0f); prd.rayOrigin = mk3(0.0f); prd.rayDirection = mk3(0.0f);
  float4 color = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
  size_t p = 0;
  f3 worldPos = mk3(0.0f), worldNrm = mk3(0.0f), albedo = mk3(0.0f);
  float roughness = 0.0f, metalness = 0.0f;
  bool shaded = false;
  if(inImage)
  {
    nPixels = 1;
    p = (size_t)lrow * P.fullW + x;
    prd.seed = tea((P.flags & 1u) ? (y * P.fullW + x) : (y * x + x), P.seed);  // rgen:55
    const float4 pixelImg = H.color[p], pixelPos = H.position[p], pixelNorm = H.normal[p];
    const float2 rm = H.rough[p];
    worldPos = mk3(pixelPos.x, pixelPos.y, pixelPos.z); worldNrm = mk3(pixelNorm.x, pixelNorm.y, pixelNorm.z);
    shaded = !(worldPos.x == 0.0f && worldPos.y == 0.0f && worldPos.z == 0.0f && worldNrm.x == 0.0f && worldNrm.y == 0.0f &&
               worldNrm.z == 0.0f);  // rgen:67
    albedo = mk3(pixelImg.w, pixelPos.w, pixelNorm.w);
    roughness = rm.x; metalness = rm.y;
  }
  if(anyLane(shaded))
  {
    RayHit hit;
    if(P.pc.useShadows == 1)  // rgen:81-131
    {
      float visibility = 1.0f, lightDistance = 0.0f;
      f3 L = mk3(1.0f, 0.0f, 0.0f);
      bool want = false;
      if(shaded)
      {
        const int random_index = (int)(rnd(prd.seed) * (float)P.pc.lightsCount);
        const float4 l0 = ((const float4*)&sc.lights[random_index])[0];
        const f3 lightDir = mk3(l0.x, l0.y, l0.z) - worldPos;
        lightDistance = length3(lightDir);
        L = normalize3(lightDir);
        if(dot3(L, worldNrm) < 0.0f)
          visibility = 0.0f;
        else
          want = true;
      }
      if(anyLane(want))
      {
        hyTrace<WIDE, SHARE, TM>(sc, want, worldPos, L, 0.1f, lightDistance - 0.1f, true, lds_stack, shareLds, hit, tc, prd.seed);
        if(want)
        {
          nShadow++;
          if(hit.slot >= 0)
            visibility = 0.0f;
        }
      }
      if(shaded)
      {
        visibility = glsl_max(visibility, 0.01f);
        color.w *= visibility;
      }
    }
    if(P.pc.useAO == 1)  // rgen:134-169
    {
      float ao = 0.0f;
      f3 tangent = mk3(0.0f), binormal = mk3(0.0f);
      if(shaded)
        createCoordinateSystem(worldNrm, tangent, binormal);
      const float weightAo = 1.0f / 4;
      for(int i = 0; i < 4; i++)
      {
        f3 rayDir = mk3(1.0f, 0.0f, 0.0f);
        if(shaded)
          rayDir = normalize3(samplingHemisphere(prd.seed, tangent, binormal, worldNrm));
        hyTrace<WIDE, SHARE, TM>(sc, shaded, worldPos, rayDir, 0.1f, 2.0f, true, lds_stack, shareLds, hit, tc, prd.seed);
        if(shaded)
        {
          nShadow++;
          if(hit.slot >= 0)
            ao += weightAo;
        }
      }
      if(shaded)
        color.w *= (1.0f - ao);
    }
    if(P.pc.useGI == 1 && !H.giLater)  // rgen:172-282
    {
      f3 curWeight = mk3(0.0f), hitValue = mk3(0.0f);
      float hitDists = 0.0f;
      if(shaded)
      {
        f3 direction;
        const float ratio = metalness * (1.0f - roughness);
        if(ratio < 0.8f)
        {
          prd.isSpecular = false;
          f3 tangent, binormal;
          createCoordinateSystem(worldNrm, tangent, binormal);
          direction = normalize3(samplingHemisphere(prd.seed, tangent, binormal, worldNrm));
          curWeight = albedo;
        }
        else
        {
          prd.isSpecular = true;
          float cam[4];
          mat4MulVec4(P.viewInverse, 0.0f, 0.0f, 0.0f, 1.0f, cam);
          const f3 V = normalize3(mk3(cam[0], cam[1], cam[2]) - worldPos);
          direction = normalize3(glsl_reflect(-V, worldNrm));
          curWeight = mk3(1.0f);
        }
        prd.hitValue = mk3(0.0f);
        prd.rayOrigin = worldPos;
        prd.rayDirection = direction;
        prd.depth = 1;
        prd.weight = mk3(0.0f);
      }
      bool active = shaded && prd.depth < (uint32_t)P.pc.depth;
      while(anyLane(active))
      {
        const f3 rd = prd.rayDirection;
        hyTrace<WIDE, SHARE, TM>(sc, active, prd.rayOrigin, rd, 0.001f, 10000.0f, false, lds_stack, shareLds, hit, tc, prd.seed);
        bool needShadow = false;
        if(active)
        {
          nClosest++;
          if(hit.slot >= 0)
            closestHitShader(sc, P.pc, hit, rd, prd, st);
          else
            missShader(P.pc, prd);
          needShadow = !prd.isSpecular && prd.depth != 100u;
        }
        bool shadowHit = false;
        if(anyLane(needShadow))
        {
          hyTrace<WIDE, SHARE, TM>(sc, needShadow, prd.rayOrigin, prd.shadowRayDir, 0.001f, prd.lightDist - 0.1f, true, lds_stack, shareLds, hit, tc, prd.seed);
          if(needShadow)
          {
            nShadow++;
            shadowHit = hit.slot >= 0;
          }
        }
        if(active)
        {
          if(!shadowHit)
          {
            const f3 q = prd.hitValue * curWeight;
            hitValue = hitValue + mk3(glsl_min(q.x, 10.0f), glsl_min(q.y, 10.0f), glsl_min(q.z, 10.0f));
          }
          if(prd.depth == 1u && !prd.isSpecular)  // rgen:253-264
            hitDists = shadowHit ? 0.5f * prd.lightDist : prd.lightDist;
          curWeight = curWeight * prd.weight;
          prd.depth++;
          active = prd.depth < (uint32_t)P.pc.depth;
        }
      }
      if(shaded)
      {
        color.x = hitValue.x; color.y = hitValue.y; color.z = hitValue.z;
        if(H.nrdRadHitD)
        {  // rgen:273-281: REBLUR front end, hitDistParams (3, 1, 20, -25), rgba16f store
          const float viewZ = H.nrdViewZ[p];
          const float t = glsl_clamp(exp2f(-25.0f * roughness * roughness), 0.0f, 1.0f);
          const float f = (3.0f + fabsf(viewZ) * 1.0f) * (1.0f * (1.0f - t) + 20.0f * t);
          float normHitDist = glsl_clamp(hitDists / f, 0.0f, 1.0f);
          f3 rad = hitValue;
          const bool bad = isnan(rad.x) || isnan(rad.y) || isnan(rad.z) || isinf(rad.x) || isinf(rad.y) || isinf(rad.z);
          rad = bad ? mk3(0.0f) : mk3(glsl_clamp(rad.x, 0.0f, 65504.0f), glsl_clamp(rad.y, 0.0f, 65504.0f), glsl_clamp(rad.z, 0.0f, 65504.0f));
          normHitDist = (isnan(normHitDist) || isinf(normHitDist)) ? 0.0f : glsl_clamp(normHitDist, 0.0f, 1.0f);
          if(normHitDist != 0.0f)
            normHitDist = glsl_max(normHitDist, 1e-7f);
          const float Y = (rad.x * 0.25f + rad.y * 0.5f) + rad.z * 0.25f;
          const float Co = (rad.x * 0.5f + rad.y * 0.0f) + rad.z * -0.5f;
          const float Cg = (rad.x * -0.25f + rad.y * 0.5f) + rad.z * -0.25f;
          H.nrdRadHitD[p] = make_float4(quantizeHalf(Y), quantizeHalf(Co), quantizeHalf(Cg), quantizeHalf(normHitDist));
        }
      }
    }
  }
  if(inImage && H.giLater && P.pc.useGI == 1)
    H.giLater[p] = make_uint2(prd.seed, __float_as_uint(color.w));  // the GI kernels continue from here and write the pixel
  else if(inImage)
  {
    // accumulateFrames, rgen:36-48 (all four channels)
    if(P.pc.frame > 0)
    {
      const float a = 1.0f / (float)(P.pc.frame + 1);
      const float4 old = H.accum[p];
      H.accum[p] = make_float4(old.x * (1.0f - a) + color.x * a, old.y * (1.0f - a) + color.y * a, old.z * (1.0f - a) + color.z * a,
                               old.w * (1.0f - a) + color.w * a);
    }
    else
      H.accum[p] = color;
  }
  __shared__ unsigned long long red[VKRT_COUNTER_STRIDE * ((SHARE ? 64 : HY_BLOCK) / 64)];
  const unsigned vals[6] = {nClosest, nShadow, st.hits, st.diffuse, st.taps, nPixels};
  blockAddCounters(&P.counters->v[blockIdx.x % VKRT_COUNTER_SLOTS][0], vals, 6, red);
}

// post.frag:36-58
__global__ void k_post(int rtMode, int viewAccumulated, int useGI, unsigned n, const float4* mainImg, const float4* rtImg, float4* out)
{
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  float4 m = mainImg[i];
  if(rtMode == 0)
  {
    const float4 r = rtImg[i];
    if(viewAccumulated == 0)
      m = make_float4(m.x * r.w + r.x, m.y * r.w + r.y, m.z * r.w + r.z, 1.0f);
    else if(useGI == 1)
    {
      m.x = r.x * r.w; m.y = r.y * r.w; m.z = r.z * r.w;
    }
    else
    {
      m.x = r.w; m.y = r.w; m.z = r.w;
    }
  }
  const float gamma = 1.0f / 2.2f;
  out[i] = make_float4(powf(m.x, gamma), powf(m.y, gamma), powf(m.z, gamma), powf(m.w, gamma));
}

hipError_t vkrt_launch_gbuffer(const TraceParams& P, const float clearColor[4], int lightsCount, float* color, float* position, float* normal,
                               float* rough, const NrdPlanes* nrd, hipStream_t stream)
{
  HybridParams H;
  H.T = P;
  H.giLater = nullptr;
  H.nrdNormRough = nrd ? (float4*)nrd->normRough : nullptr;
  H.nrdViewZ = nrd ? nrd->viewZ : nullptr;
  H.nrdRadHitD = nrd ? (float4*)nrd->radHitD : nullptr;
  for(int k = 0; k < 16; k++) H.viewMatrix[k] = nrd ? nrd->viewMatrix[k] : 0.0f;
  H.color = (float4*)color; H.position = (float4*)position; H.normal = (float4*)normal; H.rough = (float2*)rough; H.accum = nullptr;
  for(int k = 0; k < 4; k++) H.clearColor[k] = clearColor[k];
  H.lightsCount = lightsCount;
  const unsigned blocks = (P.tileCount * 64u + HY_BLOCK - 1) / HY_BLOCK;
  const size_t lds = (size_t)P.sc.stackCap * HY_BLOCK * sizeof(int);
  const int tm = (P.sc.watertight ? VKRT_TM_WATERTIGHT : 0) | (P.sc.dissolve ? VKRT_TM_MASKID : 0);
#define VKRT_GB_LAUNCH(W, TM) hipLaunchKernelGGL((k_gbuffer<W, TM>), dim3(blocks), dim3(HY_BLOCK), lds, stream, H)
  if(P.sc.layout == 1u)
  {
    switch(tm)
    {
      case 0: VKRT_GB_LAUNCH(true, 0); break;
      case VKRT_TM_WATERTIGHT: VKRT_GB_LAUNCH(true, VKRT_TM_WATERTIGHT); break;
      case VKRT_TM_MASKID: VKRT_GB_LAUNCH(true, VKRT_TM_MASKID); break;
      default: VKRT_GB_LAUNCH(true, VKRT_TM_WATERTIGHT | VKRT_TM_MASKID); break;
    }
  }
  else
  {
    switch(tm)
    {
      case 0: VKRT_GB_LAUNCH(false, 0); break;
      case VKRT_TM_WATERTIGHT: VKRT_GB_LAUNCH(false, VKRT_TM_WATERTIGHT); break;
      case VKRT_TM_MASKID: VKRT_GB_LAUNCH(false, VKRT_TM_MASKID); break;
      default: VKRT_GB_LAUNCH(false, VKRT_TM_WATERTIGHT | VKRT_TM_MASKID); break;
    }
  }
#undef VKRT_GB_LAUNCH
  return hipGetLastError();
}

hipError_t vkrt_launch_hybrid(const TraceParams& P, const float* color, const float* position, const float* normal, const float* rough, float* accum,
                              const NrdPlanes* nrd, uint2* giLater, hipStream_t stream)
{
  HybridParams H;
  H.T = P;
  H.giLater = giLater;
  H.nrdNormRough = nullptr;
  H.nrdViewZ = nrd ? nrd->viewZ : nullptr;
  H.nrdRadHitD = nrd ? (float4*)nrd->radHitD : nullptr;
  for(int k = 0; k < 16; k++) H.viewMatrix[k] = 0.0f;
  H.color = (float4*)color; H.position = (float4*)position; H.normal = (float4*)normal; H.rough = (float2*)rough; H.accum = (float4*)accum;
  for(int k = 0; k < 4; k++) H.clearColor[k] = 0.0f;
  H.lightsCount = P.pc.lightsCount;
  // with the wide layout and work sharing enabled (the defaults) one wave per workgroup walks its 64 pixels' rays together
  const bool share = P.sc.layout == 1u && P.sc.shareMinIdle != 0u && P.sc.triThreshold != 0u;
  const unsigned block = share ? 64u : (unsigned)HY_BLOCK;
  const unsigned blocks = (P.tileCount * 64u + block - 1) / block;
  const size_t lds = (size_t)P.sc.stackCap * block * sizeof(int);
  const int tm = (P.sc.watertight ? VKRT_TM_WATERTIGHT : 0) | (P.sc.dissolve ? VKRT_TM_DISSOLVE : 0);
#define VKRT_HY_LAUNCH(W, S, TM) hipLaunchKernelGGL((k_hybrid<W, S, TM>), dim3(blocks), dim3(block), lds, stream, H)
#define VKRT_HY_MODES(W, S)                                                                                                            \
  switch(tm)                                                                                                                            \
  {                                                                                                                                     \
    case 0: VKRT_HY_LAUNCH(W, S, 0); break;                                                                                             \
    case 1: VKRT_HY_LAUNCH(W, S, 1); break;                                                                                             \
    case 2: VKRT_HY_LAUNCH(W, S, 2); break;                                                                                             \
    default: VKRT_HY_LAUNCH(W, S, 3); break;                                                                                            \
  }
  if(share)
  {
    VKRT_HY_MODES(true, true)
  }
  else if(P.sc.layout == 1u)
  {
    VKRT_HY_MODES(true, false)
  }
  else
  {
    VKRT_HY_MODES(false, false)
  }
#undef VKRT_HY_MODES
#undef VKRT_HY_LAUNCH
  return hipGetLastError();
}

hipError_t vkrt_launch_post(int rtMode, int viewAccumulated, int useGI, unsigned n, const float* mainImg, const float* rtImg, float* out,
                            hipStream_t stream)
{
  hipLaunchKernelGGL(k_post, dim3((n + 255) / 256), dim3(256), 0, stream, rtMode, viewAccumulated, useGI, n, (const float4*)mainImg,
                     (const float4*)rtImg, (float4*)out);
  return hipGetLastError();
}
