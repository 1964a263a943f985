// rgen.h -- the ray-generation program (reference shaders/raytrace.rgen:24-146) as a per-path state
// machine shared by the megakernel (pathtrace.hip) and the wavefront kernels (wavefront.hip).
// One "path" = one pixel; its unit of work is one ray (closest-hit or shadow).
#pragma once
#include "device_math.h"
#include "device_scene.h"
#include "shade.h"

struct LaneState
{
  Payload prd;
  f3 curWeight, hitValue, hitValues;
  f3 camOrigin;
  uint32_t px;       // global pixel column (gl_LaunchIDEXT.x)
  uint32_t lrow;     // row in the shard-local buffer; gl_LaunchIDEXT.y = globalRow(P, lrow)
  int smpl;
  int stage;         // 0: next ray is the closest-hit ray, 1: next ray is the shadow ray
};

// shard-local row -> global row (include/vkrt.h vkrt_shard)
VKRT_DEV uint32_t globalRow(const TraceParams& P, uint32_t lrow)
{
  if(P.stripRows == 0u)
    return lrow;
  const uint32_t s = lrow / P.stripRows, r = lrow % P.stripRows;
  return (s * P.shardCount + P.shardIndex) * P.stripRows + r;
}

// raytrace.rgen:42-60 -- start sample `smpl` of the lane's pixel
VKRT_DEV void startSample(const TraceParams& P, LaneState& L)
{
  const float r1 = rnd(L.prd.seed);
  const float r2 = rnd(L.prd.seed);
  const float jx = P.pc.frame == 0 ? 0.5f : r1, jy = P.pc.frame == 0 ? 0.5f : r2;
  const float pcx = (float)L.px + jx, pcy = (float)globalRow(P, L.lrow) + jy;  // (the global row is only needed here: two integer
                                                                               //  divisions per sample when sharded, not per ray)
  const float inU = pcx / (float)P.fullW, inV = pcy / (float)P.fullH;
  const float dx = inU * 2.0f - 1.0f, dy = inV * 2.0f - 1.0f;
  float target[4], direction[4];
  mat4MulVec4(P.projInverse, dx, dy, 1.0f, 1.0f, target);
  const f3 tn = normalize3(mk3(target[0], target[1], target[2]));
  mat4MulVec4(P.viewInverse, tn.x, tn.y, tn.z, 0.0f, direction);
  L.prd.hitValue = mk3(0.0f);
  L.prd.rayOrigin = L.camOrigin;
  L.prd.rayDirection = mk3(direction[0], direction[1], direction[2]);
  L.prd.depth = 0;
  L.prd.weight = mk3(0.0f);
  L.curWeight = mk3(1.0f);
  L.hitValue = mk3(0.0f);
  L.stage = 0;
}

// raytrace.rgen:27-30 -- bind a pixel to the lane
VKRT_DEV void startPixel(const TraceParams& P, LaneState& L, uint32_t x, uint32_t y, uint32_t lrow)
{
  L.px = x; L.lrow = lrow;  // (y == globalRow(P, lrow))
  const uint32_t index = (P.flags & 1u) ? (y * P.fullW + x) : (y * x + x);
  L.prd.seed = tea(index, P.seed);
  L.prd.isSpecular = false;
  L.prd.lightDist = 0.0f;
  L.prd.shadowRayDir = mk3(0.0f);
  float origin[4];
  mat4MulVec4(P.viewInverse, 0.0f, 0.0f, 0.0f, 1.0f, origin);
  L.camOrigin = mk3(origin[0], origin[1], origin[2]);
  L.hitValues = mk3(0.0f);
  L.smpl = 0;
  startSample(P, L);
}

// raytrace.rgen:120,136-145 -- resolve and store the pixel
VKRT_DEV void storePixel(const TraceParams& P, const LaneState& L)
{
  const f3 res = L.hitValues / (float)P.pc.samples;
  float4* dst = (float4*)P.image + ((size_t)L.lrow * P.fullW + L.px);
  if(P.pc.frame > 0 && !(P.flags & VKRT_FLAG_STORE_STAGED))  // (a frame in flight leaves `res` in its staging plane; blendPixel follows in frame order)
  {
    const float a = 1.0f / (float)(P.pc.frame + 1);
    const float4 old = *dst;
    const f3 m = glsl_mix(mk3(old.x, old.y, old.z), res, a);
    *dst = make_float4(m.x, m.y, m.z, 1.0f);
  }
  else
    *dst = make_float4(res.x, res.y, res.z, 1.0f);
}

// raytrace.rgen:136-145 for a pixel whose value `res` was staged (frames in flight): same operations as storePixel's
VKRT_DEV void blendPixel(float4* dst, float4 staged, int frame)
{
  if(frame > 0)
  {
    const float a = 1.0f / (float)(frame + 1);
    const float4 old = *dst;
    const f3 m = glsl_mix(mk3(old.x, old.y, old.z), mk3(staged.x, staged.y, staged.z), a);
    *dst = make_float4(m.x, m.y, m.z, 1.0f);
  }
  else
    *dst = make_float4(staged.x, staged.y, staged.z, 1.0f);
}

// After the closest-hit ray of the current segment: run rchit / rmiss (raytrace.rgen:64-75).
// Returns true when a shadow ray has to be traced before the segment can be accumulated (rgen:79).
// `ts`: the hit triangle's shading record (sc.triShade[hit.slot]); ignored on a miss.
VKRT_DEV bool afterClosestRay(const TraceParams& P, LaneState& L, const RayHit& hit, const uint4 ts, f3 rayDir, ShadeStats& st)
{
  if(hit.slot >= 0)
    closestHitShader(P.sc, P.pc, hit, ts, rayDir, L.prd, st);
  else
    missShader(P.pc, L.prd);
  if(!L.prd.isSpecular && L.prd.depth != 100u)
  {
    L.stage = 1;
    return true;
  }
  return false;
}

// raytrace.rgen:99-102,115 -- the two products of a finished segment: its clamped radiance contribution and the path
// weight after it.  Split from the bookkeeping below so the wavefront pipeline can carry (contrib, nextWeight) across
// the shadow ray instead of (prd.hitValue, curWeight, prd.weight); the float operations are the same.
VKRT_DEV void segmentTerms(const LaneState& L, f3& contrib, f3& nextWeight)
{
  const f3 q = L.prd.hitValue * L.curWeight;
  contrib = mk3(glsl_min(q.x, 10.0f), glsl_min(q.y, 10.0f), glsl_min(q.z, 10.0f));
  nextWeight = L.curWeight * L.prd.weight;
}

// raytrace.rgen:99-120 -- accumulate the segment, advance depth / sample.  Returns false when the
// pixel is complete (its value has been stored).
VKRT_DEV bool advanceSegment(const TraceParams& P, LaneState& L, bool shadowHit, f3 contrib, f3 nextWeight)
{
  L.stage = 0;
  if(!shadowHit)  // rgen:99-102
    L.hitValue = L.hitValue + contrib;
  L.curWeight = nextWeight;  // rgen:115
  L.prd.depth++;
  if(!(L.prd.depth < (uint32_t)P.pc.depth))
  {
    L.hitValues = L.hitValues + L.hitValue;
    L.smpl++;
    if(L.smpl < P.pc.samples)
      startSample(P, L);
    else
    {
      storePixel(P, L);
      return false;
    }
  }
  return true;
}

VKRT_DEV bool accumulateAndAdvance(const TraceParams& P, LaneState& L, bool shadowHit)
{
  f3 contrib, nextWeight;
  segmentTerms(L, contrib, nextWeight);
  return advanceSegment(P, L, shadowHit, contrib, nextWeight);
}
