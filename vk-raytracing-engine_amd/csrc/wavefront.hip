// wavefront.hip -- the path-trace dispatch as a wavefront pipeline (default execution mode).
// Replaces vkCmdTraceRaysKHR(W,H,1) over raytrace.rgen/.rchit/.rmiss/raytraceShadow.rmiss
// (reference hello_vulkan.cpp:1446) exactly like pathtrace.hip, with the same per-path state machine
// (rgen.h) and therefore the same results; the work is re-scheduled for gfx950:
//
//   k_wf_init          one thread per pixel: rgen prologue (seed, camera ray of sample 0) -> closest-ray stream.
//   k_wf_traverse      one thread per queued ray, batch-synchronous: the 64 rays of a wave start together, so the
//                      top tree levels are fetched as coalesced/broadcast loads; lanes that finish early take over
//                      pending subtrees of their neighbours (traverse_share.h).  Workgroups are one wave and
//                      homogeneous in ray kind: closest-hit rays first (rgen:64-75), then shadow rays (rgen:85-97, any-hit).
//                      Per-lane stacks in LDS.  Software stand-in for traceRayEXT.
//   k_wf_shade         one launch for the three result streams: rchit / rmiss, the segment accumulation (rgen:99-120),
//                      the next ray(s) of the path.
//
// Path records MOVE with their queue position: a round reads the records of its input streams front to back
// and every surviving path writes its new record at the position a block-aggregated ballot compaction assigns
// it in the next round's stream (ping-pong buffers).  All record traffic is therefore sequential; the state is
// kept as structure-of-arrays planes of float4, so the 64 lanes of a wave read 1 KB contiguous per plane.
// (The first version kept records in place and queued path ids: every record access was a random 16-byte gather
// and the shade kernels ran at memory-system throughput, r01_experiments.md #24.)
//
// Three streams (round 2).  The shadow ray of a diffuse segment and the closest-hit ray of the NEXT segment start at the same
// point and neither needs the other's result: only the accumulation `hitValue += contrib` (rgen:99-102) waits for the shadow
// ray.  A diffuse hit that is not the last segment of its sample therefore emits ONE "pair" record carrying both rays; they
// are traced in the same round and the next shade step first finishes segment k (accumulate, weight, depth++), then shades
// segment k+1's hit.  The rays traced, their order per path and every float operation are those of rgen's loop; only the
// round in which a ray is traced changes.  A frame needs samples * (depth + 1) rounds instead of 2 * samples * depth
// (144 instead of 256 on the bench workload), each with ~1.8x the rays, and a diffuse segment moves ~35 % fewer record bytes
// (one record instead of a shadow record followed by a closest record).
//
// A frame = init + samples * (depth + 1) x (traverse, shade) enqueued back to back on the caller's
// stream; stream counts stay on the device (no host synchronisation inside a frame).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "device_math.h"
#include "device_scene.h"
#include "kernels.h"
#include "rgen.h"
#include "shade.h"
#include "traverse.h"
#include "traverse_wide.h"
#include "traverse_share.h"

#define WF_BLOCK 256
#define VKRT_FLAG_COUNT_WORK 2u  // = VKRT_TRACE_COUNT_TRAVERSAL (include/vkrt.h)

#include "wf_streams.h"

VKRT_DEV unsigned packFlags(const LaneState& L)
{
  return (L.prd.depth & 0xffu) | (((unsigned)L.smpl & 0xffffu) << 8) | ((L.prd.isSpecular ? 1u : 0u) << 25);
}

// state common to all streams (S0..S2) -> lane
VKRT_DEV void loadCommon(const TraceParams& P, const WfBuffers& B, int parity, int type, unsigned i, LaneState& L)
{
  const float4 s0 = wfLoad(rec(B, parity, type, WF_S0, i)), s1 = wfLoad(rec(B, parity, type, WF_S1, i)), s2 = wfLoad(rec(B, parity, type, WF_S2, i));
  const unsigned flags = __float_as_uint(s1.w), pix = __float_as_uint(s2.w);
  L.curWeight = mk3(s0.x, s0.y, s0.z); L.prd.seed = __float_as_uint(s0.w);
  L.hitValue = mk3(s1.x, s1.y, s1.z);
  L.prd.depth = flags & 0xffu; L.smpl = (int)((flags >> 8) & 0xffffu);
  L.prd.isSpecular = ((flags >> 25) & 1u) != 0u;
  L.hitValues = mk3(s2.x, s2.y, s2.z);
  L.px = pix & 0xffffu; L.lrow = pix >> 16;
  float origin[4];
  mat4MulVec4(P.viewInverse, 0.0f, 0.0f, 0.0f, 1.0f, origin);  // rgen:30 (uniform; cheaper to recompute than to carry)
  L.camOrigin = mk3(origin[0], origin[1], origin[2]);
  L.prd.lightDist = 0.0f;
  L.prd.shadowRayDir = mk3(0.0f);
  L.prd.hitValue = mk3(0.0f);
  L.prd.weight = mk3(0.0f);
  L.stage = 0;
}

VKRT_DEV void storeState(const WfBuffers& B, int parity, int type, unsigned i, const LaneState& L, f3 weight)
{
  wfStore(rec(B, parity, type, WF_S0, i), make_float4(weight.x, weight.y, weight.z, __uint_as_float(L.prd.seed)));
  wfStore(rec(B, parity, type, WF_S1, i), make_float4(L.hitValue.x, L.hitValue.y, L.hitValue.z, __uint_as_float(packFlags(L))));
  wfStore(rec(B, parity, type, WF_S2, i), make_float4(L.hitValues.x, L.hitValues.y, L.hitValues.z, __uint_as_float(L.px | (L.lrow << 16))));
}

// a path whose next ray is the closest-hit ray (rgen:64-75) -> slot i of stream C
VKRT_DEV void storeClosest(const WfBuffers& B, int parity, unsigned i, const LaneState& L)
{
  wfStore(rec(B, parity, WF_C, WF_R0, i), make_float4(L.prd.rayOrigin.x, L.prd.rayOrigin.y, L.prd.rayOrigin.z, 10000.0f));
  wfStore(rec(B, parity, WF_C, WF_R1, i), make_float4(L.prd.rayDirection.x, L.prd.rayDirection.y, L.prd.rayDirection.z, 0.0f));
  storeState(B, parity, WF_C, i, L, L.curWeight);
}

// a path waiting for the shadow ray of its current segment (rgen:85-97) -> slot i of stream S (last segment of the sample)
// or, with the closest-hit ray of the next segment riding along, of stream P
VKRT_DEV void storeShadow(const WfBuffers& B, int parity, int type, unsigned i, const LaneState& L, f3 contrib, f3 nextWeight)
{
  wfStore(rec(B, parity, type, WF_R0, i), make_float4(L.prd.rayOrigin.x, L.prd.rayOrigin.y, L.prd.rayOrigin.z, L.prd.lightDist - 0.1f));
  wfStore(rec(B, parity, type, WF_R1, i), make_float4(L.prd.shadowRayDir.x, L.prd.shadowRayDir.y, L.prd.shadowRayDir.z, L.prd.lightDist));
  if(type == WF_P)
    wfStore(rec(B, parity, type, WF_R2, i), make_float4(L.prd.rayDirection.x, L.prd.rayDirection.y, L.prd.rayDirection.z, 0.0f));
  storeState(B, parity, type, i, L, nextWeight);
  wfStore(rec(B, parity, type, WF_S3, i), make_float4(contrib.x, contrib.y, contrib.z, 0.0f));
}

// Block-aggregated slot assignment in the next round's streams: ballot + popcount inside each wave, wave totals
// combined through LDS, ONE global atomic per workgroup and stream (the count is a single word: per-wave atomics
// serialise near 88/us, MI355X_MICROARCH.md "dequeue").  Must be called by every thread of the block; `to` = the stream the
// thread's path goes to (WF_C / WF_S / WF_P) or -1; returns its slot there.  The slots of a block are contiguous per stream,
// in lane order.  wsum: LDS [WF_TYPES * (nw + 1)].
VKRT_DEV unsigned claimSlots(const WfBuffers& B, int parity, int to, unsigned lane, unsigned* wsum)
{
  const unsigned wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const unsigned long long m[WF_TYPES] = {__ballot(to == WF_C), __ballot(to == WF_S), __ballot(to == WF_P)};
  if(lane == 0)
  {
#pragma unroll
    for(int t = 0; t < WF_TYPES; t++) wsum[t * (nw + 1) + wave] = (unsigned)__popcll(m[t]);
  }
  __syncthreads();
  if(threadIdx.x < WF_TYPES)
  {
    unsigned* w = wsum + threadIdx.x * (nw + 1);
    unsigned tot = 0;
    for(unsigned k = 0; k < nw; k++)
    {
      const unsigned c = w[k];
      w[k] = tot;
      tot += c;
    }
    w[nw] = tot ? atomicAdd(countOf(B, parity, (int)threadIdx.x), tot) : 0u;
  }
  __syncthreads();
  if(to < 0)
    return 0u;
  const unsigned long long below = (1ull << lane) - 1ull;
  const unsigned long long mine = to == WF_C ? m[0] : to == WF_S ? m[1] : m[2];
  return wsum[to * (nw + 1) + nw] + wsum[to * (nw + 1) + wave] + (unsigned)__popcll(mine & below);
}

// ---- init: raytrace.rgen:27-60 for every pixel of the shard -------------------------------------------------
__global__ __launch_bounds__(WF_BLOCK) void k_wf_init(const TraceParams P, const WfBuffers B)
{
  const unsigned lane = lane_id();
  const unsigned w = blockIdx.x * blockDim.x + threadIdx.x;  // tile-major work index
  bool alive = false;
  unsigned nPixels = 0;
  LaneState L;
  if(w < P.tileCount * 64u)
  {
    const unsigned tile = P.tileFirst + (w >> 6), inTile = w & 63u;
    const uint32_t x = (tile % P.tilesX) * 8u + (inTile & 7u);
    const uint32_t lrow = (tile / P.tilesX) * 8u + (inTile >> 3);
    if(x < P.fullW && lrow < P.localRows)
    {
      const uint32_t y = globalRow(P, lrow);
      if(y < P.fullH)
      {
        startPixel(P, L, x, y, lrow);
        nPixels = 1;
        if(P.pc.samples <= 0 || P.pc.depth <= 0)
          storePixel(P, L);  // degenerate launch: no rays
        else
          alive = true;
      }
    }
  }
  __shared__ unsigned wsum[WF_TYPES * (WF_BLOCK / 64 + 1)];
  const unsigned slot = claimSlots(B, 0, alive ? WF_C : -1, lane, wsum);
  if(alive)
    storeClosest(B, 0, slot, L);
  __shared__ unsigned long long red[VKRT_COUNTER_STRIDE * (WF_BLOCK / 64)];
  const unsigned vals[6] = {0, 0, 0, 0, 0, nPixels};
  blockAddCounters(&P.counters->v[blockIdx.x % VKRT_COUNTER_SLOTS][0], vals, 6, red);
}

// ---- hybrid mode: the GI path of raytraceHybrid.rgen:172-282 on the same streams ----------------------------------------------
// A pixel's GI path is one sample that starts at depth 1 from the G-buffer position; its record is the path tracer's, with the
// S2 plane (unused: there is no sum over samples) carrying hitDists (.x, rgen:253-264) and the visibility term of the direct
// part (.y, the alpha of the accumulation image).  HYBRID instantiations of the shade functions end a sample with hybridFinish
// instead of storePixel and keep hitDists; everything else -- rays, shaders, compaction -- is shared with the path tracer.
// raytraceHybrid.rgen:266-282 + 36-48: the pixel's GI radiance is complete
VKRT_DEV void hybridFinish(const TraceParams& P, const HybridGi& G, uint32_t px, uint32_t lrow, bool shaded, f3 hitValue, float hitDists, float alpha)
{
  const size_t p = (size_t)lrow * P.fullW + px;
  float4 color = make_float4(0.0f, 0.0f, 0.0f, alpha);
  if(shaded)
  {
    color.x = hitValue.x; color.y = hitValue.y; color.z = hitValue.z;
    if(G.nrdRadHitD)
    {  // REBLUR front end, hitDistParams (3, 1, 20, -25), rgba16f store (same operation order as k_hybrid)
      const float roughness = G.rough[p].x;
      const float viewZ = G.nrdViewZ[p];
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
      G.nrdRadHitD[p] = make_float4(quantizeHalf(Y), quantizeHalf(Co), quantizeHalf(Cg), quantizeHalf(normHitDist));
    }
  }
  if(P.pc.frame > 0)  // accumulateFrames, rgen:36-48 (all four channels)
  {
    const float a = 1.0f / (float)(P.pc.frame + 1);
    const float4 old = G.accum[p];
    G.accum[p] = make_float4(old.x * (1.0f - a) + color.x * a, old.y * (1.0f - a) + color.y * a, old.z * (1.0f - a) + color.z * a,
                             old.w * (1.0f - a) + color.w * a);
  }
  else
    G.accum[p] = color;
}
// rgen:240-266 for one finished segment of the GI path; false = the path (and the pixel) is complete
VKRT_DEV bool advanceSegmentHybrid(const TraceParams& P, const HybridGi& G, LaneState& L, bool shadowHit, f3 contrib, f3 nextWeight, float lightDist)
{
  L.stage = 0;
  if(!shadowHit)
    L.hitValue = L.hitValue + contrib;
  if(L.prd.depth == 1u && !L.prd.isSpecular)  // rgen:253-264 (only segments that traced a shadow ray get here with depth 1)
    L.hitValues.x = shadowHit ? 0.5f * lightDist : lightDist;
  L.curWeight = nextWeight;
  L.prd.depth++;
  if(!(L.prd.depth < (uint32_t)P.pc.depth))
  {
    hybridFinish(P, G, L.px, L.lrow, true, L.hitValue, L.hitValues.x, L.hitValues.y);
    return false;
  }
  return true;
}

// ---- shade, results of streams C and P: [finish segment k,] rchit / rmiss of the traced closest-hit ray, then the next ray(s) ----
// (Regrouping the 256 results of a workgroup by lobe through LDS between the hit shader's front half and its diffuse / specular
// tail -- closestHitFront / closestHitLobe / closestHitTail in shade.h -- so that a wave runs one branch only was built and
// measured in round 2: bit-identical images, ~30 % fewer VALU instructions, no change in kernel time (5.94 vs 5.91 ms per
// 4-spp frame): the stage is bound by its stream traffic to HBM and the latency of its gathers, not by issue.  Removed again;
// profiles/r02_experiments.md #52.)
// LDS of a shade workgroup (one allocation for the kernel: the C and the pair instantiation of shadeHitBlock share it)
struct ShadeLds
{
  float lut[512];
  unsigned wsum[WF_TYPES * (WF_BLOCK / 64 + 1)];
};

template <bool PAIR, bool HYBRID>
VKRT_DEV void shadeHitBlock(const TraceParams& P, const WfBuffers& B, const HybridGi& G, const int par, const unsigned count, const unsigned block, ShadeLds& lds)
{
  const int type = PAIR ? WF_P : WF_C;
  const unsigned lane = lane_id();
  const unsigned qi = block * WF_BLOCK + threadIdx.x;
  ShadeStats st;
  st.hits = 0; st.diffuse = 0; st.taps = 0;
  st.lut = ldsTexelLut(P.sc, lds.lut);
  unsigned* wsum = lds.wsum;
  int to = -1;
  LaneState L;
  f3 contrib = mk3(0.0f), nextWeight = mk3(0.0f);
  if(qi < count)
  {
    loadCommon(P, B, par, type, qi, L);
    const float4 h = wfLoad(rec(B, par, type, WF_H0, qi)), t4 = wfLoad(rec(B, par, type, WF_H1, qi));
    const float4 rd = wfLoad(rec(B, par, type, PAIR ? WF_R2 : WF_R1, qi));
    L.prd.rayDirection = mk3(rd.x, rd.y, rd.z);  // direction of the closest-hit ray that was traced
    L.prd.rayOrigin = mk3(0.0f);                 // rchit / rmiss do not read it
    if(PAIR)
    {
      // finish segment k first (rgen:99-116): its shadow ray came back with this record.  S0 holds the weight after it;
      // the segment is never the last of its sample (emission rule below), so advanceSegment only accumulates and steps depth
      const float4 s3 = wfLoad(rec(B, par, WF_P, WF_S3, qi));
      const bool shadowHit = __float_as_int(h.x) != 0;
      if(HYBRID)
        (void)advanceSegmentHybrid(P, G, L, shadowHit, mk3(s3.x, s3.y, s3.z), L.curWeight, L.prd.depth == 1u ? wfLoad(rec(B, par, WF_P, WF_R1, qi)).w : 0.0f);
      else
        (void)advanceSegment(P, L, shadowHit, mk3(s3.x, s3.y, s3.z), L.curWeight);
    }
    RayHit hit;
    hit.t = h.x; hit.u = h.y; hit.v = h.z; hit.slot = __float_as_int(h.w);  // instance id of the hit (>= 0) or -1; t is not used by the shaders
    const uint4 ts = make_uint4(__float_as_uint(t4.x), __float_as_uint(t4.y), __float_as_uint(t4.z), __float_as_uint(t4.w));
    if(hit.slot >= 0)
      closestHitShaderInst(P.sc, P.pc, hit, (uint32_t)hit.slot, ts, L.prd.rayDirection, L.prd, st);
    else
      missShader(P.pc, L.prd);
    segmentTerms(L, contrib, nextWeight);
    // VKRT_OPT_SKIP_DEAD_SHADOW_RAYS (launch-uniform, path-tracing mode only): rgen:99-102 adds `contrib` when the shadow ray is
    // not occluded; a contribution of exactly (+-0, +-0, +-0) -- light behind the surface and no emission, or a zero weight --
    // leaves hitValue bit for bit as it is either way (x + 0 = x; the sum is never -0), so the ray need not be traced
    const bool dead = !HYBRID && (P.flags & VKRT_FLAG_SKIP_DEAD_SHADOW) != 0u && contrib.x == 0.0f && contrib.y == 0.0f && contrib.z == 0.0f;
    if(!L.prd.isSpecular && L.prd.depth != 100u && !dead)  // rgen:79: a shadow ray decides whether this segment contributes
      to = (L.prd.depth + 1u < (uint32_t)P.pc.depth) ? WF_P : WF_S;  // not the last segment: the next closest-hit ray rides along
    else if(HYBRID)
      to = advanceSegmentHybrid(P, G, L, false, contrib, nextWeight, 0.0f) ? WF_C : -1;
    else
      to = advanceSegment(P, L, false, contrib, nextWeight) ? WF_C : -1;
  }
  const unsigned slot = claimSlots(B, par ^ 1, to, lane, wsum);
  if(to == WF_C)
    storeClosest(B, par ^ 1, slot, L);
  else if(to >= 0)
    storeShadow(B, par ^ 1, to, slot, L, contrib, nextWeight);
  if(P.flags & VKRT_FLAG_COUNT_WORK)  // launch-uniform: hit / lobe / texture-tap tallies are instrumentation, not needed for the ray rate
  {
    __shared__ unsigned long long red[VKRT_COUNTER_STRIDE * (WF_BLOCK / 64)];
    const unsigned vals[5] = {0, 0, st.hits, st.diffuse, st.taps};
    blockAddCounters(&P.counters->v[blockIdx.x % VKRT_COUNTER_SLOTS][0], vals, 5, red);
  }
}

// ---- shade, results of stream S: the last segment of a sample (rgen:99-120), next sample or pixel store (light; many waves) ----
template <bool HYBRID>
VKRT_DEV void shadeShadowBlock(const TraceParams& P, const WfBuffers& B, const HybridGi& G, const int par, const unsigned count, const unsigned block, ShadeLds& lds)
{
  const unsigned lane = lane_id();
  const unsigned qi = block * WF_BLOCK + threadIdx.x;
  unsigned* wsum = lds.wsum;
  bool toClosest = false;
  LaneState L;
  if(qi < count)
  {
    loadCommon(P, B, par, WF_S, qi, L);  // S0 holds the weight after this segment
    const float4 h = wfLoad(rec(B, par, WF_S, WF_H0, qi)), s3 = wfLoad(rec(B, par, WF_S, WF_S3, qi));
    L.prd.rayOrigin = mk3(0.0f);     // the sample ends here: startSample sets the next ray, or the pixel is stored
    L.prd.rayDirection = mk3(0.0f);
    const bool shadowHit = __float_as_int(h.w) >= 0;
    if(HYBRID)
      toClosest = advanceSegmentHybrid(P, G, L, shadowHit, mk3(s3.x, s3.y, s3.z), L.curWeight, L.prd.depth == 1u ? wfLoad(rec(B, par, WF_S, WF_R1, qi)).w : 0.0f);
    else
      toClosest = advanceSegment(P, L, shadowHit, mk3(s3.x, s3.y, s3.z), L.curWeight);
  }
  const unsigned slot = claimSlots(B, par ^ 1, toClosest ? WF_C : -1, lane, wsum);
  if(toClosest)
    storeClosest(B, par ^ 1, slot, L);
}

// One launch shades the three result streams of a round: the heavy workgroups (C, then P) are dispatched first, the light
// shadow-result workgroups fill in behind them.
template <bool HYBRID>
VKRT_DEV void shadeRound(const TraceParams& P, const WfBuffers& B, const HybridGi& G, const int round)
{
  const int par = round & 1;
  const unsigned cC = *countOf(B, par, WF_C), cS = *countOf(B, par, WF_S), cP = *countOf(B, par, WF_P);
  const unsigned nC = (cC + WF_BLOCK - 1) / WF_BLOCK, nP = (cP + WF_BLOCK - 1) / WF_BLOCK, nS = (cS + WF_BLOCK - 1) / WF_BLOCK;
  unsigned blk = blockIdx.x;
  __shared__ ShadeLds lds;
  if(blk < nC)
    shadeHitBlock<false, HYBRID>(P, B, G, par, cC, blk, lds);
  else if((blk -= nC) < nP)
    shadeHitBlock<true, HYBRID>(P, B, G, par, cP, blk, lds);
  else if((blk -= nP) < nS)
    shadeShadowBlock<HYBRID>(P, B, G, par, cS, blk, lds);
}
__global__ __launch_bounds__(WF_BLOCK) void k_wf_shade(const TraceParams P, const WfBuffers B, const int round)
{
  const HybridGi none{};
  shadeRound<false>(P, B, none, round);
}
__global__ __launch_bounds__(WF_BLOCK) void k_wf_shade_hybrid(const TraceParams P, const WfBuffers B, const HybridGi G, const int round)
{
  shadeRound<true>(P, B, G, round);
}

// First ray of the GI path of every shaded pixel (raytraceHybrid.rgen:172-204), or the pixel's final value when there is no path
// to trace.  tmp: (seed after the direct part, visibility) per pixel, left by k_hybrid.
__global__ __launch_bounds__(WF_BLOCK) void k_hy_gi_init(const TraceParams P, const WfBuffers B, const HybridGi G, const uint2* tmp)
{
  const unsigned lane = lane_id();
  const unsigned w = blockIdx.x * blockDim.x + threadIdx.x;  // tile-major work index
  bool alive = false;
  LaneState L;
  if(w < P.tileCount * 64u)
  {
    const unsigned tile = P.tileFirst + (w >> 6), inTile = w & 63u;
    const uint32_t x = (tile % P.tilesX) * 8u + (inTile & 7u);
    const uint32_t lrow = (tile / P.tilesX) * 8u + (inTile >> 3);
    if(x < P.fullW && lrow < P.localRows && globalRow(P, lrow) < P.fullH)
    {
      const size_t p = (size_t)lrow * P.fullW + x;
      const float4 pixelImg = G.color[p], pixelPos = G.position[p], pixelNorm = G.normal[p];
      const float2 rm = G.rough[p];
      const uint2 t = tmp[p];
      const f3 worldPos = mk3(pixelPos.x, pixelPos.y, pixelPos.z), worldNrm = mk3(pixelNorm.x, pixelNorm.y, pixelNorm.z);
      const bool shaded = !(worldPos.x == 0.0f && worldPos.y == 0.0f && worldPos.z == 0.0f && worldNrm.x == 0.0f && worldNrm.y == 0.0f &&
                            worldNrm.z == 0.0f);  // rgen:67
      const float alpha = __uint_as_float(t.y);
      if(!shaded)
        hybridFinish(P, G, x, lrow, false, mk3(0.0f), 0.0f, alpha);
      else
      {
        L.px = x; L.lrow = lrow;
        L.prd.seed = t.x;
        L.prd.lightDist = 0.0f; L.prd.shadowRayDir = mk3(0.0f);
        const float roughness = rm.x, metalness = rm.y;
        const float ratio = metalness * (1.0f - roughness);
        f3 direction;
        if(ratio < 0.8f)
        {
          L.prd.isSpecular = false;
          f3 tangent, binormal;
          createCoordinateSystem(worldNrm, tangent, binormal);
          direction = normalize3(samplingHemisphere(L.prd.seed, tangent, binormal, worldNrm));
          L.curWeight = mk3(pixelImg.w, pixelPos.w, pixelNorm.w);  // albedo
        }
        else
        {
          L.prd.isSpecular = true;
          float cam[4];
          mat4MulVec4(P.viewInverse, 0.0f, 0.0f, 0.0f, 1.0f, cam);
          const f3 V = normalize3(mk3(cam[0], cam[1], cam[2]) - worldPos);
          direction = normalize3(glsl_reflect(-V, worldNrm));
          L.curWeight = mk3(1.0f);
        }
        L.prd.hitValue = mk3(0.0f);
        L.prd.rayOrigin = worldPos;
        L.prd.rayDirection = direction;
        L.prd.depth = 1;
        L.prd.weight = mk3(0.0f);
        L.hitValue = mk3(0.0f);
        L.hitValues = mk3(0.0f, alpha, 0.0f);  // .x hitDists, .y visibility
        L.smpl = 0;
        L.stage = 0;
        if(L.prd.depth < (uint32_t)P.pc.depth)
          alive = true;
        else
          hybridFinish(P, G, x, lrow, true, mk3(0.0f), 0.0f, alpha);
      }
    }
  }
  __shared__ unsigned wsum[WF_TYPES * (WF_BLOCK / 64 + 1)];
  const unsigned slot = claimSlots(B, 0, alive ? WF_C : -1, lane, wsum);
  if(alive)
    storeClosest(B, 0, slot, L);
}

// ---- frames in flight: ordered blend of a staged frame (raytrace.rgen:136-145) ---------------------------------------------------
// One thread per pixel of the lane's tiles (tile-major, as k_wf_init): image = mix(image, staged, 1 / (frame + 1)), or the plain
// store of frame 0.  Launched at the end of a frame on the frame's own lane, behind the blend of the frame before it.
__global__ __launch_bounds__(WF_BLOCK) void k_wf_blend(const TraceParams P, const float4* stage)
{
  const unsigned w = blockIdx.x * blockDim.x + threadIdx.x;
  if(w >= P.tileCount * 64u)
    return;
  const unsigned tile = P.tileFirst + (w >> 6), inTile = w & 63u;
  const uint32_t x = (tile % P.tilesX) * 8u + (inTile & 7u);
  const uint32_t lrow = (tile / P.tilesX) * 8u + (inTile >> 3);
  if(x < P.fullW && lrow < P.localRows && globalRow(P, lrow) < P.fullH)
  {
    const size_t p = (size_t)lrow * P.fullW + x;
    blendPixel((float4*)P.image + p, stage[p], P.pc.frame);
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------
#define WF_CTRL_BYTES (256 * VKRT_WF_MAX_LANES)  // 64 count words per lane
size_t vkrt_wf_state_bytes(uint32_t pathCapacity, int groups)
{
  const size_t g = (size_t)std::max(groups, 1);
  return g * pathCapacity * 2 * WF_SLOTS * sizeof(float4) + (g > 1 ? g * pathCapacity * sizeof(float4) : 0) + WF_CTRL_BYTES;
}

void vkrt_wf_carve(void* base, uint32_t pathCapacity, int groups, WfBuffers* B)
{
  const size_t g = (size_t)std::max(groups, 1);
  char* p = (char*)base;
  B->ctrl = (unsigned*)p;
  p += WF_CTRL_BYTES;
  B->planes = (float4*)p;
  p += g * pathCapacity * 2 * WF_SLOTS * sizeof(float4);
  B->stage = g > 1 ? (float4*)p : nullptr;
  B->capacity = pathCapacity;
  B->groups = (uint32_t)g;
}

// One sub-frame = the tiles [tileFirst, tileFirst + tileCount) of the shard with their own streams and counts, on one HIP stream.
// subframeBegin: counts cleared, one closest-ray record per pixel (raytrace.rgen:27-60).
static hipError_t subframeBegin(const TraceParams& P, const WfBuffers& B, hipStream_t stream)
{
  const unsigned work = P.tileCount * 64u;
  const hipError_t e = hipMemsetAsync(B.ctrl, 0, 64, stream);
  if(e != hipSuccess)
    return e;
  hipLaunchKernelGGL(k_wf_init, dim3((work + WF_BLOCK - 1) / WF_BLOCK), dim3(WF_BLOCK), 0, stream, P, B);
  return hipSuccess;
}
// a sample takes at most depth + 1 rounds: its first closest-hit ray, then one round per further segment (the shadow ray of
// segment k travels with the closest-hit ray of segment k + 1), then the shadow ray of its last segment
static int subframeRounds(const TraceParams& P) { return (P.pc.samples <= 0 || P.pc.depth <= 0) ? 0 : P.pc.samples * (P.pc.depth + 1); }
// round r: traverse the rays of the records, shade the results into the next round's records
static void subframeRound(const TraceParams& P, const WfBuffers& B, int r, unsigned travBlock, bool count, hipStream_t stream, WfTiming* timing)
{
  const unsigned work = P.tileCount * 64u;
  // every path holds one record and a record at most two rays; +4 blocks for the partial tails of the four ray kinds.
  // One wavefront per workgroup by default: a finished wave frees its slot and LDS without waiting for three others.
  const dim3 tg(2 * ((work + travBlock - 1) / travBlock) + 4);
  const size_t tlds = (size_t)P.sc.stackCap * travBlock * sizeof(int);
  const bool timed = timing && timing->events && 2 * (timing->used + 1) <= timing->capacity;
  if(timed)
    (void)hipEventRecord(timing->events[2 * timing->used], stream);
  vkrt_wf_launch_traverse(P, B, r, travBlock, count, tg, tlds, stream);
  if(timed)
  {
    (void)hipEventRecord(timing->events[2 * timing->used + 1], stream);
    timing->used++;
  }
  hipLaunchKernelGGL(k_wf_shade, dim3((work + WF_BLOCK - 1) / WF_BLOCK + 3), dim3(WF_BLOCK), 0, stream, P, B, r);
}

// parameters of frame k of a call (progressive frames of an unchanged camera: main.cpp:503-508, hello_vulkan.cpp:1501-1521)
static TraceParams frameParams(const TraceParams& P, int k, uint32_t seedStep)
{
  TraceParams Q = P;
  Q.pc.frame = P.pc.frame + k;
  Q.seed = P.seed + (uint32_t)k * seedStep;
  return Q;
}

// frames a call of `frames` frames keeps in flight when the option allows `want`: as many rounds of `want` as the call needs, the frames
// dealt evenly over them (4 frames, want 3 -> two rounds of 2, not 3 + 1: measured 30.8 against 31.5 ms per frame on a 4K / 8 shard)
static int balancedInFlight(int frames, int want)
{
  want = std::max(1, std::min(want, frames));
  const int turns = (frames + want - 1) / want;
  return (frames + turns - 1) / turns;
}

hipError_t vkrt_launch_wavefront(const TraceParams& P, const WfBuffers& B, const WfOptions& opt, int frames, uint32_t seedStep, bool count,
                                 hipStream_t stream, WfTiming* timing, const WfAsync* async)
{
  const unsigned travBlock = opt.travBlock == 256 ? 256u : opt.travBlock == 128 ? 128u : 64u;
  const int rounds = subframeRounds(P);
  frames = std::max(frames, 1);
  // Lanes: F frame groups x S tile ranges.  Frames in flight keep every launch at full size, sub-frames cut it into S pieces -- and
  // what overlaps well on this device is few, large launches (profiles/r04_experiments.md #110: two or three full-size lanes reach
  // 0.96-0.97 of linear on a 4K / 8 shard, three third-size lanes 0.90, four quarter-size lanes 0.77): a call with several frames
  // runs them in flight, one lane each; a single frame is split into sub-frames.  Per-kernel timing wants the kernels one after
  // another; tiny frames are not worth splitting.
  int F = (timing || !async || rounds == 0) ? 1 : std::min(balancedInFlight(frames, opt.inFlight), (int)B.groups);
  F = std::max(1, std::min(F, async ? std::min(VKRT_WF_MAX_LANES, async->count) : 1));
  int S = (timing || !async || F > 1) ? 1 : std::max(1, std::min(std::min(VKRT_WF_MAX_LANES, opt.subframes), async->count));
  S = (int)std::min<uint32_t>((uint32_t)S, std::max(1u, P.tileCount / 256u));
  const int L = S * F;
  hipError_t e;
  if(L <= 1)
  {
    for(int k = 0; k < frames; k++)
    {
      TraceParams Q = frameParams(P, k, seedStep);
      Q.tileFirst = 0;
      if((e = subframeBegin(Q, B, stream)) != hipSuccess) return e;
      for(int r = 0; r < rounds; r++) subframeRound(Q, B, r, travBlock, count, stream, timing);
    }
    return hipGetLastError();
  }
  // ---- several lanes ------------------------------------------------------------------------------------------------------
  // lane q = g * S + j: tile range j of the frames of group g (frames g, g + F, g + 2 F, ...)
  const bool staged = F > 1;
  if(staged && vkrt_wf_pool_events(frames, S) > async->poolSize)
    return hipErrorInvalidValue;  // (the caller sizes the pool: vkrt_api.cpp)
  hipStream_t laneStream[VKRT_WF_MAX_LANES];
  uint32_t tile0[VKRT_WF_MAX_LANES], tileN[VKRT_WF_MAX_LANES];
  WfBuffers Bq[VKRT_WF_MAX_LANES];
  const size_t groupQuads = (size_t)2 * WF_SLOTS * B.capacity;
  for(int q = 0; q < L; q++)
  {
    const int g = q / S, j = q % S;
    tile0[q] = (uint32_t)((uint64_t)P.tileCount * j / S);
    tileN[q] = (uint32_t)((uint64_t)P.tileCount * (j + 1) / S) - tile0[q];
    Bq[q].ctrl = B.ctrl + 64 * q;
    Bq[q].planes = B.planes + (size_t)g * groupQuads + (size_t)2 * WF_SLOTS * ((size_t)tile0[q] * 64u);
    Bq[q].stage = staged ? B.stage + (size_t)g * B.capacity : nullptr;
    Bq[q].capacity = tileN[q] * 64u;
    Bq[q].groups = 1;
    laneStream[q] = async->streams[q];
  }
  // fork: every lane starts behind what the caller enqueued before this call; whatever happens afterwards, the lanes that were
  // started are joined to the caller's stream again before the function returns (nothing of a failed call runs on unordered)
  int forked = 0;
  auto joinLanes = [&]() {
    hipError_t first = hipSuccess;
    for(int q = 0; q < forked; q++)
    {
      hipError_t x = hipEventRecord(async->join[q], laneStream[q]);
      if(x == hipSuccess) x = hipStreamWaitEvent(stream, async->join[q], 0);
      if(x != hipSuccess)
      {
        (void)hipStreamSynchronize(laneStream[q]);
        if(first == hipSuccess) first = x;
      }
    }
    return first;
  };
  if((e = hipEventRecord(async->fork, stream)) != hipSuccess) return e;
  for(int q = 0; q < L; q++)
  {
    if((e = hipStreamWaitEvent(laneStream[q], async->fork, 0)) != hipSuccess)
    {
      (void)joinLanes();
      return e;
    }
    forked = q + 1;
  }
  // Items of a lane's m-th frame: 0 = begin, 1..rounds = the rounds, rounds + 1 = ordered blend (frames in flight only).  The host
  // deals the items of all lanes round-robin -- item i of every lane before item i + 1 of any: enqueueing one lane's hundreds of
  // launches after the other's makes the later lanes start milliseconds late (profiles/r03_experiments.md #106).  The blend of
  // frame k waits for the blend of frame k - 1 (same tiles, another lane) through an event of the pool, one per (frame, tile
  // range), so no record call ever replaces one that still has a wait to come; a lane whose blend would wait for a record the
  // host has not made yet is skipped for a turn (hipStreamWaitEvent refers to the record calls made BEFORE it).
  const int items = rounds + 1 + (staged ? 1 : 0);
  hipEvent_t* evBlend = async->pool;  // [frame k][tile range j]
  int laneFrames[VKRT_WF_MAX_LANES], cursor[VKRT_WF_MAX_LANES];
  std::vector<char> blendRecorded(staged ? (size_t)frames * S : 0, 0);
  int remaining = 0;
  for(int q = 0; q < L; q++)
  {
    laneFrames[q] = (frames - q / S + F - 1) / F;
    cursor[q] = 0;
    remaining += laneFrames[q] * items;
  }
  e = hipSuccess;
  while(remaining > 0 && e == hipSuccess)
  {
    bool progressed = false;
    for(int q = 0; q < L && e == hipSuccess; q++)
    {
      if(cursor[q] >= laneFrames[q] * items)
        continue;
      const int g = q / S, j = q % S, m = cursor[q] / items, it = cursor[q] % items, k = g + m * F;
      TraceParams Q = frameParams(P, k, seedStep);
      Q.tileFirst = tile0[q];
      Q.tileCount = tileN[q];
      if(staged)
      {
        Q.image = (float*)Bq[q].stage;
        Q.flags |= VKRT_FLAG_STORE_STAGED;
      }
      hipStream_t st = laneStream[q];
      if(it == 0)
      {
        if((e = subframeBegin(Q, Bq[q], st)) != hipSuccess) break;
      }
      else if(it <= rounds)
        subframeRound(Q, Bq[q], it - 1, travBlock, count, st, nullptr);
      else
      {
        // the frame's pixels are staged: blend them into the image behind the blend of frame k - 1
        if(k > 0)
        {
          if(!blendRecorded[(size_t)(k - 1) * S + j])
            continue;
          if((e = hipStreamWaitEvent(st, evBlend[(k - 1) * S + j], 0)) != hipSuccess) break;
        }
        TraceParams R = Q;
        R.image = P.image;
        const unsigned work = R.tileCount * 64u;
        hipLaunchKernelGGL(k_wf_blend, dim3((work + WF_BLOCK - 1) / WF_BLOCK), dim3(WF_BLOCK), 0, st, R, (const float4*)Bq[q].stage);
        if(k + 1 < frames && (e = hipEventRecord(evBlend[k * S + j], st)) != hipSuccess) break;
        blendRecorded[(size_t)k * S + j] = 1;
      }
      cursor[q]++;
      remaining--;
      progressed = true;
    }
    if(!progressed && e == hipSuccess)
      e = hipErrorUnknown;  // (cannot happen: the blends follow frame order)
  }
  const hipError_t ej = joinLanes();
  if(e == hipSuccess) e = ej;
  if(e == hipSuccess) e = hipGetLastError();
  return e;
}

// ---- hybrid GI -------------------------------------------------------------------------------------------------------
uint2* vkrt_wf_hybrid_tmp(const WfBuffers& B)
{
  // the streams of parity 1 are first written by the shade step of round 0: until then their first plane is free
  return (uint2*)(B.planes + (size_t)(1 * WF_SLOTS + 0) * B.capacity);
}

hipError_t vkrt_launch_hybrid_gi(const TraceParams& P, const WfBuffers& B, const HybridGi& G, unsigned travBlock, hipStream_t stream)
{
  const unsigned work = P.tileCount * 64u;
  hipError_t e = hipMemsetAsync(B.ctrl, 0, 64, stream);
  if(e != hipSuccess)
    return e;
  const unsigned blocks = (work + WF_BLOCK - 1) / WF_BLOCK;
  hipLaunchKernelGGL(k_hy_gi_init, dim3(blocks), dim3(WF_BLOCK), 0, stream, P, B, G, (const uint2*)vkrt_wf_hybrid_tmp(B));
  // the GI path is one sample that starts at depth 1: at most pc.depth - 1 segments, i.e. pc.depth rounds of the paired pipeline
  const int rounds = P.pc.depth;
  const unsigned tbs = (travBlock == 256u || travBlock == 128u) ? travBlock : 64u;
  const dim3 tg(2 * ((work + tbs - 1) / tbs) + 4);
  const size_t tlds = (size_t)P.sc.stackCap * tbs * sizeof(int);
  for(int r = 0; r < rounds; r++)
  {
    vkrt_wf_launch_traverse(P, B, r, tbs, false, tg, tlds, stream);
    hipLaunchKernelGGL(k_wf_shade_hybrid, dim3(blocks + 3), dim3(WF_BLOCK), 0, stream, P, B, G, r);
  }
  return hipGetLastError();
}
