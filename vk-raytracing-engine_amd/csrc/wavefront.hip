// wavefront.hip -- the path-trace dispatch as a wavefront pipeline (default execution mode).
// Replaces vkCmdTraceRaysKHR(W,H,1) over raytrace.rgen/.rchit/.rmiss/raytraceShadow.rmiss
// (reference hello_vulkan.cpp:1446) exactly like pathtrace.hip, with the same per-path state machine
// (rgen.h) and therefore the same results; the work is re-scheduled for gfx950:
//
//   k_wf_init      one thread per pixel: rgen prologue (seed, camera ray of sample 0) -> closest-ray queue.
//   k_wf_traverse  one thread per queued ray, batch-synchronous: the 64 rays of a wave start together, so the
//                  top tree levels are fetched as coalesced/broadcast loads.  Workgroups are homogeneous:
//                  the first blocks take the closest-hit queue (rgen:64-75), the rest the shadow queue
//                  (rgen:85-97, any-hit).  Per-lane stacks in LDS.  Software stand-in for traceRayEXT.
//   k_wf_shade     one thread per finished ray, again split by type: closest-hit results run rchit / rmiss
//                  and either request a shadow ray or accumulate; shadow results accumulate (rgen:99-120).
//                  Surviving paths are appended to the next round's closest / shadow queue with a
//                  block-aggregated ballot compaction (one global atomic per workgroup).
//
// A frame = init + up to samples*depth*2 (traverse, shade) rounds enqueued back to back on the caller's
// stream; queue counts stay on the device (no host synchronisation inside a frame).  Path state lives in
// HBM as 16-byte SoA records indexed by path id.
//
// Measured alternatives (profiles/r01_experiments.md): refilling single lanes from the queue inside the
// traversal loop (ballot + wave-aggregated atomic) is 4.5x slower on MI355X because it desynchronises the
// lanes of a wave and every node load then touches 64 cache lines; the ballot compaction therefore sits
// at the queue boundary.
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

#define WF_BLOCK 256

// ---- path state in HBM: one 192-byte record per path (three 64-byte lines) ---------------------------------------
//   line 0: Q0 ray origin.xyz, tmax | Q1 ray direction.xyz, anyHit | Q2 hit t,u,v,slot | Q3 triShade[slot] (closest hits)
//   line 1: Q4 rayOrigin.xyz, lightDist | Q5 rayDirection.xyz, seed | Q6 shadowRayDir.xyz, flags | Q7 prd.hitValue.xyz, px|lrow
//   line 2: Q8 prd.weight.xyz | Q9 curWeight.xyz | Q10 hitValue.xyz | Q11 hitValues.xyz
// Traversal touches line 0 only; queue order gets scrambled by the per-type compaction, so records (not SoA
// planes) keep every 16-byte lane access inside a line the lane uses completely.
// flags: depth[0:8) | smpl[8:24) | stage[24] | isSpecular[25]
#define WF_REC_QUADS 12
VKRT_DEV float4* rec(const WfBuffers& B, unsigned p) { return B.rec + (size_t)p * WF_REC_QUADS; }

VKRT_DEV void storeState(const WfBuffers& B, unsigned p, const LaneState& L)
{
  const unsigned flags = (L.prd.depth & 0xffu) | (((unsigned)L.smpl & 0xffffu) << 8) | ((unsigned)L.stage << 24) |
                         ((L.prd.isSpecular ? 1u : 0u) << 25);
  float4* r = rec(B, p);
  r[4] = make_float4(L.prd.rayOrigin.x, L.prd.rayOrigin.y, L.prd.rayOrigin.z, L.prd.lightDist);
  r[5] = make_float4(L.prd.rayDirection.x, L.prd.rayDirection.y, L.prd.rayDirection.z, __uint_as_float(L.prd.seed));
  r[6] = make_float4(L.prd.shadowRayDir.x, L.prd.shadowRayDir.y, L.prd.shadowRayDir.z, __uint_as_float(flags));
  r[7] = make_float4(L.prd.hitValue.x, L.prd.hitValue.y, L.prd.hitValue.z, __uint_as_float(L.px | (L.lrow << 16)));
  r[8] = make_float4(L.prd.weight.x, L.prd.weight.y, L.prd.weight.z, 0.0f);
  r[9] = make_float4(L.curWeight.x, L.curWeight.y, L.curWeight.z, 0.0f);
  r[10] = make_float4(L.hitValue.x, L.hitValue.y, L.hitValue.z, 0.0f);
  r[11] = make_float4(L.hitValues.x, L.hitValues.y, L.hitValues.z, 0.0f);
}

VKRT_DEV void loadState(const TraceParams& P, const WfBuffers& B, unsigned p, LaneState& L)
{
  const float4* r = rec(B, p);
  const float4 s0 = r[4], s1 = r[5], s2 = r[6], s3 = r[7];
  const float4 s4 = r[8], s5 = r[9], s6 = r[10], s7 = r[11];
  const unsigned flags = __float_as_uint(s2.w), pix = __float_as_uint(s3.w);
  L.prd.rayOrigin = mk3(s0.x, s0.y, s0.z); L.prd.lightDist = s0.w;
  L.prd.rayDirection = mk3(s1.x, s1.y, s1.z); L.prd.seed = __float_as_uint(s1.w);
  L.prd.shadowRayDir = mk3(s2.x, s2.y, s2.z);
  L.prd.depth = flags & 0xffu; L.smpl = (int)((flags >> 8) & 0xffffu); L.stage = (int)((flags >> 24) & 1u);
  L.prd.isSpecular = ((flags >> 25) & 1u) != 0u;
  L.prd.hitValue = mk3(s3.x, s3.y, s3.z);
  L.px = pix & 0xffffu; L.lrow = pix >> 16; L.py = globalRow(P, L.lrow);
  L.prd.weight = mk3(s4.x, s4.y, s4.z);
  L.curWeight = mk3(s5.x, s5.y, s5.z);
  L.hitValue = mk3(s6.x, s6.y, s6.z);
  L.hitValues = mk3(s7.x, s7.y, s7.z);
  float origin[4];
  mat4MulVec4(P.viewInverse, 0.0f, 0.0f, 0.0f, 1.0f, origin);  // rgen:30 (uniform; cheaper to recompute than to store)
  L.camOrigin = mk3(origin[0], origin[1], origin[2]);
}

// the ray the traversal kernel traces next for this path (rgen:64-75 or :85-97)
VKRT_DEV void storeRay(const WfBuffers& B, unsigned p, const LaneState& L)
{
  float4* r = rec(B, p);
  if(L.stage == 1)
  {
    r[0] = make_float4(L.prd.rayOrigin.x, L.prd.rayOrigin.y, L.prd.rayOrigin.z, L.prd.lightDist - 0.1f);
    r[1] = make_float4(L.prd.shadowRayDir.x, L.prd.shadowRayDir.y, L.prd.shadowRayDir.z, __uint_as_float(1u));
  }
  else
  {
    r[0] = make_float4(L.prd.rayOrigin.x, L.prd.rayOrigin.y, L.prd.rayOrigin.z, 10000.0f);
    r[1] = make_float4(L.prd.rayDirection.x, L.prd.rayDirection.y, L.prd.rayDirection.z, __uint_as_float(0u));
  }
}

// Block-aggregated append to the next round's queues: ballot + popcount inside each wave, wave totals combined
// through LDS, ONE global atomic per workgroup and queue (the count is a single word: per-wave atomics
// serialise near 88/us, MI355X_MICROARCH.md "dequeue").  Must be called by every thread of the block.
// A thread sets at most one of toC / toS.  wsum: LDS [2*(nw+1)].
VKRT_DEV void appendQueues(unsigned* qC, unsigned* cntC, unsigned* qS, unsigned* cntS, bool toC, bool toS, unsigned value, unsigned lane,
                           unsigned* wsum)
{
  const unsigned wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const unsigned long long mC = __ballot(toC), mS = __ballot(toS);
  if(lane == 0)
  {
    wsum[wave] = (unsigned)__popcll(mC);
    wsum[nw + 1 + wave] = (unsigned)__popcll(mS);
  }
  __syncthreads();
  if(threadIdx.x < 2)
  {
    unsigned* w = wsum + threadIdx.x * (nw + 1);
    unsigned tot = 0;
    for(unsigned k = 0; k < nw; k++)
    {
      const unsigned c = w[k];
      w[k] = tot;
      tot += c;
    }
    w[nw] = tot ? atomicAdd(threadIdx.x == 0 ? cntC : cntS, tot) : 0u;
  }
  __syncthreads();
  const unsigned long long below = (1ull << lane) - 1ull;
  if(toC)
    qC[wsum[nw] + wsum[wave] + (unsigned)__popcll(mC & below)] = value;
  if(toS)
    qS[wsum[2 * nw + 1] + wsum[nw + 1 + wave] + (unsigned)__popcll(mS & below)] = value;
}

// ctrl words: [parity*2 + type] = queue count (type 0 closest, 1 shadow)
VKRT_DEV unsigned* qPtr(const WfBuffers& B, int parity, int type) { return B.queue[parity * 2 + type]; }

// ---- init: raytrace.rgen:27-60 for every pixel of the shard -------------------------------------------------
__global__ __launch_bounds__(WF_BLOCK) void k_wf_init(const TraceParams P, const WfBuffers B)
{
  const unsigned lane = lane_id();
  const unsigned w = blockIdx.x * blockDim.x + threadIdx.x;  // tile-major work index
  bool alive = false;
  unsigned nPixels = 0;
  if(w < P.tileCount * 64u)
  {
    const unsigned tile = w >> 6, inTile = w & 63u;
    const uint32_t x = (tile % P.tilesX) * 8u + (inTile & 7u);
    const uint32_t lrow = (tile / P.tilesX) * 8u + (inTile >> 3);
    if(x < P.fullW && lrow < P.localRows)
    {
      const uint32_t y = globalRow(P, lrow);
      if(y < P.fullH)
      {
        LaneState L;
        startPixel(P, L, x, y, lrow);
        nPixels = 1;
        if(P.pc.samples <= 0 || P.pc.depth <= 0)
          storePixel(P, L);  // degenerate launch: no rays
        else
        {
          storeState(B, w, L);
          storeRay(B, w, L);
          alive = true;
        }
      }
    }
  }
  __shared__ unsigned wsum[2 * (WF_BLOCK / 64 + 1)];
  appendQueues(qPtr(B, 0, 0), &B.ctrl[0], qPtr(B, 0, 1), &B.ctrl[1], alive, false, w, lane, wsum);
  __shared__ unsigned long long red[VKRT_COUNTER_STRIDE * (WF_BLOCK / 64)];
  const unsigned vals[6] = {0, 0, 0, 0, 0, nPixels};
  blockAddCounters(&P.counters->v[blockIdx.x % VKRT_COUNTER_SLOTS][0], vals, 6, red);
}

// ---- traversal: one thread per queued ray, workgroups homogeneous in ray type -----------------------------------
template <bool COUNT, bool WIDE, int TB>
__global__ __launch_bounds__(TB) void k_wf_traverse(const TraceParams P, const WfBuffers B, const int round)
{
  extern __shared__ int lds_stack[];
  const int par = round & 1;
  const unsigned countC = B.ctrl[par * 2 + 0], countS = B.ctrl[par * 2 + 1];
  if(blockIdx.x == 0 && threadIdx.x == 0)
  {
    B.ctrl[(par ^ 1) * 2 + 0] = 0u;  // next round's counts; this round's k_wf_shade appends to them
    B.ctrl[(par ^ 1) * 2 + 1] = 0u;
  }
  const unsigned nbC = (countC + TB - 1) / TB, nbS = (countS + TB - 1) / TB;
  if(blockIdx.x >= nbC + nbS)
    return;
  const bool anyHit = blockIdx.x >= nbC;  // workgroup-uniform
  const unsigned qi = (anyHit ? blockIdx.x - nbC : blockIdx.x) * TB + threadIdx.x;
  const unsigned count = anyHit ? countS : countC;
  unsigned nRays = 0;
  TravCount tc;
  if(qi < count)
  {
    const unsigned pid = qPtr(B, par, anyHit ? 1 : 0)[qi];
    float4* r = rec(B, pid);
    const float4 r0 = r[0], r1 = r[1];
    RayHit hit;
    traverse_any<COUNT, WIDE>(P.sc, mk3(r0.x, r0.y, r0.z), mk3(r1.x, r1.y, r1.z), 0.001f, r0.w, anyHit, lds_stack, (int)threadIdx.x, TB, hit,
                              tc);
    r[2] = make_float4(hit.t, hit.u, hit.v, __int_as_float(hit.slot));
    if(!anyHit && hit.slot >= 0)
    {
      // first hop of the hit shader's attribute fetch, taken here so k_wf_shade_closest finds it in the line it reads anyway
      const uint4 ts = P.sc.triShade[hit.slot];
      r[3] = make_float4(__uint_as_float(ts.x), __uint_as_float(ts.y), __uint_as_float(ts.z), __uint_as_float(ts.w));
    }
    nRays = 1;
  }
  __shared__ unsigned long long red[VKRT_COUNTER_STRIDE * (TB / 64)];
  const unsigned vals[10] = {anyHit ? 0u : nRays, anyHit ? nRays : 0u, 0, 0, 0, 0, tc.nodes, tc.tris, tc.waveNodeSteps, tc.waveTriSteps};
  blockAddCounters(&P.counters->v[blockIdx.x % VKRT_COUNTER_SLOTS][0], vals, COUNT ? 10 : 2, red);
}

// ---- traversal variant: persistent workgroups, idle lanes refilled from the queue inside the loop ------------------
// (wide8 only; VKRT_WF_TRAVERSE=refill).  Workgroups are still homogeneous in ray type; a wave owns a private
// chunk of queue slots (one global atomic per WF_CHUNK rays) and refills when >= refillMin lanes are idle.
#define WF_CHUNK 128u
template <bool COUNT>
__global__ __launch_bounds__(WF_BLOCK) void k_wf_traverse_refill(const TraceParams P, const WfBuffers B, const int round, const unsigned refillMin)
{
  extern __shared__ int lds_stack[];
  uint2* stk = ((uint2*)lds_stack) + threadIdx.x;
  const unsigned lane = lane_id();
  const int par = round & 1;
  const unsigned countC = B.ctrl[par * 2 + 0], countS = B.ctrl[par * 2 + 1];
  if(blockIdx.x == 0 && threadIdx.x == 0)
  {
    B.ctrl[(par ^ 1) * 2 + 0] = 0u;
    B.ctrl[(par ^ 1) * 2 + 1] = 0u;
  }
  // split the persistent grid between the two queues in proportion to their lengths
  const unsigned total = countC + countS;
  if(total == 0u)
    return;
  unsigned blocksS = (unsigned)(((unsigned long long)gridDim.x * countS + total - 1u) / total);
  if(countS == 0u) blocksS = 0u;
  if(countC != 0u && blocksS >= gridDim.x) blocksS = gridDim.x - 1u;
  const bool anyHit = blockIdx.x < blocksS;
  const unsigned count = anyHit ? countS : countC;
  const unsigned* __restrict__ queue = qPtr(B, par, anyHit ? 1 : 0);
  unsigned* cursor = &B.ctrl[4 + (anyHit ? 1 : 0)];

  bool active = false, exhausted = false;
  unsigned chunkNext = 0, chunkEnd = 0, pid = 0;
  W8State S;
  S.G = make_uint2(0u, 0u); S.sp = 0; S.steps = 0; S.anyHit = anyHit;
  S.o = mk3(0.0f); S.d = mk3(0.0f); S.id = mk3(0.0f); S.tmax = 0.0f; S.bestT = 0.0f; S.bestU = 0.0f; S.bestV = 0.0f; S.bestSlot = -1; S.bestGid = -1;
  unsigned nRays = 0;
  TravCount tc;
  for(;;)
  {
    const unsigned long long idleMask = __ballot(!active);
    if(idleMask != 0ull && (chunkNext < chunkEnd || !exhausted))
    {
      const unsigned nIdle = (unsigned)__popcll(idleMask);
      if(nIdle >= refillMin || idleMask == ~0ull)
      {
        const unsigned rank = (unsigned)__popcll(idleMask & ((1ull << lane) - 1ull));
        const unsigned avail = chunkEnd - chunkNext;
        unsigned qi = 0xffffffffu;
        if(nIdle <= avail)
        {
          qi = chunkNext + rank;
          chunkNext += nIdle;
        }
        else
        {
          unsigned nb = 0, ne = 0;
          if(!exhausted)
          {
            const unsigned leader = (unsigned)__ffsll((long long)idleMask) - 1u;
            unsigned base = 0;
            if(lane == leader)
              base = atomicAdd(cursor, WF_CHUNK);
            base = (unsigned)__shfl((int)base, (int)leader);
            if(base >= count)
              exhausted = true;
            else
            {
              nb = base;
              ne = min(base + WF_CHUNK, count);
            }
          }
          if(rank < avail)
            qi = chunkNext + rank;
          else if(rank - avail < ne - nb)
            qi = nb + (rank - avail);
          chunkNext = nb + min(nIdle - avail, ne - nb);
          chunkEnd = ne;
        }
        if(!active && qi != 0xffffffffu)
        {
          pid = queue[qi];
          const float4* r = rec(B, pid);
          const float4 r0 = r[0], r1 = r[1];
          w8_begin(P.sc, S, mk3(r0.x, r0.y, r0.z), mk3(r1.x, r1.y, r1.z), r0.w, anyHit);
          active = S.G.y != 0u;
          if(!active)
            rec(B, pid)[2] = make_float4(S.bestT, 0.0f, 0.0f, __int_as_float(-1));
          nRays++;
        }
      }
    }
    const unsigned long long liveMask = __ballot(active);
    if(liveMask == 0ull)
    {
      if(exhausted && chunkNext >= chunkEnd)
        break;
      continue;
    }
    if(active)
    {
      const bool more = anyHit ? w8_iterate<COUNT, true>(P.sc, S, 0.001f, stk, WF_BLOCK, tc)
                               : w8_iterate<COUNT, false>(P.sc, S, 0.001f, stk, WF_BLOCK, tc);
      if(!more)
      {
        rec(B, pid)[2] = make_float4(S.bestT, S.bestU, S.bestV, __int_as_float(S.bestSlot));
        active = false;
      }
    }
  }
  __shared__ unsigned long long red[VKRT_COUNTER_STRIDE * (WF_BLOCK / 64)];
  const unsigned vals[10] = {anyHit ? 0u : nRays, anyHit ? nRays : 0u, 0, 0, 0, 0, tc.nodes, tc.tris, tc.waveNodeSteps, tc.waveTriSteps};
  blockAddCounters(&P.counters->v[blockIdx.x % VKRT_COUNTER_SLOTS][0], vals, COUNT ? 10 : 2, red);
}

// ---- shade, closest-hit results: rchit / rmiss, then shadow request or accumulation (heavy; few waves/SIMD) ---------
__global__ __launch_bounds__(WF_BLOCK) void k_wf_shade_closest(const TraceParams P, const WfBuffers B, const int round)
{
  const unsigned lane = lane_id();
  const int par = round & 1;
  const unsigned count = B.ctrl[par * 2 + 0];
  if(blockIdx.x * WF_BLOCK >= count)
    return;
  const unsigned qi = blockIdx.x * WF_BLOCK + threadIdx.x;
  __shared__ float lut[512];
  ShadeStats st;
  st.hits = 0; st.diffuse = 0; st.taps = 0;
  st.lut = ldsTexelLut(P.sc, lut);
  __shared__ unsigned wsum[2 * (WF_BLOCK / 64 + 1)];
  bool toClosest = false, toShadow = false;
  unsigned pid = 0;
  if(qi < count)
  {
    pid = qPtr(B, par, 0)[qi];
    LaneState L;
    loadState(P, B, pid, L);
    const float4 h = rec(B, pid)[2], t4 = rec(B, pid)[3];
    RayHit hit;
    hit.t = h.x; hit.u = h.y; hit.v = h.z; hit.slot = __float_as_int(h.w);
    const uint4 ts = make_uint4(__float_as_uint(t4.x), __float_as_uint(t4.y), __float_as_uint(t4.z), __float_as_uint(t4.w));
    if(afterClosestRay(P, L, hit, ts, L.prd.rayDirection, st))
      toShadow = true;
    else
      toClosest = accumulateAndAdvance(P, L, false);
    if(toClosest || toShadow)
    {
      storeState(B, pid, L);
      storeRay(B, pid, L);
    }
  }
  appendQueues(qPtr(B, par ^ 1, 0), &B.ctrl[(par ^ 1) * 2 + 0], qPtr(B, par ^ 1, 1), &B.ctrl[(par ^ 1) * 2 + 1], toClosest, toShadow, pid, lane,
               wsum);
  __shared__ unsigned long long red[VKRT_COUNTER_STRIDE * (WF_BLOCK / 64)];
  const unsigned vals[5] = {0, 0, st.hits, st.diffuse, st.taps};
  blockAddCounters(&P.counters->v[blockIdx.x % VKRT_COUNTER_SLOTS][0], vals, 5, red);
}

// ---- shade, shadow results: accumulate the segment (rgen:99-120), next sample or pixel store (light; many waves) --------
__global__ __launch_bounds__(WF_BLOCK) void k_wf_shade_shadow(const TraceParams P, const WfBuffers B, const int round)
{
  const unsigned lane = lane_id();
  const int par = round & 1;
  const unsigned count = B.ctrl[par * 2 + 1];
  if(blockIdx.x * WF_BLOCK >= count)
    return;
  const unsigned qi = blockIdx.x * WF_BLOCK + threadIdx.x;
  __shared__ unsigned wsum[2 * (WF_BLOCK / 64 + 1)];
  bool toClosest = false;
  unsigned pid = 0;
  if(qi < count)
  {
    pid = qPtr(B, par, 1)[qi];
    LaneState L;
    loadState(P, B, pid, L);
    const bool shadowHit = __float_as_int(rec(B, pid)[2].w) >= 0;
    toClosest = accumulateAndAdvance(P, L, shadowHit);
    if(toClosest)
    {
      storeState(B, pid, L);
      storeRay(B, pid, L);
    }
  }
  appendQueues(qPtr(B, par ^ 1, 0), &B.ctrl[(par ^ 1) * 2 + 0], qPtr(B, par ^ 1, 1), &B.ctrl[(par ^ 1) * 2 + 1], toClosest, false, pid, lane,
               wsum);
}

// ---- host side ------------------------------------------------------------------------------------------------------
size_t vkrt_wf_state_bytes(uint32_t pathCapacity)
{
  return (size_t)pathCapacity * (WF_REC_QUADS * sizeof(float4) + 4 * sizeof(unsigned)) + 256;
}

void vkrt_wf_carve(void* base, uint32_t pathCapacity, WfBuffers* B)
{
  char* p = (char*)base;
  B->ctrl = (unsigned*)p;
  p += 256;
  B->rec = (float4*)p; p += (size_t)pathCapacity * WF_REC_QUADS * sizeof(float4);
  for(int k = 0; k < 4; k++) { B->queue[k] = (unsigned*)p; p += (size_t)pathCapacity * sizeof(unsigned); }
  B->capacity = pathCapacity;
}

hipError_t vkrt_launch_wavefront(const TraceParams& P, const WfBuffers& B, int cuCount, bool count, hipStream_t stream, WfTiming* timing)
{
  const unsigned work = P.tileCount * 64u;
  hipError_t e = hipMemsetAsync(B.ctrl, 0, 64, stream);
  if(e != hipSuccess)
    return e;
  const unsigned blocks = (work + WF_BLOCK - 1) / WF_BLOCK;
  hipLaunchKernelGGL(k_wf_init, dim3(blocks), dim3(WF_BLOCK), 0, stream, P, B);
  if(P.pc.samples <= 0 || P.pc.depth <= 0)
    return hipGetLastError();
  const size_t lds = (size_t)P.sc.stackCap * WF_BLOCK * sizeof(int);
  // closest + shadow entries never exceed the number of paths; +2 blocks for the two partial tails
  const dim3 grid(blocks + 2), bb(WF_BLOCK);
  const bool wide = P.sc.layout == 1u;
  // a path issues at most 2 rays per segment, depth segments per sample, samples per pixel
  const int rounds = 2 * P.pc.samples * P.pc.depth;
  if(timing)
    timing->used = 0;
  static int refill = -1, refillMin = 16, refillBlocks = 0;
  static unsigned travBlock = 64;
  if(refill < 0)
  {
    const char* e = getenv("VKRT_WF_TRAVERSE");
    refill = (e && !strcmp(e, "refill")) ? 1 : 0;
    if((e = getenv("VKRT_WF_REFILL"))) refillMin = std::max(1, std::min(64, atoi(e)));
    if((e = getenv("VKRT_WF_BLOCKS_PER_CU"))) refillBlocks = atoi(e);
    if((e = getenv("VKRT_WF_TRAV_BLOCK"))) travBlock = atoi(e) == 64 ? 64u : atoi(e) == 128 ? 128u : 256u;
  }
  int perCU = 4;
  if(refill && wide)
  {
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, k_wf_traverse_refill<false>, WF_BLOCK, lds);
    if(refillBlocks > 0) perCU = std::min(perCU, refillBlocks);
    perCU = std::max(perCU, 1);
  }
  for(int r = 0; r < rounds; r++)
  {
    const bool timed = timing && timing->events && 2 * (timing->used + 1) <= timing->capacity;
    if(timed)
      (void)hipEventRecord(timing->events[2 * timing->used], stream);
    if(wide && refill)
    {
      (void)hipMemsetAsync(&B.ctrl[4], 0, 8, stream);
      const dim3 pg((unsigned)std::min<long long>((long long)cuCount * perCU, (long long)blocks + 2));
      if(count) hipLaunchKernelGGL(k_wf_traverse_refill<true>, pg, bb, lds, stream, P, B, r, (unsigned)refillMin);
      else hipLaunchKernelGGL(k_wf_traverse_refill<false>, pg, bb, lds, stream, P, B, r, (unsigned)refillMin);
    }
    else
    {
      // one wavefront per workgroup by default: a finished wave frees its slot and LDS without waiting for three others
      const dim3 tg((work + travBlock - 1) / travBlock + 2), tb(travBlock);
      const size_t tlds = (size_t)P.sc.stackCap * travBlock * sizeof(int);
#define VKRT_TRAV_LAUNCH(C, W, TB) hipLaunchKernelGGL((k_wf_traverse<C, W, TB>), tg, tb, tlds, stream, P, B, r)
      if(travBlock == 64)
      {
        if(wide) { if(count) VKRT_TRAV_LAUNCH(true, true, 64); else VKRT_TRAV_LAUNCH(false, true, 64); }
        else { if(count) VKRT_TRAV_LAUNCH(true, false, 64); else VKRT_TRAV_LAUNCH(false, false, 64); }
      }
      else if(travBlock == 128)
      {
        if(wide) { if(count) VKRT_TRAV_LAUNCH(true, true, 128); else VKRT_TRAV_LAUNCH(false, true, 128); }
        else { if(count) VKRT_TRAV_LAUNCH(true, false, 128); else VKRT_TRAV_LAUNCH(false, false, 128); }
      }
      else
      {
        if(wide) { if(count) VKRT_TRAV_LAUNCH(true, true, 256); else VKRT_TRAV_LAUNCH(false, true, 256); }
        else { if(count) VKRT_TRAV_LAUNCH(true, false, 256); else VKRT_TRAV_LAUNCH(false, false, 256); }
      }
#undef VKRT_TRAV_LAUNCH
    }
    if(timed)
    {
      (void)hipEventRecord(timing->events[2 * timing->used + 1], stream);
      timing->used++;
    }
    hipLaunchKernelGGL(k_wf_shade_closest, dim3(blocks), bb, 0, stream, P, B, r);
    hipLaunchKernelGGL(k_wf_shade_shadow, dim3(blocks), bb, 0, stream, P, B, r);
  }
  return hipGetLastError();
}
