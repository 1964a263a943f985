// wavefront.hip -- the path-trace dispatch as a wavefront pipeline (default execution mode).
// Replaces vkCmdTraceRaysKHR(W,H,1) over raytrace.rgen/.rchit/.rmiss/raytraceShadow.rmiss
// (reference hello_vulkan.cpp:1446) exactly like pathtrace.hip, with the same per-path state machine
// (rgen.h) and therefore the same results; the work is re-scheduled for gfx950:
//
//   k_wf_init      one thread per pixel: rgen prologue (seed, camera ray of sample 0), queue 0.
//   k_wf_traverse  persistent wavefronts eat a ray queue.  A lane that finishes its ray is refilled
//                  from the queue with a wave-aggregated atomic (__ballot + prefix popcount), so a wave
//                  never idles on its slowest ray; ~50 VGPRs, per-lane stacks in LDS.  This is the
//                  software stand-in for traceRayEXT (rgen:64-75 closest hit, :85-97 shadow).
//   k_wf_shade     one thread per finished ray: rchit / rmiss, NEE bookkeeping, accumulation, next
//                  sample or pixel store; surviving paths are appended (ballot compaction) to the next
//                  queue together with their next ray (closest or shadow).
//
// A frame = init + up to samples*depth*2 (traverse, shade) rounds, all enqueued back to back on the
// caller's stream with device-side queue counts (no host synchronisation inside a frame).
// Path state lives in HBM as 16-byte SoA records indexed by path id (coalesced by queue order).
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
#define WF_REFILL_MIN 16u  // refill a wave when at least this many lanes are idle (or all)
#define WF_CHUNK 128u       // queue slots a wave takes per global atomic

// ---- path state in HBM -----------------------------------------------------------------------------------
// S0 (rayOrigin.xyz, lightDist)   S1 (rayDirection.xyz, seed)    S2 (shadowRayDir.xyz, flags)
// S3 (prd.hitValue.xyz, px|lrow)  S4 (prd.weight.xyz, -)         S5 (curWeight.xyz, -)
// S6 (hitValue.xyz, -)            S7 (hitValues.xyz, -)
// R0 (ray origin.xyz, tmax)       R1 (ray dir.xyz, anyHit)       H  (t, u, v, slot)
// flags: depth[0:8) | smpl[8:24) | stage[24] | isSpecular[25]
VKRT_DEV void storeState(const WfBuffers& B, unsigned p, const LaneState& L)
{
  const unsigned flags = (L.prd.depth & 0xffu) | (((unsigned)L.smpl & 0xffffu) << 8) | ((unsigned)L.stage << 24) |
                         ((L.prd.isSpecular ? 1u : 0u) << 25);
  B.S[0][p] = make_float4(L.prd.rayOrigin.x, L.prd.rayOrigin.y, L.prd.rayOrigin.z, L.prd.lightDist);
  B.S[1][p] = make_float4(L.prd.rayDirection.x, L.prd.rayDirection.y, L.prd.rayDirection.z, __uint_as_float(L.prd.seed));
  B.S[2][p] = make_float4(L.prd.shadowRayDir.x, L.prd.shadowRayDir.y, L.prd.shadowRayDir.z, __uint_as_float(flags));
  B.S[3][p] = make_float4(L.prd.hitValue.x, L.prd.hitValue.y, L.prd.hitValue.z, __uint_as_float(L.px | (L.lrow << 16)));
  B.S[4][p] = make_float4(L.prd.weight.x, L.prd.weight.y, L.prd.weight.z, 0.0f);
  B.S[5][p] = make_float4(L.curWeight.x, L.curWeight.y, L.curWeight.z, 0.0f);
  B.S[6][p] = make_float4(L.hitValue.x, L.hitValue.y, L.hitValue.z, 0.0f);
  B.S[7][p] = make_float4(L.hitValues.x, L.hitValues.y, L.hitValues.z, 0.0f);
}

VKRT_DEV void loadState(const TraceParams& P, const WfBuffers& B, unsigned p, LaneState& L)
{
  const float4 s0 = B.S[0][p], s1 = B.S[1][p], s2 = B.S[2][p], s3 = B.S[3][p];
  const float4 s4 = B.S[4][p], s5 = B.S[5][p], s6 = B.S[6][p], s7 = B.S[7][p];
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
  if(L.stage == 1)
  {
    B.R0[p] = make_float4(L.prd.rayOrigin.x, L.prd.rayOrigin.y, L.prd.rayOrigin.z, L.prd.lightDist - 0.1f);
    B.R1[p] = make_float4(L.prd.shadowRayDir.x, L.prd.shadowRayDir.y, L.prd.shadowRayDir.z, __uint_as_float(1u));
  }
  else
  {
    B.R0[p] = make_float4(L.prd.rayOrigin.x, L.prd.rayOrigin.y, L.prd.rayOrigin.z, 10000.0f);
    B.R1[p] = make_float4(L.prd.rayDirection.x, L.prd.rayDirection.y, L.prd.rayDirection.z, __uint_as_float(0u));
  }
}

// Block-aggregated append of `alive` lanes to a queue: ballot + popcount inside each wave, wave totals
// combined through LDS, ONE global atomic per workgroup (the queue count is a single word: per-wave
// atomics would serialise, see WF_CHUNK).  Must be called by every thread of the block; thread order kept.
VKRT_DEV void appendQueue(unsigned* queue, unsigned* count, bool alive, unsigned value, unsigned lane, unsigned* wsum /*LDS [nw+1]*/)
{
  const unsigned wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const unsigned long long m = __ballot(alive);
  if(lane == 0)
    wsum[wave] = (unsigned)__popcll(m);
  __syncthreads();
  if(threadIdx.x == 0)
  {
    unsigned tot = 0;
    for(unsigned w = 0; w < nw; w++)
    {
      const unsigned c = wsum[w];
      wsum[w] = tot;
      tot += c;
    }
    wsum[nw] = tot ? atomicAdd(count, tot) : 0u;
  }
  __syncthreads();
  if(alive)
    queue[wsum[nw] + wsum[wave] + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = value;
  __syncthreads();  // wsum is reused by the next call
}


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
  __shared__ unsigned wsum[WF_BLOCK / 64 + 1];
  appendQueue(B.queue[0], &B.ctrl[0], alive, w, lane, wsum);
  __shared__ unsigned long long red[8 * (WF_BLOCK / 64)];
  const unsigned vals[6] = {0, 0, 0, 0, 0, nPixels};
  blockAddCounters(&P.counters->v[blockIdx.x % VKRT_COUNTER_SLOTS][0], vals, 6, red);
}

// ---- traversal ------------------------------------------------------------------------------------------------
// Conservative slab test in fused form: t = b*(1/d) - o*(1/d).  The fused form has an absolute error
// of about eps*|o/d| on top of the relative one, so the far side is padded relatively AND by errAbs
// (= 2^-22 * max_axis |o/d|, computed once per ray).  Pruning can therefore only visit more, never less,
// and the defined result (smallest t, then smallest triangle id) is unchanged.
VKRT_DEV bool box_test_fma(f3 oid, f3 id, float lox, float loy, float loz, float hix, float hiy, float hiz, float tmin, float tmax,
                           float errAbs, float& tnear)
{
  const float t0x = fmaf(lox, id.x, -oid.x), t1x = fmaf(hix, id.x, -oid.x);
  const float t0y = fmaf(loy, id.y, -oid.y), t1y = fmaf(hiy, id.y, -oid.y);
  const float t0z = fmaf(loz, id.z, -oid.z), t1z = fmaf(hiz, id.z, -oid.z);
  const float tn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), tmin));
  const float tf = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), tmax));
  tnear = tn;
  return tn <= fmaf(tf, 1.0000004f, errAbs);
}

template <bool COUNT>
__global__ __launch_bounds__(WF_BLOCK) void k_wf_traverse(const TraceParams P, const WfBuffers B, const int round, const unsigned refillMin)
{
  extern __shared__ int lds_stack[];
  int* stk = lds_stack + threadIdx.x;
  const unsigned lane = lane_id();
  const unsigned* __restrict__ queue = B.queue[round & 1];
  const unsigned count = B.ctrl[round & 1];
  if(blockIdx.x == 0 && threadIdx.x == 0)
    B.ctrl[(round + 1) & 1] = 0u;  // next queue's count; k_wf_shade of this round appends to it
  if(count == 0u)
    return;
  const float4* __restrict__ nodes = P.sc.nodes;
  const float4* __restrict__ tris = P.sc.tris;
  const int cap = (int)P.sc.stackCap;

  bool active = false, exhausted = false;
  unsigned chunkNext = 0, chunkEnd = 0;  // wave-uniform private slice of the queue
  unsigned pid = 0;
  f3 o = mk3(0.0f), d = mk3(0.0f), id = mk3(0.0f), oid = mk3(0.0f);
  float tmax = 0.0f, errAbs = 0.0f;
  bool anyHit = false;
  float bestT = 0.0f, bestU = 0.0f, bestV = 0.0f;
  int bestSlot = -1, bestGid = -1;
  int cur = VKRT_TRAV_DONE, sp = 0;
  unsigned steps = 0;
  unsigned nClosest = 0, nShadow = 0, nNodes = 0, nTris = 0;
  const float tmin = 0.001f;  // rgen:36

  for(;;)
  {
    // ---- refill idle lanes from the ray queue -----------------------------------------------------------
    // The wave owns a private chunk [chunkNext, chunkEnd) of queue slots; only when it runs dry does one
    // lane pull the next WF_CHUNK slots from the global cursor (one word saturates near 88 dequeues/us,
    // MI355X_MICROARCH.md "dequeue", so per-refill atomics would cap the kernel at ~1.4 Grays/s).
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
              base = atomicAdd(&B.ctrl[2], WF_CHUNK);
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
          const float4 r0 = B.R0[pid], r1 = B.R1[pid];
          o = mk3(r0.x, r0.y, r0.z);
          d = mk3(r1.x, r1.y, r1.z);
          tmax = r0.w;
          anyHit = __float_as_uint(r1.w) != 0u;
          id = mk3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
          oid = mk3(o.x * id.x, o.y * id.y, o.z * id.z);
          errAbs = 2.4e-7f * fmaxf(fabsf(oid.x), fmaxf(fabsf(oid.y), fabsf(oid.z)));
          bestT = tmax; bestU = 0.0f; bestV = 0.0f; bestSlot = -1; bestGid = -1;
          cur = P.sc.rootRef; sp = 0; steps = P.sc.stepLimit;
          active = true;
          if(anyHit) nShadow++; else nClosest++;
        }
      }
    }
    if(__ballot(active) == 0ull)
    {
      if(exhausted && chunkNext >= chunkEnd)
        break;
      continue;
    }

    // ---- inner nodes: descend until every live lane sits on a leaf (or is done) ----------------------------
    while(__ballot(active && cur >= 0) != 0ull)
    {
      if(active && cur >= 0)
      {
        if(--steps == 0u)
          cur = VKRT_TRAV_DONE;  // termination safety net (malformed tree)
        else
        {
          const float4 q0 = nodes[cur * VKRT_NODE_QUADS + 0];
          const float4 q1 = nodes[cur * VKRT_NODE_QUADS + 1];
          const float4 q2 = nodes[cur * VKRT_NODE_QUADS + 2];
          const float4 q3 = nodes[cur * VKRT_NODE_QUADS + 3];
          if(COUNT) nNodes++;
          float tn0, tn1;
          const bool h0 = box_test_fma(oid, id, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, tmin, bestT, errAbs, tn0);
          const bool h1 = box_test_fma(oid, id, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, tmin, bestT, errAbs, tn1);
          const int c0 = __float_as_int(q3.x), c1 = __float_as_int(q3.y);
          if(h0 && h1)
          {
            const bool swap = tn1 < tn0;
            const int nearC = swap ? c1 : c0, farC = swap ? c0 : c1;
            if(sp < cap)
            {
              stk[sp * WF_BLOCK] = farC;
              sp++;
            }
            cur = nearC;
          }
          else if(h0)
            cur = c0;
          else if(h1)
            cur = c1;
          else if(sp == 0)
            cur = VKRT_TRAV_DONE;
          else
          {
            sp--;
            cur = stk[sp * WF_BLOCK];
          }
        }
      }
    }

    // ---- leaves: every live lane now holds a leaf reference or is done -----------------------------------------
    if(active && cur != VKRT_TRAV_DONE)
    {
      const unsigned code = ~(unsigned)cur;
      const unsigned first = code >> 3, cnt = (code & 7u) + 1u;
      bool done = (--steps == 0u);
      for(unsigned k = 0; k < cnt && !done; k++)
      {
        const unsigned s = first + k;
        const float4 a = tris[s * VKRT_TRI_QUADS + 0];
        const float4 b = tris[s * VKRT_TRI_QUADS + 1];
        const float4 c = tris[s * VKRT_TRI_QUADS + 2];
        if(COUNT) nTris++;
        float t, u, v;
        if(tri_test(o, d, mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x), t, u, v))
        {
          if(t > tmin)
          {
            if(anyHit)
            {
              if(t < tmax)
              {
                bestSlot = (int)s;
                bestT = t;
                done = true;
              }
            }
            else
            {
              const int gid = __float_as_int(c.y);
              if(t < bestT || (t == bestT && gid < bestGid))
              {
                bestT = t; bestU = u; bestV = v; bestSlot = (int)s; bestGid = gid;
              }
            }
          }
        }
      }
      if(done || sp == 0)
        cur = VKRT_TRAV_DONE;
      else
      {
        sp--;
        cur = stk[sp * WF_BLOCK];
      }
    }

    // ---- retire finished rays -----------------------------------------------------------------------------------
    if(active && cur == VKRT_TRAV_DONE)
    {
      B.H[pid] = make_float4(bestT, bestU, bestV, __int_as_float(bestSlot));
      active = false;
    }
  }
  __shared__ unsigned long long red[8 * (WF_BLOCK / 64)];
  const unsigned vals[8] = {nClosest, nShadow, 0, 0, 0, 0, nNodes, nTris};
  blockAddCounters(&P.counters->v[blockIdx.x % VKRT_COUNTER_SLOTS][0], vals, COUNT ? 8 : 2, red);
}

// Variant without the persistent refill loop: one thread per queue entry, the megakernel's traverse<>().
template <bool COUNT, bool WIDE>
__global__ __launch_bounds__(WF_BLOCK) void k_wf_traverse_simple(const TraceParams P, const WfBuffers B, const int round)
{
  extern __shared__ int lds_stack[];
  const unsigned* __restrict__ queue = B.queue[round & 1];
  const unsigned count = B.ctrl[round & 1];
  if(blockIdx.x == 0 && threadIdx.x == 0)
    B.ctrl[(round + 1) & 1] = 0u;
  unsigned nClosest = 0, nShadow = 0, nNodes = 0, nTris = 0;
  for(unsigned qi = blockIdx.x * blockDim.x + threadIdx.x; qi < count; qi += gridDim.x * blockDim.x)
  {
    const unsigned pid = queue[qi];
    const float4 r0 = B.R0[pid], r1 = B.R1[pid];
    const bool anyHit = __float_as_uint(r1.w) != 0u;
    RayHit hit;
    traverse_any<COUNT, WIDE>(P.sc, mk3(r0.x, r0.y, r0.z), mk3(r1.x, r1.y, r1.z), 0.001f, r0.w, anyHit, lds_stack, (int)threadIdx.x, WF_BLOCK, hit,
                              nNodes, nTris);
    B.H[pid] = make_float4(hit.t, hit.u, hit.v, __int_as_float(hit.slot));
    if(anyHit) nShadow++; else nClosest++;
  }
  __shared__ unsigned long long red[8 * (WF_BLOCK / 64)];
  const unsigned vals[8] = {nClosest, nShadow, 0, 0, 0, 0, nNodes, nTris};
  blockAddCounters(&P.counters->v[blockIdx.x % VKRT_COUNTER_SLOTS][0], vals, COUNT ? 8 : 2, red);
}

// ---- shade: one thread per finished ray -----------------------------------------------------------------------------
__global__ __launch_bounds__(WF_BLOCK) void k_wf_shade(const TraceParams P, const WfBuffers B, const int round)
{
  const unsigned lane = lane_id();
  const unsigned* __restrict__ queue = B.queue[round & 1];
  unsigned* __restrict__ next = B.queue[(round + 1) & 1];
  const unsigned count = B.ctrl[round & 1];
  if(blockIdx.x == 0 && threadIdx.x == 0)
    B.ctrl[2] = 0u;  // traversal cursor for the next round
  ShadeStats st;
  st.hits = 0; st.diffuse = 0; st.taps = 0;
  __shared__ unsigned wsum[WF_BLOCK / 64 + 1];
  const unsigned stride = gridDim.x * blockDim.x;
  // every thread runs the same number of trips so the barriers inside appendQueue stay converged
  const unsigned trips = (count + stride - 1u) / stride;
  for(unsigned k = 0, q = blockIdx.x * blockDim.x + threadIdx.x; k < trips; k++, q += stride)
  {
    bool alive = false;
    unsigned pid = 0;
    if(q < count)
    {
      pid = queue[q];
      LaneState L;
      loadState(P, B, pid, L);
      const float4 h = B.H[pid];
      RayHit hit;
      hit.t = h.x; hit.u = h.y; hit.v = h.z; hit.slot = __float_as_int(h.w);
      bool needShadow = false, shadowHit = false;
      if(L.stage == 0)
        needShadow = afterClosestRay(P, L, hit, L.prd.rayDirection, st);
      else
        shadowHit = hit.slot >= 0;
      alive = needShadow ? true : accumulateAndAdvance(P, L, shadowHit);
      if(alive)
      {
        storeState(B, pid, L);
        storeRay(B, pid, L);
      }
    }
    appendQueue(next, &B.ctrl[(round + 1) & 1], alive, pid, lane, wsum);
  }
  __shared__ unsigned long long red[8 * (WF_BLOCK / 64)];
  const unsigned vals[5] = {0, 0, st.hits, st.diffuse, st.taps};
  blockAddCounters(&P.counters->v[blockIdx.x % VKRT_COUNTER_SLOTS][0], vals, 5, red);
}

// ---- host side ------------------------------------------------------------------------------------------------------
size_t vkrt_wf_state_bytes(uint32_t pathCapacity)
{
  return (size_t)pathCapacity * (11 * sizeof(float4) + 2 * sizeof(unsigned)) + 256;
}

void vkrt_wf_carve(void* base, uint32_t pathCapacity, WfBuffers* B)
{
  char* p = (char*)base;
  B->ctrl = (unsigned*)p;
  p += 256;
  for(int k = 0; k < 8; k++) { B->S[k] = (float4*)p; p += (size_t)pathCapacity * sizeof(float4); }
  B->R0 = (float4*)p; p += (size_t)pathCapacity * sizeof(float4);
  B->R1 = (float4*)p; p += (size_t)pathCapacity * sizeof(float4);
  B->H = (float4*)p; p += (size_t)pathCapacity * sizeof(float4);
  B->queue[0] = (unsigned*)p; p += (size_t)pathCapacity * sizeof(unsigned);
  B->queue[1] = (unsigned*)p;
  B->capacity = pathCapacity;
}

hipError_t vkrt_launch_wavefront(const TraceParams& P, const WfBuffers& B, int cuCount, bool count, hipStream_t stream, WfTiming* timing)
{
  const unsigned work = P.tileCount * 64u;
  hipError_t e = hipMemsetAsync(B.ctrl, 0, 64, stream);
  if(e != hipSuccess)
    return e;
  hipLaunchKernelGGL(k_wf_init, dim3((work + WF_BLOCK - 1) / WF_BLOCK), dim3(WF_BLOCK), 0, stream, P, B);
  if(P.pc.samples <= 0 || P.pc.depth <= 0)
    return hipGetLastError();
  const size_t lds = (size_t)P.sc.stackCap * WF_BLOCK * sizeof(int);
  int perCU = 0;
  e = count ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, k_wf_traverse<true>, WF_BLOCK, lds)
            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, k_wf_traverse<false>, WF_BLOCK, lds);
  if(e != hipSuccess)
    return e;
  if(perCU < 1)
    return hipErrorInvalidConfiguration;
  static int envRefill = -1, envPerCU = -1, envSimple = 0;
  if(envRefill < 0)
  {
    // Batch-synchronous traversal is the default: lanes of a wave that start together share the top tree
    // levels (coalesced node loads).  VKRT_WF_TRAVERSE=refill selects the per-lane dynamic-refill kernel,
    // measured 4.5x slower on MI355X (desynchronised lanes -> every load touches 64 lines; profiles/).
    const char* es = getenv("VKRT_WF_TRAVERSE");
    envSimple = (es && !strcmp(es, "refill")) ? 0 : 2;
    const char* e = getenv("VKRT_WF_REFILL");
    envRefill = e ? atoi(e) : (int)WF_REFILL_MIN;
    e = getenv("VKRT_WF_BLOCKS_PER_CU");
    envPerCU = e ? atoi(e) : 0;
  }
  if(envPerCU > 0)
    perCU = std::min(perCU, envPerCU);
  const unsigned refillMin = (unsigned)std::max(1, std::min(64, envRefill));
  const unsigned maxBlocks = (work + WF_BLOCK - 1) / WF_BLOCK;
  const unsigned travGrid = std::max(1u, std::min<unsigned>(maxBlocks, (unsigned)(cuCount * perCU)));
  const unsigned shadeGrid = std::max(1u, std::min<unsigned>(maxBlocks, (unsigned)(cuCount * 8)));
  // a path issues at most 2 rays per segment, depth segments per sample, samples per pixel
  const int rounds = 2 * P.pc.samples * P.pc.depth;
  if(timing)
    timing->used = 0;
  for(int r = 0; r < rounds; r++)
  {
    const bool timed = timing && timing->events && 2 * (timing->used + 1) <= timing->capacity;
    if(timed)
      (void)hipEventRecord(timing->events[2 * timing->used], stream);
    const dim3 sg(maxBlocks), bb(WF_BLOCK);
    if(P.sc.layout == 1u)
    {
      if(count)
        hipLaunchKernelGGL((k_wf_traverse_simple<true, true>), sg, bb, lds, stream, P, B, r);
      else
        hipLaunchKernelGGL((k_wf_traverse_simple<false, true>), sg, bb, lds, stream, P, B, r);
    }
    else if(envSimple)
    {
      if(count)
        hipLaunchKernelGGL((k_wf_traverse_simple<true, false>), sg, bb, lds, stream, P, B, r);
      else
        hipLaunchKernelGGL((k_wf_traverse_simple<false, false>), sg, bb, lds, stream, P, B, r);
    }
    else if(count)
      hipLaunchKernelGGL(k_wf_traverse<true>, dim3(travGrid), dim3(WF_BLOCK), lds, stream, P, B, r, refillMin);
    else
      hipLaunchKernelGGL(k_wf_traverse<false>, dim3(travGrid), dim3(WF_BLOCK), lds, stream, P, B, r, refillMin);
    if(timed)
    {
      (void)hipEventRecord(timing->events[2 * timing->used + 1], stream);
      timing->used++;
    }
    hipLaunchKernelGGL(k_wf_shade, dim3(shadeGrid), dim3(WF_BLOCK), 0, stream, P, B, r);
  }
  return hipGetLastError();
}
