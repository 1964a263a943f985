// pathtrace.hip -- the path-trace dispatch: replaces vkCmdTraceRaysKHR(W,H,1) over
// raytrace.rgen / .rchit / .rmiss / raytraceShadow.rmiss (reference hello_vulkan.cpp:1446,
// shaders/raytrace.rgen:24-146).
//
// Execution model (gfx950, wave64): persistent wavefronts.  Every lane owns one pixel at a time and
// runs the rgen sample/bounce loops as a state machine whose unit of work is ONE ray (closest-hit or
// shadow), so all live lanes of a wave are always inside the same BVH traversal loop.  Lanes whose
// pixel is finished are refilled from a global work counter with one wave-aggregated atomic
// (__ballot + popcount prefix); a wave exits when the counter is exhausted and all lanes are idle.
// Pixels are handed out in 8x8 tiles so the 64 pixels a fresh wave pulls are one screen tile.
// Per-lane traversal stacks live in LDS (entry k of lane l at stack[k*BLOCK + l]: bank-conflict free).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "device_math.h"
#include "device_scene.h"
#include "shade.h"
#include "traverse.h"
#include "kernels.h"

#define VKRT_BLOCK 256

VKRT_DEV unsigned lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

struct LaneState
{
  Payload prd;
  f3 curWeight, hitValue, hitValues;
  f3 camOrigin;
  uint32_t px, py;   // global pixel (gl_LaunchIDEXT.xy)
  uint32_t lrow;     // row in the shard-local buffer
  int smpl;
  int stage;         // 0: next ray is the closest-hit ray, 1: next ray is the shadow ray
};

// raytrace.rgen:42-60 -- start sample `smpl` of the lane's pixel
VKRT_DEV void startSample(const TraceParams& P, LaneState& L)
{
  const float r1 = rnd(L.prd.seed);
  const float r2 = rnd(L.prd.seed);
  const float jx = P.pc.frame == 0 ? 0.5f : r1, jy = P.pc.frame == 0 ? 0.5f : r2;
  const float pcx = (float)L.px + jx, pcy = (float)L.py + jy;
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
  L.px = x; L.py = y; L.lrow = lrow;
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
  if(P.pc.frame > 0)
  {
    const float a = 1.0f / (float)(P.pc.frame + 1);
    const float4 old = *dst;
    const f3 m = glsl_mix(mk3(old.x, old.y, old.z), res, a);
    *dst = make_float4(m.x, m.y, m.z, 1.0f);
  }
  else
    *dst = make_float4(res.x, res.y, res.z, 1.0f);
}

// shard-local row -> global row (include/vkrt.h vkrt_shard)
VKRT_DEV uint32_t globalRow(const TraceParams& P, uint32_t lrow)
{
  if(P.stripRows == 0u)
    return lrow;
  const uint32_t s = lrow / P.stripRows, r = lrow % P.stripRows;
  return (s * P.shardCount + P.shardIndex) * P.stripRows + r;
}

template <bool COUNT, int MINW>
__global__ __launch_bounds__(VKRT_BLOCK, MINW) void k_pathtrace(const TraceParams P)
{
  extern __shared__ int lds_stack[];
  int* stk = lds_stack + threadIdx.x;
  const unsigned lane = lane_id();

  LaneState L;
  bool active = false;
  bool exhausted = false;  // wave-uniform: work counter ran out
  unsigned nClosest = 0, nShadow = 0, nPixels = 0, nNodes = 0, nTris = 0;
  ShadeStats st;
  st.hits = 0; st.diffuse = 0; st.taps = 0;
  const uint32_t totalWork = P.tileCount * 64u;

  for(;;)
  {
    // ---- refill idle lanes (wave-aggregated dequeue) ------------------------------------------
    const unsigned long long idleMask = __ballot(!active);
    if(idleMask != 0ull && !exhausted)
    {
      const unsigned nIdle = (unsigned)__popcll(idleMask);
      // refill when the wave is empty or at least a quarter of it idles
      if(nIdle >= 16u || idleMask == ~0ull)
      {
        unsigned base = 0;
        const unsigned leader = (unsigned)__ffsll((long long)idleMask) - 1u;
        if(lane == leader)
          base = atomicAdd(P.workCounter, nIdle);
        base = (unsigned)__shfl((int)base, (int)leader);
        if(base + nIdle >= totalWork)
          exhausted = true;
        if(!active)
        {
          const unsigned w = base + (unsigned)__popcll(idleMask & ((1ull << lane) - 1ull));
          if(w < totalWork)
          {
            const unsigned tile = w >> 6, inTile = w & 63u;
            const uint32_t x = (tile % P.tilesX) * 8u + (inTile & 7u);
            const uint32_t lrow = (tile / P.tilesX) * 8u + (inTile >> 3);
            if(x < P.fullW && lrow < P.localRows)
            {
              const uint32_t y = globalRow(P, lrow);
              if(y < P.fullH)
              {
                startPixel(P, L, x, y, lrow);
                active = true;
                nPixels++;
                if(P.pc.samples <= 0 || P.pc.depth <= 0)
                {  // degenerate launch: no rays, store the resolved (0/samples) value
                  storePixel(P, L);
                  active = false;
                }
              }
            }
          }
        }
      }
    }
    if(__ballot(active) == 0ull)
    {
      if(exhausted)
        break;
      continue;
    }

    // ---- one ray per live lane -------------------------------------------------------------------
    if(active)
    {
      f3 o = L.prd.rayOrigin, d;
      float tmin = 0.001f, tmax;
      const bool shadow = (L.stage == 1);
      if(shadow)
      {
        d = L.prd.shadowRayDir;
        tmax = L.prd.lightDist - 0.1f;  // rgen:94
        nShadow++;
      }
      else
      {
        d = L.prd.rayDirection;
        tmax = 10000.0f;
        nClosest++;
      }
      RayHit hit;
      traverse<COUNT>(P.sc, o, d, tmin, tmax, shadow, stk, VKRT_BLOCK, hit, nNodes, nTris);

      bool shadowHit = false;
      bool accumulate = true;
      if(!shadow)
      {
        if(hit.slot >= 0)
          closestHitShader(P.sc, P.pc, hit, d, L.prd, st);
        else
          missShader(P.pc, L.prd);
        if(!L.prd.isSpecular && L.prd.depth != 100u)  // rgen:79
        {
          L.stage = 1;
          accumulate = false;
        }
      }
      else
      {
        shadowHit = hit.slot >= 0;
        L.stage = 0;
      }
      if(accumulate)
      {
        if(!shadowHit)  // rgen:99-102
        {
          const f3 q = L.prd.hitValue * L.curWeight;
          L.hitValue = L.hitValue + mk3(glsl_min(q.x, 10.0f), glsl_min(q.y, 10.0f), glsl_min(q.z, 10.0f));
        }
        L.curWeight = L.curWeight * L.prd.weight;  // rgen:115
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
            active = false;
          }
        }
      }
    }
  }

  // ---- counters: wave reduce, one atomic per counter per wave -----------------------------------
  unsigned vals[8] = {nClosest, nShadow, st.hits, st.diffuse, st.taps, nPixels, nNodes, nTris};
#pragma unroll
  for(int k = 0; k < 8; k++)
  {
    if(!COUNT && k >= 6)
      break;
    unsigned long long v = vals[k];
#pragma unroll
    for(int off = 32; off > 0; off >>= 1)
      v += __shfl_xor(v, off);
    if(lane == 0 && v != 0ull)
      atomicAdd(&P.counters->v[k], v);
  }
}

// ---- debug / test kernels --------------------------------------------------------------------------
__global__ __launch_bounds__(VKRT_BLOCK) void k_trace_rays(DevScene sc, unsigned n, const float* o, const float* d, float tmin,
                                                           float tmax, int anyHit, float* t, float* u, float* v, int* gid)
{
  extern __shared__ int lds_stack[];
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  RayHit hit;
  unsigned a = 0, b = 0;
  traverse<false>(sc, mk3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), mk3(d[3 * i], d[3 * i + 1], d[3 * i + 2]), tmin, tmax, anyHit != 0,
                  lds_stack + threadIdx.x, VKRT_BLOCK, hit, a, b);
  if(anyHit)
  {
    gid[i] = hit.slot >= 0 ? 0 : -1;
    t[i] = 0; u[i] = 0; v[i] = 0;
  }
  else
  {
    t[i] = hit.t; u[i] = hit.u; v[i] = hit.v;
    gid[i] = hit.slot >= 0 ? __float_as_int(sc.tris[hit.slot * VKRT_TRI_QUADS + 2].y) : -1;
  }
}

__global__ void k_eval_math(int op, unsigned n, const float* a, const float* b, float* out)
{
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  float s, c;
  switch(op)
  {
    case 0: vk_sincos(a[i], &s, &c); out[i] = s; break;
    case 1: vk_sincos(a[i], &s, &c); out[i] = c; break;
    case 2: out[i] = sqrtf(a[i]); break;
    case 3: out[i] = a[i] / b[i]; break;
    case 4: out[i] = vk_pow5(a[i]); break;
    default: out[i] = normalize3(mk3(a[i], b[i], 0.0f)).x; break;
  }
}

// ---- launch wrappers (called from vkrt_api.cpp) ------------------------------------------------------
// MINW = minimum waves per SIMD the register allocator must allow (occupancy knob; variant 0 = default)
static int g_variant = -1;
static int pathtraceVariant()
{
  if(g_variant < 0)
  {
    const char* e = getenv("VKRT_MINWAVES");
    g_variant = e ? atoi(e) : 3;  // 3 waves/SIMD measured +34% over the unconstrained build (profiles/r01_v1_*)
  }
  return g_variant;
}

hipError_t vkrt_launch_pathtrace(const TraceParams& P, unsigned gridBlocks, bool count, hipStream_t stream)
{
  const size_t lds = (size_t)P.sc.stackCap * VKRT_BLOCK * sizeof(int);
  const dim3 g(gridBlocks), b(VKRT_BLOCK);
  if(count)
    hipLaunchKernelGGL((k_pathtrace<true, 1>), g, b, lds, stream, P);
  else
    switch(pathtraceVariant())
    {
      case 3: hipLaunchKernelGGL((k_pathtrace<false, 3>), g, b, lds, stream, P); break;
      case 4: hipLaunchKernelGGL((k_pathtrace<false, 4>), g, b, lds, stream, P); break;
      case 2: hipLaunchKernelGGL((k_pathtrace<false, 2>), g, b, lds, stream, P); break;
      default: hipLaunchKernelGGL((k_pathtrace<false, 1>), g, b, lds, stream, P); break;
    }
  return hipGetLastError();
}

int vkrt_pathtrace_block_size() { return VKRT_BLOCK; }

hipError_t vkrt_pathtrace_occupancy(size_t ldsBytes, int* blocksPerCU)
{
  switch(pathtraceVariant())
  {
    case 3: return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocksPerCU, k_pathtrace<false, 3>, VKRT_BLOCK, ldsBytes);
    case 4: return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocksPerCU, k_pathtrace<false, 4>, VKRT_BLOCK, ldsBytes);
    case 2: return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocksPerCU, k_pathtrace<false, 2>, VKRT_BLOCK, ldsBytes);
    default: return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocksPerCU, k_pathtrace<false, 1>, VKRT_BLOCK, ldsBytes);
  }
}

hipError_t vkrt_launch_trace_rays(const DevScene& sc, unsigned n, const float* o, const float* d, float tmin, float tmax, int anyHit,
                                  float* t, float* u, float* v, int* gid, hipStream_t stream)
{
  const size_t lds = (size_t)sc.stackCap * VKRT_BLOCK * sizeof(int);
  hipLaunchKernelGGL(k_trace_rays, dim3((n + VKRT_BLOCK - 1) / VKRT_BLOCK), dim3(VKRT_BLOCK), lds, stream, sc, n, o, d, tmin, tmax,
                     anyHit, t, u, v, gid);
  return hipGetLastError();
}

hipError_t vkrt_launch_eval_math(int op, unsigned n, const float* a, const float* b, float* out, hipStream_t stream)
{
  hipLaunchKernelGGL(k_eval_math, dim3((n + 255) / 256), dim3(256), 0, stream, op, n, a, b, out);
  return hipGetLastError();
}
