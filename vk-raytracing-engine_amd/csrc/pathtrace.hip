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
#include "traverse_wide.h"
#include "rgen.h"
#include "kernels.h"

#define VKRT_BLOCK 256


template <bool COUNT, int MINW, int TM>
__global__ __launch_bounds__(VKRT_BLOCK, MINW) void k_pathtrace(const TraceParams P)
{
  extern __shared__ int lds_stack[];
  int* stk = lds_stack + threadIdx.x;
  const unsigned lane = lane_id();

  LaneState L;
  bool active = false;
  bool exhausted = false;  // wave-uniform: work counter ran out
  unsigned nClosest = 0, nShadow = 0, nPixels = 0;
  TravCount tc;
  __shared__ float lut[512];
  ShadeStats st;
  st.hits = 0; st.diffuse = 0; st.taps = 0;
  st.lut = ldsTexelLut(P.sc, lut);
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
      traverse<COUNT, TM>(P.sc, o, d, tmin, tmax, shadow, stk, VKRT_BLOCK, hit, tc, L.prd.seed);

      bool shadowHit = false;
      bool accumulate = true;
      if(!shadow)
        accumulate = !afterClosestRay(P, L, hit, hit.slot >= 0 ? P.sc.triShade[hit.slot] : make_uint4(0u, 0u, 0u, 0u), d, st);
      else
        shadowHit = hit.slot >= 0;
      if(accumulate)
        active = accumulateAndAdvance(P, L, shadowHit);
    }
  }

  // ---- counters: block reduce, one atomic per counter per block ------------------------------------
  __shared__ unsigned long long red[VKRT_COUNTER_STRIDE * (VKRT_BLOCK / 64)];
  const unsigned vals[10] = {nClosest, nShadow, st.hits, st.diffuse, st.taps, nPixels, tc.nodes, tc.tris, tc.waveNodeSteps, tc.waveTriSteps};
  blockAddCounters(&P.counters->v[blockIdx.x % VKRT_COUNTER_SLOTS][0], vals, COUNT ? 10 : 6, red);
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
  TravCount tc;
  const f3 ro = mk3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), rd = mk3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
  // (test hook: the payload seed of these rays is 0)
  const int tm = (sc.watertight ? VKRT_TM_WATERTIGHT : 0) | (sc.dissolve ? VKRT_TM_DISSOLVE : 0);
#define VKRT_DBG(W, TM) traverse_any<false, W, TM>(sc, ro, rd, tmin, tmax, anyHit != 0, lds_stack, (int)threadIdx.x, VKRT_BLOCK, hit, tc, 0u)
  if(sc.layout == 1u)
  {
    if(tm == 0) VKRT_DBG(true, 0); else if(tm == 1) VKRT_DBG(true, 1); else if(tm == 2) VKRT_DBG(true, 2); else VKRT_DBG(true, 3);
  }
  else
  {
    if(tm == 0) VKRT_DBG(false, 0); else if(tm == 1) VKRT_DBG(false, 1); else if(tm == 2) VKRT_DBG(false, 2); else VKRT_DBG(false, 3);
  }
#undef VKRT_DBG
  if(anyHit)
  {
    gid[i] = hit.slot >= 0 ? 0 : -1;
    t[i] = 0; u[i] = 0; v[i] = 0;
  }
  else
  {
    t[i] = hit.t; u[i] = hit.u; v[i] = hit.v;
    gid[i] = hit.slot >= 0 ? (__float_as_int(sc.tris[hit.slot * VKRT_TRI_QUADS + 2].y) & 0x7fffffff) : -1;
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
// MINW = minimum waves per SIMD the register allocator must allow: 3 measured +34 % over the unconstrained build and
// level with 4 (profiles/r01_experiments.md #2, #3), so the product build carries that one variant.
hipError_t vkrt_launch_pathtrace(const TraceParams& P, unsigned gridBlocks, bool count, hipStream_t stream)
{
  const size_t lds = (size_t)P.sc.stackCap * VKRT_BLOCK * sizeof(int);
  const dim3 g(gridBlocks), b(VKRT_BLOCK);
  const int tm = (P.sc.watertight ? VKRT_TM_WATERTIGHT : 0) | (P.sc.dissolve ? VKRT_TM_DISSOLVE : 0);
#define VKRT_MEGA(TM)                                                                        \
  do {                                                                                       \
    if(count) hipLaunchKernelGGL((k_pathtrace<true, 1, TM>), g, b, lds, stream, P);         \
    else hipLaunchKernelGGL((k_pathtrace<false, 3, TM>), g, b, lds, stream, P);             \
  } while(0)
  switch(tm)
  {
    case 0: VKRT_MEGA(0); break;
    case 1: VKRT_MEGA(1); break;
    case 2: VKRT_MEGA(2); break;
    default: VKRT_MEGA(3); break;
  }
#undef VKRT_MEGA
  return hipGetLastError();
}

int vkrt_pathtrace_block_size() { return VKRT_BLOCK; }

hipError_t vkrt_pathtrace_occupancy(size_t ldsBytes, int* blocksPerCU)
{
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocksPerCU, k_pathtrace<false, 3, 0>, VKRT_BLOCK, ldsBytes);
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
