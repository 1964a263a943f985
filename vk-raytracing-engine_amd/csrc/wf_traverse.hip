// wf_traverse.hip -- the traversal kernel of the wavefront pipeline (wavefront.hip describes the pipeline; kernels.h the launch).
// A translation unit of its own because it wants other compiler flags than the shading kernels (csrc/Makefile): clang's SLP
// vectoriser packs pairs of the triangle test's multiplies and FMAs into v_pk_mul_f32 / v_pk_fma_f32.  A packed FP32 instruction
// issues at half rate on gfx950 -- two of them cost what the four scalar ones do (profiles/r02_issue_microbench.json) -- and the
// operands have to be moved into adjacent registers first: 12 extra v_mov_b32 per triangle step.  Without the vectoriser the
// traversal launch is 1.9 % shorter; the gather-bound shade kernel keeps it (0.6 % faster with it; profiles/r04_experiments.md #126).
#include <hip/hip_runtime.h>

#include "device_math.h"
#include "device_scene.h"
#include "kernels.h"
#include "traverse.h"
#include "traverse_wide.h"
#include "traverse_share.h"
#include "wf_streams.h"

// Ray kinds a traversal workgroup can hold, in dispatch order: heavy closest-hit walks first, any-hit walks behind them.
enum { WF_K_CLOSEST_C = 0, WF_K_CLOSEST_P = 1, WF_K_SHADOW_S = 2, WF_K_SHADOW_P = 3 };

// result of a finished walk -> its record
VKRT_DEV void storeHit(const TraceParams& P, const WfBuffers& B, int par, int kind, unsigned qi, const RayHit& hit)
{
  if(kind == WF_K_SHADOW_S)
    wfStore(rec(B, par, WF_S, WF_H0, qi), make_float4(0.0f, 0.0f, 0.0f, __int_as_float(hit.slot >= 0 ? 0 : -1)));
  else if(kind == WF_K_SHADOW_P)
    ((float*)rec(B, par, WF_P, WF_H0, qi))[0] = __int_as_float(hit.slot >= 0 ? 1 : 0);  // the closest-hit lane of the record owns .yzw
  else
  {
    const int type = kind == WF_K_CLOSEST_C ? WF_C : WF_P;
    int inst = -1;
    if(hit.slot >= 0)
    {
      // first hops of the hit shader's attribute fetch, taken here: the triangle's shading record and its instance id
      // travel with the hit, so the closest-hit shading starts at the vertex / material loads
      const uint4 ts = P.sc.triShade[hit.slot];
      inst = __float_as_int(P.sc.tris[hit.slot * VKRT_TRI_QUADS + 2].z);
      wfStore(rec(B, par, type, WF_H1, qi), make_float4(__uint_as_float(ts.x), __uint_as_float(ts.y), __uint_as_float(ts.z), __uint_as_float(ts.w)));
    }
    float* h = (float*)rec(B, par, type, WF_H0, qi);
    if(type == WF_C)
      wfStore(rec(B, par, WF_C, WF_H0, qi), make_float4(hit.t, hit.u, hit.v, __int_as_float(inst)));
    else
    {
      h[1] = hit.u; h[2] = hit.v; h[3] = __int_as_float(inst);
    }
  }
}

// ---- traversal: one thread per queued ray, workgroups homogeneous in ray kind -----------------------------------
template <bool COUNT, bool WIDE, int TB, int TM = 0>
__global__ __launch_bounds__(TB)
__attribute__((amdgpu_waves_per_eu(TM != 0 && WIDE && TB == 64 ? 5 : 1)))
void k_wf_traverse(const TraceParams P, const WfBuffers B, const int round)
{
  extern __shared__ int lds_stack[];
  const int par = round & 1;
  const unsigned cC = *countOf(B, par, WF_C), cS = *countOf(B, par, WF_S), cP = *countOf(B, par, WF_P);
  if(blockIdx.x == 0 && threadIdx.x == 0)
  {
    for(int t = 0; t < WF_TYPES; t++) *countOf(B, par ^ 1, t) = 0u;  // next round's counts; this round's shade kernel claims slots from them
    // every slot below the counts is traced exactly once: the ray counters of the launch are the stream counts
    if(cC + cP) atomicAdd(&P.counters->v[round % VKRT_COUNTER_SLOTS][0], (unsigned long long)cC + cP);
    if(cS + cP) atomicAdd(&P.counters->v[round % VKRT_COUNTER_SLOTS][1], (unsigned long long)cS + cP);
    if(cP) atomicAdd(&P.counters->v[round % VKRT_COUNTER_SLOTS][11], (unsigned long long)cP);  // pair records (vkrt_counters.pair_records)
  }
  // block ranges: [closest rays of C][closest rays of P][shadow rays of S][shadow rays of P]
  const unsigned nC = (cC + TB - 1) / TB, nP = (cP + TB - 1) / TB, nS = (cS + TB - 1) / TB;
  unsigned blk = blockIdx.x;
  int kind;
  unsigned count;
  if(blk < nC) { kind = WF_K_CLOSEST_C; count = cC; }
  else if((blk -= nC) < nP) { kind = WF_K_CLOSEST_P; count = cP; }
  else if((blk -= nP) < nS) { kind = WF_K_SHADOW_S; count = cS; }
  else if((blk -= nS) < nP) { kind = WF_K_SHADOW_P; count = cP; }
  else return;
  const bool anyHit = kind >= WF_K_SHADOW_S;  // workgroup-uniform
  const int type = kind == WF_K_CLOSEST_C ? WF_C : kind == WF_K_SHADOW_S ? WF_S : WF_P;
  const unsigned qi = blk * TB + threadIdx.x;
  const bool valid = qi < count;
  // the ray of this lane: origin from R0; the shadow ray of a pair record and every C / S ray take R1 and R0.w, the closest-hit
  // ray of a pair record takes R2 and the closest-hit tmax
  float4 r0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), r1 = make_float4(1.0f, 0.0f, 0.0f, 0.0f);
  if(valid)
  {
    r0 = wfLoad(rec(B, par, type, WF_R0, qi));
    r1 = wfLoad(rec(B, par, type, kind == WF_K_CLOSEST_P ? WF_R2 : WF_R1, qi));
    if(kind == WF_K_CLOSEST_P)
      r0.w = 10000.0f;
  }
  // any-hit stage: the payload's seed when the ray is traced (S0.w: after the shading that produced the ray, raytrace.rgen:64-97)
  uint32_t raySeed = 0u;
  if((TM & VKRT_TM_DISSOLVE) && valid)
    raySeed = __float_as_uint(wfLoad(rec(B, par, type, WF_S0, qi)).w);
  TravCount tc;
  __shared__ int shareLds[VKRT_SHARE_LDS_WORDS];
  RayHit hit;
  if(WIDE && TB == 64 && P.sc.shareMinIdle != 0u && P.sc.triThreshold != 0u)  // launch-uniform
  {
    // the whole wave walks together: lanes past the end of the stream have no ray of their own but help
    uint2* stk = ((uint2*)lds_stack) + threadIdx.x;
    if(anyHit)
      traverse_wide8_share<COUNT, true, TM>(P.sc, valid, mk3(r0.x, r0.y, r0.z), mk3(r1.x, r1.y, r1.z), 0.001f, r0.w, stk, shareRes(shareLds), hit, tc, raySeed);
    else
      traverse_wide8_share<COUNT, false, TM>(P.sc, valid, mk3(r0.x, r0.y, r0.z), mk3(r1.x, r1.y, r1.z), 0.001f, r0.w, stk, shareRes(shareLds), hit, tc, raySeed);
    if(valid)
      storeHit(P, B, par, kind, qi, hit);
  }
  else if(valid)
  {
    traverse_any<COUNT, WIDE, TM>(P.sc, mk3(r0.x, r0.y, r0.z), mk3(r1.x, r1.y, r1.z), 0.001f, r0.w, anyHit, lds_stack, (int)threadIdx.x, TB, hit, tc, raySeed);
    storeHit(P, B, par, kind, qi, hit);
  }
  if(COUNT)
  {
    __shared__ unsigned long long red[VKRT_COUNTER_STRIDE * (TB / 64)];
    const unsigned vals[10] = {0, 0, 0, 0, 0, 0, tc.nodes, tc.tris, tc.waveNodeSteps, tc.waveTriSteps};
    blockAddCounters(&P.counters->v[blockIdx.x % VKRT_COUNTER_SLOTS][0], vals, 10, red);
  }
}

// One traversal launch: the instantiation for (instrumented?, node layout, workgroup size, triangle mode).  The non-default triangle
// modes (watertight test, any-hit dissolve stage) are built for the default 64-thread workgroups only (vkrt_accel_build refuses the
// other sizes with them).
void vkrt_wf_launch_traverse(const TraceParams& P, const WfBuffers& B, int r, unsigned travBlock, bool count, dim3 tg, size_t tlds, hipStream_t stream)
{
  const bool wide = P.sc.layout == 1u;
  const int tm = (P.sc.watertight ? VKRT_TM_WATERTIGHT : 0) | (P.sc.dissolve ? VKRT_TM_DISSOLVE : 0);
  const dim3 tb(travBlock);
#define VKRT_TRAV_LAUNCH(C, W, TB, TM) hipLaunchKernelGGL((k_wf_traverse<C, W, TB, TM>), tg, tb, tlds, stream, P, B, r)
#define VKRT_TRAV_MODES(TB, TM)                                                                                                        \
  do {                                                                                                                                 \
    if(wide) { if(count) VKRT_TRAV_LAUNCH(true, true, TB, TM); else VKRT_TRAV_LAUNCH(false, true, TB, TM); }                           \
    else { if(count) VKRT_TRAV_LAUNCH(true, false, TB, TM); else VKRT_TRAV_LAUNCH(false, false, TB, TM); }                             \
  } while(0)
  if(travBlock == 64)
  {
    switch(tm)
    {
      case 0: VKRT_TRAV_MODES(64, 0); break;
      case 1: VKRT_TRAV_MODES(64, 1); break;
      case 2: VKRT_TRAV_MODES(64, 2); break;
      default: VKRT_TRAV_MODES(64, 3); break;
    }
  }
  else if(travBlock == 128)
    VKRT_TRAV_MODES(128, 0);
  else
    VKRT_TRAV_MODES(256, 0);
#undef VKRT_TRAV_MODES
#undef VKRT_TRAV_LAUNCH
}

