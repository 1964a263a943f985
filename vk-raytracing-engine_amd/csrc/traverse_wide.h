// traverse_wide.h -- traversal of the 8-wide compressed BVH (layout: bvh_host.h "wide8").
// Same result definition as traverse.h (smallest t in (tmin,tmax), ties -> smallest triangle id; any hit =
// exists), so images are identical to the BVH2 path; only the work per ray changes: ~10 node visits of
// 5 lane-loads instead of ~31 visits of 4.
//
// Per node: cell size 2^(e-127) and origin are folded with the ray into adj_scale = 2^e/d and
// adj_origin = (origin - o)/d, so each child plane costs one v_cvt_f32_ubyteN + one v_fma_f32.  The fused
// form has an absolute error ~ eps*|adj_origin| per axis; near planes are therefore moved down and far
// planes up by VKRT_BOX_PAD_ABS*|adj_origin| (folded into the per-node constants) and the final comparison carries a
// relative pad, which keeps the box test conservative.  Children are visited front to back by ray octant
// using the slot order the builder prepared (bit 24 + (slot ^ octinv) of the hit mask).
#pragma once
#include "device_math.h"
#include "device_scene.h"
#include "traverse.h"
#include "wide_node.h"

typedef float vkrt_v4f __attribute__((ext_vector_type(4)));
VKRT_DEV float ubyte_f32(unsigned w, int k) { return (float)((w >> (8 * k)) & 0xffu); }

// Resumable per-lane traversal state: one w8_iterate() = take the nearest pending child node of the
// current group, test its 8 children, intersect the triangles the ray's boxes touched, then pop if the
// group is exhausted.  Used as a plain loop (traverse_wide8) and by the refilling kernel (wavefront.hip).
template <int TM>
struct W8State
{
  TriRay<(TM & VKRT_TM_WATERTIGHT) != 0> tr;
  uint32_t raySeed;
  bool farFirst;
  f3 o, d, id;
  float tmax, bestT, bestU, bestV;
  int bestSlot, bestGid;
  uint2 G;
  int sp;
  unsigned steps;
  bool anyHit;
};

template <int TM>
VKRT_DEV void w8_begin(const DevScene& sc, W8State<TM>& S, f3 o, f3 d, float tmax, bool anyHit, uint32_t raySeed)
{
  S.o = o; S.d = d;
  S.tr.set(d);
  S.raySeed = raySeed;
  S.farFirst = anyHit && anyhit_far_first(sc, o, d, tmax);
  S.id = mk3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
  S.tmax = tmax; S.bestT = tmax; S.bestU = 0.0f; S.bestV = 0.0f;
  S.bestSlot = -1; S.bestGid = -1;
  S.G = make_uint2(0u, sc.rootRef == VKRT_TRAV_DONE ? 0u : 0x80000000u);
  S.sp = 0;
  S.steps = sc.stepLimit;
  S.anyHit = anyHit;
}

// byte k of bits4 shifted left by byte k of idx4 (its bits [4:0]): ONE instruction with SDWA byte selects on both operands, where the
// compiler emits v_bfe_u32 + v_lshrrev_b32 + v_lshlrev_b32 -- eight times per node test (profiles/r04_experiments.md #127)
VKRT_DEV unsigned w8_piece(unsigned idx4, unsigned bits4, int k)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__GFX9__)
  return ((bits4 >> (8 * k)) & 0xffu) << ((idx4 >> (8 * k)) & 31u);  // (SDWA is a GFX9 encoding; the library is built for gfx950)
#else
  unsigned r;
  if(k == 0)
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0" : "=v"(r) : "v"(idx4), "v"(bits4));
  else if(k == 1)
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1" : "=v"(r) : "v"(idx4), "v"(bits4));
  else if(k == 2)
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_2" : "=v"(r) : "v"(idx4), "v"(bits4));
  else
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_3" : "=v"(r) : "v"(idx4), "v"(bits4));
  return r;
#endif
}

// Test the 8 children of wide node `child` against the ray (oct4 = the ray's octinv in each of the four bytes, a per-ray constant the
// callers keep: one v_mul_lo_u32 per node test otherwise): G = (child base, hit internal children | imask),
// T = (triangle base, 24-bit mask of the leaf triangles whose boxes the ray touched).
template <bool COUNT>
VKRT_DEV void w8_test_children(const float4* __restrict__ nodes, unsigned child, f3 o, f3 id, unsigned oct4, bool px, bool py, bool pz, float tmin,
                               float bestT_in, uint2& G, uint2& T, TravCount& tc)
{
  const float4* __restrict__ np = nodes + (size_t)child * VKRT_WNODE_QUADS;  // one 64-bit address, immediate offsets
  const float4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3], q4 = np[4];
  if(COUNT)
  {
    tc.nodes++;
    if((int)lane_id() == __ffsll((long long)__ballot(1)) - 1) tc.waveNodeSteps++;
  }
  const unsigned ew = __float_as_uint(q0.w);
  const unsigned imask = ew >> 24;
  const float asx = __uint_as_float((ew & 0xffu) << 23) * id.x;
  const float asy = __uint_as_float(((ew >> 8) & 0xffu) << 23) * id.y;
  const float asz = __uint_as_float(((ew >> 16) & 0xffu) << 23) * id.z;
  const float aox = (q0.x - o.x) * id.x, aoy = (q0.y - o.y) * id.y, aoz = (q0.z - o.z) * id.z;
  // Conservative pads folded into per-node constants (margins: traverse.h): near planes move down and far planes up by an
  // absolute pad; the relative pad of both sides sits on the far planes (scale and offset x (1 + 8e-5), and so the current best t):
  // six multiplies per node test less than scaling the near side by (1 - 4e-5) as well (profiles/r04_experiments.md #126).
  // (the absolute pad scales with the largest |t| a plane of this node can have on the axis, |adj_origin| + QMAX |adj_scale|: a pad
  // relative to |adj_origin| alone vanishes when the node origin shares a coordinate with the ray origin -- a ray leaving a
  // wall along the wall, 1/d ~ 1e6 -- and then a hit 2e-8 outside the slab was pruned, r01_experiments.md #44)
  const float padx = VKRT_BOX_PAD_ABS * fmaf((float)VKRT_WNODE_QMAX, fabsf(asx), fabsf(aox));
  const float pady = VKRT_BOX_PAD_ABS * fmaf((float)VKRT_WNODE_QMAX, fabsf(asy), fabsf(aoy));
  const float padz = VKRT_BOX_PAD_ABS * fmaf((float)VKRT_WNODE_QMAX, fabsf(asz), fabsf(aoz));
  const float nox = aox - padx, fox = (aox + padx) * VKRT_BOX_PAD_REL2;
  const float noy = aoy - pady, foy = (aoy + pady) * VKRT_BOX_PAD_REL2;
  const float noz = aoz - padz, foz = (aoz + padz) * VKRT_BOX_PAD_REL2;
  const float fsx = asx * VKRT_BOX_PAD_REL2, fsy = asy * VKRT_BOX_PAD_REL2, fsz = asz * VKRT_BOX_PAD_REL2;
  const float nsx = asx, nsy = asy, nsz = asz;
  // quantised planes, near/far by ray direction sign
  const unsigned lx0 = __float_as_uint(q2.x), lx1 = __float_as_uint(q2.y), ly0 = __float_as_uint(q2.z), ly1 = __float_as_uint(q2.w);
  const unsigned lz0 = __float_as_uint(q3.x), lz1 = __float_as_uint(q3.y), hx0 = __float_as_uint(q3.z), hx1 = __float_as_uint(q3.w);
  const unsigned hy0 = __float_as_uint(q4.x), hy1 = __float_as_uint(q4.y), hz0 = __float_as_uint(q4.z), hz1 = __float_as_uint(q4.w);
  const unsigned nx[2] = {px ? lx0 : hx0, px ? lx1 : hx1}, fx[2] = {px ? hx0 : lx0, px ? hx1 : lx1};
  const unsigned ny[2] = {py ? ly0 : hy0, py ? ly1 : hy1}, fy[2] = {py ? hy0 : ly0, py ? hy1 : ly1};
  const unsigned nz[2] = {pz ? lz0 : hz0, pz ? lz1 : hz1}, fz[2] = {pz ? hz0 : lz0, pz ? hz1 : lz1};
#define VKRT_WN_PLANE(a, w, k) ubyte_f32((a)[w], k)
  const float bestT = bestT_in;
  unsigned hitmask = 0u;
#pragma unroll
  for(int w = 0; w < 2; w++)
  {
    // four children at a time (one byte each): where the bits of a hit child go in the 32-bit hit mask.
    // internal child: meta = 0x20 | (24 + slot) -> bit 24 + (slot ^ octinv), one bit;
    // leaf child: meta = unary count << 5 | triangle offset -> `count` bits from bit `offset`.
    const unsigned meta4 = __float_as_uint(w == 0 ? q1.z : q1.w);
    const unsigned inner4 = (meta4 & (meta4 << 1)) & 0x10101010u;         // bit 4 of a byte set <=> internal (24..31)
    const unsigned innerMask4 = (inner4 >> 4) * 0x07u;                      // 0x07 per internal byte
    const unsigned bitIndex4 = meta4 ^ (oct4 & innerMask4);  // (a shift reads bits [4:0] of its count: no mask)
    const unsigned bits4 = (meta4 >> 5) & 0x07070707u;
#pragma unroll
    for(int k = 0; k < 4; k++)
    {
      const float tnx = fmaf(VKRT_WN_PLANE(nx, w, k), nsx, nox), tfx = fmaf(VKRT_WN_PLANE(fx, w, k), fsx, fox);
      const float tny = fmaf(VKRT_WN_PLANE(ny, w, k), nsy, noy), tfy = fmaf(VKRT_WN_PLANE(fy, w, k), fsy, foy);
      const float tnz = fmaf(VKRT_WN_PLANE(nz, w, k), nsz, noz), tfz = fmaf(VKRT_WN_PLANE(fz, w, k), fsz, foz);
      const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, tmin));
      const float tf = fminf(fminf(tfx, tfy), fminf(tfz, bestT * VKRT_BOX_PAD_REL2));
      const unsigned piece = w8_piece(bitIndex4, bits4, k);
      hitmask |= (tn <= tf) ? piece : 0u;
    }
  }
#undef VKRT_WN_PLANE
  G = make_uint2(__float_as_uint(q1.x), (hitmask & 0xff000000u) | imask);
  T = make_uint2(__float_as_uint(q1.y), hitmask & 0x00ffffffu);
}

// returns true while the ray has more work
template <bool COUNT, bool ANYHIT, int TM>
VKRT_DEV bool w8_iterate(const DevScene& sc, W8State<TM>& S, float tmin, uint2* stk, int stride, TravCount& tc)
{
  const float4* __restrict__ nodes = sc.nodes;
  const float4* __restrict__ tris = sc.tris;
  const int cap = (int)(sc.stackCap >> 1);
  const f3 o = S.o, d = S.d, id = S.id;
  const bool px = !(id.x < 0.0f), py = !(id.y < 0.0f), pz = !(id.z < 0.0f);
  const unsigned octinv = (px ? 1u : 0u) | (py ? 2u : 0u) | (pz ? 4u : 0u);
  uint2 G = S.G;
  uint2 T = make_uint2(0u, 0u);
  if(G.y & 0xff000000u)
  {
    // take the nearest pending internal child of the group (any-hit walks: the farthest, traverse.h anyhit_far_first)
    const unsigned bitIdx = (ANYHIT && S.farFirst) ? (unsigned)__ffs((int)(G.y & 0xff000000u)) - 1u : 31u - (unsigned)__clz((int)G.y);
    const unsigned slot = (bitIdx - 24u) ^ octinv;
    const unsigned child = G.x + (unsigned)__popc(G.y & 0xffu & ((1u << slot) - 1u));
    G.y &= ~(1u << bitIdx);
    if(G.y & 0xff000000u)
    {
      if(S.sp < cap)
      {
        stk[S.sp * stride] = G;
        S.sp++;
      }
      else
        VKRT_TRAV_FAULT(sc);
    }
    if(--S.steps == 0u)
    {
      VKRT_TRAV_FAULT(sc);
      return false;
    }
    w8_test_children<COUNT>(nodes, child, o, id, octinv * 0x01010101u, px, py, pz, tmin, S.bestT, G, T, tc);
  }
  // triangles of this node that the ray's boxes touched
  while(T.y != 0u)
  {
    const unsigned i = (unsigned)__ffs((int)T.y) - 1u;
    T.y &= T.y - 1u;
    if(--S.steps == 0u)
    {
      VKRT_TRAV_FAULT(sc);
      return false;
    }
    const unsigned s = T.x + i;
    const float4* __restrict__ tp = tris + (size_t)s * VKRT_TRI_QUADS;
    const float4 a = tp[0];
    const float4 b = tp[1];
    const float4 c = tp[2];
    if(COUNT)
    {
      tc.tris++;
      if((int)lane_id() == __ffsll((long long)__ballot(1)) - 1) tc.waveTriSteps++;
    }
    float t, u, v;
    if(S.tr.hit(o, d, a, b, c, t, u, v))
    {
      if(t > tmin)
      {
        if(ANYHIT)
        {
          if(t < S.tmax && !anyhit_ignores<TM>(sc, s, c.y, S.raySeed))
          {
            S.bestSlot = (int)s;
            S.bestT = t;
            return false;
          }
        }
        else
        {
          const int gid = tri_gid<TM>(c.y);
          if((t < S.bestT || (t == S.bestT && gid < S.bestGid)) && !anyhit_ignores<TM>(sc, s, c.y, S.raySeed))
          {
            S.bestT = t; S.bestU = u; S.bestV = v; S.bestSlot = (int)s; S.bestGid = gid;
          }
        }
      }
    }
  }
  if((G.y & 0xff000000u) == 0u)
  {
    if(S.sp == 0)
      return false;
    S.sp--;
    G = stk[S.sp * stride];
  }
  S.G = G;
  return true;
}

// ---- variant with triangle postponing -----------------------------------------------------------------------------
// Measured on the 1080p atrium (profiles/r01_experiments.md #17): with the immediate `while(T.y)` loop above a node
// step runs at ~47 % lane efficiency but a triangle step at ~7 % (0.35 triangles per node visit, so nearly every wave
// step has a few lanes with triangles and everyone else waits).  Here a lane keeps its pending triangle group T next
// to its node group G; the wave tests triangles (one per lane per iteration) only when at least sc.triThreshold of 64 walking
// lanes hold one, or when no lane has node work left.  A second group arriving while T is still pending is parked on the
// top end of the lane's stack column (at most VKRT_W8_MAX_POSTPONED per lane, in whatever room the node stack leaves free beyond the
// VKRT_W8_POSTPONE_ROOM entries reserved for them; a node push that needs the slot tests the newest parked group out; else tested at once).
// The result does not depend on the order (closest = smallest t, ties -> smallest triangle id; any = exists).
template <bool COUNT, bool ANYHIT, int TM>
VKRT_DEV void traverse_wide8_postpone(const DevScene& sc, f3 o, f3 d, float tmin, float tmax, uint2* stk, int stride, RayHit& hit, TravCount& tc,
                                      uint32_t raySeed)
{
  TriRay<(TM & VKRT_TM_WATERTIGHT) != 0> tr;
  tr.set(d);
  const bool farFirst = ANYHIT && anyhit_far_first(sc, o, d, tmax);
  const float4* __restrict__ nodes = sc.nodes;
  const float4* __restrict__ tris = sc.tris;
  const int cap = (int)(sc.stackCap >> 1);
  const unsigned thresh = sc.triThreshold;
  const f3 id = mk3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
  const bool px = !(id.x < 0.0f), py = !(id.y < 0.0f), pz = !(id.z < 0.0f);
  const unsigned octinv = (px ? 1u : 0u) | (py ? 2u : 0u) | (pz ? 4u : 0u);
  float bestT = tmax, bestU = 0.0f, bestV = 0.0f;
  int bestSlot = -1, bestGid = -1;
  uint2 G = make_uint2(0u, sc.rootRef == VKRT_TRAV_DONE ? 0u : 0x80000000u);
  uint2 T = make_uint2(0u, 0u);
  int sp = 0, nPost = 0;
  unsigned steps = sc.stepLimit;

  // one triangle of T; returns true when an any-hit ray is finished
  auto testOne = [&]() -> bool {
    const unsigned i = (unsigned)__ffs((int)T.y) - 1u;
    T.y &= T.y - 1u;
    const unsigned s = T.x + i;
    const float4* __restrict__ tp = tris + (size_t)s * VKRT_TRI_QUADS;
    const float4 a = tp[0];
    const float4 b = tp[1];
    const float4 c = tp[2];
    if(COUNT)
    {
      tc.tris++;
      if((int)lane_id() == __ffsll((long long)__ballot(1)) - 1) tc.waveTriSteps++;
    }
    float t, u, v;
    if(tr.hit(o, d, a, b, c, t, u, v) && t > tmin)
    {
      if(ANYHIT)
      {
        if(t < tmax && !anyhit_ignores<TM>(sc, s, c.y, raySeed))
        {
          bestSlot = (int)s; bestT = t;
          return true;
        }
      }
      else
      {
        const int gid = tri_gid<TM>(c.y);
        if((t < bestT || (t == bestT && gid < bestGid)) && !anyhit_ignores<TM>(sc, s, c.y, raySeed))
        {
          bestT = t; bestU = u; bestV = v; bestSlot = (int)s; bestGid = gid;
        }
      }
    }
    return false;
  };

  bool done = G.y == 0u;
  while(!done)
  {
    if(G.y & 0xff000000u)
    {
      // nearest pending internal child of the group (any-hit walks: the farthest); the rest of the group waits on the stack
      const unsigned bitIdx = farFirst ? (unsigned)__ffs((int)(G.y & 0xff000000u)) - 1u : 31u - (unsigned)__clz((int)G.y);
      const unsigned slot = (bitIdx - 24u) ^ octinv;
      const unsigned child = G.x + (unsigned)__popc(G.y & 0xffu & ((1u << slot) - 1u));
      G.y &= ~(1u << bitIdx);
      if(G.y & 0xff000000u)
      {
        if(__builtin_expect(sp + nPost >= cap && nPost > VKRT_W8_POSTPONE_ROOM, 0))
        {
          // parked groups beyond the reserved entries live in free node-stack room: the node push needs the slot back, so the most
          // recently parked group is tested out now
          const uint2 keep = T;
          T = stk[(cap - nPost) * stride];
          nPost--;
          while(T.y != 0u)
            if(testOne())
            {
              done = true;
              break;
            }
          if(done)
            break;
          T = keep;
        }
        if(sp + nPost < cap)
        {
          stk[sp * stride] = G;
          sp++;
        }
        else
          VKRT_TRAV_FAULT(sc);
      }
      uint2 Tn;
      w8_test_children<COUNT>(nodes, child, o, id, octinv * 0x01010101u, px, py, pz, tmin, bestT, G, Tn, tc);
      if(Tn.y != 0u)
      {
        if(T.y != 0u)
        {
          if(nPost < VKRT_W8_MAX_POSTPONED && sp + nPost < cap)
          {
            nPost++;
            stk[(cap - nPost) * stride] = T;  // parked groups grow down from the top of the lane's column
          }
          else
          {
            while(T.y != 0u)
              if(testOne())
              {
                done = true;
                break;
              }
            if(done)
              break;
          }
        }
        T = Tn;
      }
    }
    // refill G from the node stack and T from the parked groups
    if((G.y & 0xff000000u) == 0u && sp > 0)
    {
      sp--;
      G = stk[sp * stride];
    }
    if(T.y == 0u && nPost > 0)
    {
      T = stk[(cap - nPost) * stride];
      nPost--;
    }
    const bool hasT = T.y != 0u, hasG = (G.y & 0xff000000u) != 0u;
    const unsigned nT = (unsigned)__popcll(__ballot(hasT));
    if(nT * 64u >= thresh * (unsigned)__popcll(__ballot(1)) || __ballot(hasG) == 0ull)  // thresh lanes of 64 still walking
    {
      if(hasT && testOne())
        break;
    }
    if(!hasG && T.y == 0u && nPost == 0)
      break;
    if(--steps == 0u)
    {
      VKRT_TRAV_FAULT(sc);
      break;
    }
  }
  hit.t = bestT; hit.u = bestU; hit.v = bestV; hit.slot = bestSlot;
}

template <bool COUNT, int TM = 0>
VKRT_DEV void traverse_wide8(const DevScene& sc, f3 o, f3 d, float tmin, float tmax, bool anyHit, uint2* stk, int stride, RayHit& hit,
                             TravCount& tc, uint32_t raySeed = 0u)
{
  if(sc.triThreshold != 0u)  // launch-uniform
  {
    if(anyHit)
      traverse_wide8_postpone<COUNT, true, TM>(sc, o, d, tmin, tmax, stk, stride, hit, tc, raySeed);
    else
      traverse_wide8_postpone<COUNT, false, TM>(sc, o, d, tmin, tmax, stk, stride, hit, tc, raySeed);
    return;
  }
  W8State<TM> S;
  w8_begin(sc, S, o, d, tmax, anyHit, raySeed);
  if(S.G.y != 0u)
  {
    if(anyHit)  // workgroup-uniform in the wavefront kernels: two specialised loops, no per-triangle branch
      while(w8_iterate<COUNT, true, TM>(sc, S, tmin, stk, stride, tc))
      {
      }
    else
      while(w8_iterate<COUNT, false, TM>(sc, S, tmin, stk, stride, tc))
      {
      }
  }
  hit.t = S.bestT; hit.u = S.bestU; hit.v = S.bestV; hit.slot = S.bestSlot;
}

// layout dispatch used by the kernels: stkWords = this lane's LDS stack column (4-byte words, stride in words)
template <bool COUNT, bool WIDE, int TM = 0>
VKRT_DEV void traverse_any(const DevScene& sc, f3 o, f3 d, float tmin, float tmax, bool anyHit, int* lds, int tid, int block, RayHit& hit,
                           TravCount& tc, uint32_t raySeed = 0u)
{
  if(WIDE)
    traverse_wide8<COUNT, TM>(sc, o, d, tmin, tmax, anyHit, ((uint2*)lds) + tid, block, hit, tc, raySeed);
  else
    traverse<COUNT, TM>(sc, o, d, tmin, tmax, anyHit, lds + tid, block, hit, tc, raySeed);
}
