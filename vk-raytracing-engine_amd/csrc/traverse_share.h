// traverse_share.h -- wide8 traversal of 64 rays by one wave with work sharing between the lanes.
//
// Measured on the bench workload (profiles/r01_experiments.md #17, #25): the traversal kernel is VALU-issue bound and
// its node steps run with 47 % of the lanes busy, because a wave lasts as long as its longest walk while the other
// lanes sit idle.  Handing whole rays to other waves does not pay on gfx950 (#13, #21, #28: the walk loses its
// co-resident neighbours and becomes memory-bound).  Here the idle lanes stay in the wave and take over PART of a
// busy lane's walk instead: a lane with pending node groups on its stack gives the oldest one (the largest, farthest
// subtree) to an idle lane, which continues that ray from there.
//
// A ray's result lives in LDS (`res`, indexed by the ray's home lane) so that every lane working on the ray prunes
// against the same bound and publishes improvements to it
// ballot... (see publish step: one LDS 64-bit atomic minimum on (t, triangle id), then the winner stores its payload).  The result is the same as in traverse_wide.h:
// closest = smallest t in (tmin, tmax), ties -> smallest triangle id; any = exists (order-independent definitions).
#pragma once
#include "traverse_wide.h"

// per-wave LDS block: 64 x (u64 key, slot, u, v, donor lane by rank)
#define VKRT_SHARE_LDS_WORDS 384
struct ShareRes
{
  unsigned long long* key;  // closest: float bits of the best t << 32 | triangle id of the best hit (tie rule); initial tmax << 32 | ~0
  int* slot;                // triangle slot of the best hit, -1 = none
  float* u;
  float* v;
  int* donor;               // scratch of a sharing step: lane of the r-th donor
};
// The lanes of the sharing wave talk to each other through LDS (res.*, donor[]).  They run in lockstep, but the HIP memory
// model does not know that: order every cross-lane write before the reads that depend on it with a wavefront-scope
// release / acquire pair around a wave barrier (no instruction beyond the s_waitcnt the accesses need anyway; it stops
// the compiler from caching or reordering the LDS accesses).
VKRT_DEV void shareSync()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
VKRT_DEV ShareRes shareRes(int* lds320)
{
  ShareRes r;
  r.key = (unsigned long long*)lds320; r.slot = lds320 + 128; r.u = (float*)(lds320 + 192); r.v = (float*)(lds320 + 256); r.donor = lds320 + 320;
  return r;
}

// Must be called by all 64 lanes of a one-wave workgroup (lanes without a ray pass valid = false and only help).
// stk: this lane's stack column (stride 64 entries), res: the wave's ShareRes block.
template <bool COUNT, bool ANYHIT, int TM = 0>
VKRT_DEV void traverse_wide8_share(const DevScene& sc, bool valid, f3 o, f3 d, float tmin, float tmax, uint2* stk, ShareRes res, RayHit& hit,
                                   TravCount& tc, uint32_t raySeed = 0u)
{
  TriRay<(TM & VKRT_TM_WATERTIGHT) != 0> tr;
  tr.set(d);
  const float4* __restrict__ nodes = sc.nodes;
  const float4* __restrict__ tris = sc.tris;
  const int stride = 64;
  const int cap = (int)(sc.stackCap >> 1);
  const int lane = (int)lane_id();
  const unsigned shareMin = sc.shareMinIdle;
  const unsigned long long below = (1ull << lane) - 1ull;
  bool farFirst = ANYHIT && anyhit_far_first(sc, o, d, tmax);  // child order of this lane's ray (traverse.h)

  res.key[lane] = ((unsigned long long)__float_as_uint(tmax) << 32) | 0xffffffffull;
  res.slot[lane] = -1; res.u[lane] = 0.0f; res.v[lane] = 0.0f;
  shareSync();  // adopting lanes read res.*[owner] of other lanes

  int owner = lane;  // home lane of the ray this lane is working on
  f3 id = mk3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
  bool px = !(id.x < 0.0f), py = !(id.y < 0.0f), pz = !(id.z < 0.0f);
  unsigned oct4 = ((px ? 1u : 0u) | (py ? 2u : 0u) | (pz ? 4u : 0u)) * 0x01010101u;  // octinv in every byte (w8_test_children)
  uint2 G = make_uint2(0u, (valid && sc.rootRef != VKRT_TRAV_DONE) ? 0x80000000u : 0u);
  uint2 T = make_uint2(0u, 0u);
  int sp = 0, sb = 0, nPost = 0;  // node groups live in [sb, sp), parked triangle groups in [cap - nPost, cap)
  unsigned steps = sc.stepLimit;
  bool busy = G.y != 0u;

  for(unsigned iter = 0;; iter++)
  {
    const unsigned long long busyMask = __ballot(busy);
    if(busyMask == 0ull)
      break;
    // ---- work sharing (every 4th step): idle lanes adopt the oldest pending node group of busy lanes ---------------------
    if((iter & sc.sharePeriodMask) == sc.sharePeriodMask && (unsigned)__popcll(~busyMask) >= shareMin)
    {
      // donors: lanes with a pending group on their stack give the oldest one (largest, farthest subtree); with shareFlags bit 0
      // a lane whose stack is empty but whose current group still has two or more pending children gives the farthest of them;
      // with shareFlags bit 4 (round 4) a lane that has no node work to give gives TRIANGLES: a parked triangle group whole, or the
      // upper half (by position in the node's triangle block) of two or more pending triangles.  A ray grazing a plane of thin strips
      // comes out of ONE node test with up to 24 triangles to test, one per step, while the rest of the wave has long finished: on
      // Sponza-like drapery half of all steps were such triangle-only steps at 16 % lane efficiency (profiles/r04_experiments.md
      // #113, #116).  (A second matching round, so that lanes giving node work can give triangles in the same step, costs more per
      // step than it saves: -2.7 % on the uniform scene, #116b.)
      const bool giveStack = busy && sp - sb >= 1;
      const bool giveChild = (sc.shareFlags & 1u) != 0u && busy && !giveStack && (G.y & 0xff000000u & ((G.y & 0xff000000u) - 1u)) != 0u;
      const bool mayTris = (sc.shareFlags & 16u) != 0u && busy && !giveStack && !giveChild;
      const bool giveParked = mayTris && nPost > 0;
      const bool giveTris = mayTris && !giveParked && (T.y & (T.y - 1u)) != 0u;
      const bool wants = giveStack || giveChild || giveParked || giveTris;
      const unsigned long long donorMask = __ballot(wants), idleMask = ~busyMask;
      if(donorMask != 0ull)
      {
        const unsigned n = min((unsigned)__popcll(donorMask), (unsigned)__popcll(idleMask));
        const unsigned giveRank = (unsigned)__popcll(donorMask & below), takeRank = (unsigned)__popcll(idleMask & below);
        const bool gives = wants && giveRank < n;
        const bool takes = !busy && takeRank < n;
        uint2 e = make_uint2(0u, 0u);
        if(gives)
        {
          if(giveStack)
          {
            e = stk[sb * stride];
            sb++;
          }
          else if(giveChild)
          {
            const unsigned top = G.y & 0xff000000u;
            // the child this lane would visit LAST: the lowest pending bit in front-to-back order, the highest with farFirst
            const unsigned low = farFirst ? (0x80000000u >> (unsigned)__clz((int)top)) : (top & (0u - top));
            e = make_uint2(G.x, low | (G.y & 0xffu));
            G.y &= ~low;
          }
          else if(giveParked)
          {
            e = stk[(cap - nPost) * stride];
            e.x |= 0x80000000u;  // bit 31 of the base marks the entry as a triangle group
            nPost--;
          }
          else
          {
            // the triangles above the middle of the span of pending positions
            const unsigned lo = (unsigned)__ffs((int)T.y) - 1u, hi = 31u - (unsigned)__clz((int)T.y);
            const unsigned upper = T.y & ~((1u << ((lo + hi + 1u) >> 1)) - 1u);
            e = make_uint2(T.x | 0x80000000u, upper);
            T.y &= ~upper;
          }
          res.donor[giveRank] = lane;  // r-th donor feeds the r-th idle lane
        }
        shareSync();
        const int src = takes ? res.donor[takeRank] : lane;
        const unsigned ex = (unsigned)__shfl((int)e.x, src), ey = (unsigned)__shfl((int)e.y, src);
        const float ox = __shfl(o.x, src), oy = __shfl(o.y, src), oz = __shfl(o.z, src);
        const float dx = __shfl(d.x, src), dy = __shfl(d.y, src), dz = __shfl(d.z, src);
        const float ix = __shfl(id.x, src), iy = __shfl(id.y, src), iz = __shfl(id.z, src);
        const float tm = __shfl(tmax, src);
        const int ow = __shfl(owner, src);
        if(TM & VKRT_TM_DISSOLVE)
        {
          const uint32_t sd = (uint32_t)__shfl((int)raySeed, src);  // the any-hit stage decides per (ray seed, triangle)
          if(takes) raySeed = sd;
        }
        if(takes)
        {
          o = mk3(ox, oy, oz); d = mk3(dx, dy, dz); id = mk3(ix, iy, iz); tmax = tm;
          tr.set(d);  // (watertight: the adopted ray's shear constants, recomputed rather than shuffled)
          farFirst = ANYHIT && anyhit_far_first(sc, o, d, tmax);
          px = !(id.x < 0.0f); py = !(id.y < 0.0f); pz = !(id.z < 0.0f);
          oct4 = ((px ? 1u : 0u) | (py ? 2u : 0u) | (pz ? 4u : 0u)) * 0x01010101u;
          if(ex & 0x80000000u)  // a triangle group of the donor's ray: nothing to walk, only to test
          {
            G = make_uint2(0u, 0u);
            T = make_uint2(ex & 0x7fffffffu, ey);
          }
          else
          {
            G = make_uint2(ex, ey);
            T = make_uint2(0u, 0u);
          }
          sp = 0; sb = 0; nPost = 0;
          owner = ow;
          steps = sc.stepLimit;
          busy = true;
        }
      }
    }
    // ---- one step of every busy lane's walk (same step as w8run) -----------------------------------------------------
    shareSync();  // results published in the previous step are visible to every lane of the ray
    if(busy)
    {
      float bt = tmax;
      int bg = -1;
      bool finished = false;
      bool occluded = false;  // any-hit walks: this lane found an occluder in this step
      if(ANYHIT)
        finished = __hip_atomic_load(&res.slot[owner], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) >= 0;  // somebody already found an occluder for this ray
      else
      {
        const unsigned long long k = __hip_atomic_load(&res.key[owner], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        bt = __uint_as_float((unsigned)(k >> 32));
        bg = (int)(unsigned)k;
      }
      if(!finished)
      {
        // A candidate hit lives only between the tests of one site and its publication right behind them (kept across the whole
        // step, its six registers were re-initialised on every level of the nested control flow: 20 v_mov_b32 per iteration).
        struct Cand
        {
          bool found;
          float t, u, v;
          int slot, gid;
        };
        auto testOne = [&](Cand& cd) {
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
            if(lane == __ffsll((long long)__ballot(1)) - 1) tc.waveTriSteps++;
          }
          float t, u, v;
          if(tr.hit(o, d, a, b, c, t, u, v) && t > tmin)
          {
            if(ANYHIT)
            {
              if(t < tmax && !cd.found && !anyhit_ignores<TM>(sc, s, c.y, raySeed))
              {
                cd.found = true; cd.t = t; cd.slot = (int)s;
              }
            }
            else
            {
              const int gid = tri_gid<TM>(c.y);
              const float rt = cd.found ? cd.t : bt;
              const int rg = cd.found ? cd.gid : bg;
              if((t < rt || (t == rt && gid < rg)) && !anyhit_ignores<TM>(sc, s, c.y, raySeed))
              {
                cd.found = true; cd.t = t; cd.u = u; cd.v = v; cd.slot = (int)s; cd.gid = gid;
              }
            }
          }
        };
        // publish an improvement: LDS atomic minimum on (t, id), the winner leaves its payload.  Called where the candidate was found,
        // by the lanes that found one: the lanes of a site run it in lockstep, and the sequences of two sites never interleave.
        auto publish = [&](const Cand& cd) {
          if(!cd.found)
            return;
          if(ANYHIT)
          {
            __hip_atomic_store(&res.slot[owner], cd.slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            occluded = true;
          }
          else
          {
            // (t bits, triangle id) is unique per candidate, so "the minimum is mine" identifies exactly one winner among the
            // lanes publishing for this ray in this step; it alone writes the payload.  Keys only decrease.
            const unsigned long long mine = ((unsigned long long)__float_as_uint(cd.t) << 32) | (unsigned)cd.gid;
            __hip_atomic_fetch_min(&res.key[owner], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            shareSync();  // every lane's minimum has been applied before anyone checks who won
            if(__hip_atomic_load(&res.key[owner], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) == mine)
            {
              res.slot[owner] = cd.slot; res.u[owner] = cd.u; res.v[owner] = cd.v;
            }
            bt = cd.t; bg = cd.gid;  // (the ray's best is at least this good now)
          }
        };
        // test a whole group out on the spot (rare: a parked group in the way of a node push, or no room to park)
        auto flushT = [&]() {
          Cand cd = {false, 0.0f, 0.0f, 0.0f, -1, 0};
          while(T.y != 0u && !(ANYHIT && cd.found))
            testOne(cd);
          publish(cd);
        };
        if(G.y & 0xff000000u)
        {
          // closest-hit walks go front to back; any-hit walks take the FARTHEST pending child first (farFirst): see anyhit_far_first
          const unsigned bitIdx = farFirst ? (unsigned)__ffs((int)(G.y & 0xff000000u)) - 1u : 31u - (unsigned)__clz((int)G.y);
          const unsigned slot = (bitIdx - 24u) ^ (oct4 & 7u);
          const unsigned child = G.x + (unsigned)__popc(G.y & 0xffu & ((1u << slot) - 1u));
          G.y &= ~(1u << bitIdx);
          if(G.y & 0xff000000u)
          {
            if(__builtin_expect(sp + nPost >= cap && nPost > VKRT_W8_POSTPONE_ROOM, 0))
            {
              // a triangle group parked beyond its reserved room sits where this node group has to go: test it out now
              const uint2 keep = T;
              T = stk[(cap - nPost) * stride];
              nPost--;
              flushT();
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
          w8_test_children<COUNT>(nodes, child, o, id, oct4, px, py, pz, tmin, bt, G, Tn, tc);
          if(Tn.y != 0u)
          {
            if(T.y != 0u)
            {
              if(__builtin_expect(nPost < VKRT_W8_MAX_POSTPONED && sp + nPost < cap, 1))
              {
                nPost++;
                stk[(cap - nPost) * stride] = T;
              }
              else
              {
                if(!occluded)
                  flushT();
                T.y = 0u;
              }
            }
            T = Tn;
          }
        }
        if((G.y & 0xff000000u) == 0u && sp > sb)
        {
          sp--;
          G = stk[sp * stride];
        }
        if(T.y == 0u && nPost > 0)
        {
          T = stk[(cap - nPost) * stride];
          nPost--;
        }
        // triangle step: with triThreshold > 1 the wave tests triangles only when that many of 64 walking lanes hold a pending group
        // (the share of the lanes still busy, so that the thin tail of a wave is not left waiting for a count it cannot reach), or
        // none of them has node work left.  1 = every iteration (experiment #57); default 32: with eight parked groups per lane the
        // triangle step runs at 45 % instead of 29 % lane efficiency for 1.5 % more node visits (profiles/r04_experiments.md #118)
        bool triStep = true;
        if(sc.triThreshold > 1u)
          triStep = (unsigned)__popcll(__ballot(T.y != 0u)) * 64u >= sc.triThreshold * (unsigned)__popcll(busyMask) ||
                    __ballot((G.y & 0xff000000u) != 0u) == 0ull;
        if(T.y != 0u && triStep && !occluded)
        {
          Cand cd = {false, 0.0f, 0.0f, 0.0f, -1, 0};
          testOne(cd);
          publish(cd);
        }
        // (testing two or three triangles per lane in steps where no lane of the wave has node work left -- all loop overhead around one
        //  test -- did not pay: 4110 -> 4082 / 4078 Mrays/s on the strip scene, profiles/r04_experiments.md #116c)
        if(ANYHIT && occluded)
          finished = true;
        else if((G.y & 0xff000000u) == 0u && T.y == 0u && nPost == 0)
          finished = true;  // (sp == sb here: the refill above would have popped otherwise)
        else if(--steps == 0u)
        {
          VKRT_TRAV_FAULT(sc);
          finished = true;
        }
      }
      if(finished)
        busy = false;  // (G, T and the stack indices are dead until the lane adopts new work)
    }
  }
  shareSync();
  hit.t = __uint_as_float((unsigned)(res.key[lane] >> 32)); hit.u = res.u[lane]; hit.v = res.v[lane]; hit.slot = res.slot[lane];
  if(ANYHIT)
    hit.t = tmax;
}
