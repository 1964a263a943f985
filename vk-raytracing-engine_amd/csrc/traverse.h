// traverse.h -- software replacement for traceRayEXT (reference raytrace.rgen:64-75 closest hit,
// :85-97 shadow/any hit): per-lane BVH2 traversal with a per-lane stack in LDS and
// Moeller-Trumbore on pre-transformed 48-byte triangle records.
//
// Result definition (must not depend on the tree): closest hit = smallest t in
// (tmin, tmax), ties -> smallest flattened triangle id; any hit = exists t in (tmin, tmax).
// Box tests are conservative (far side padded) so pruning can never remove the defined result.
#pragma once
#include "device_math.h"
#include "device_scene.h"

struct RayHit
{
  float t, u, v;
  int slot;  // index of the triangle record, -1 = miss
};

VKRT_DEV float safe_inv(float d)
{
  const float tiny = 1e-20f;
  float dd = (fabsf(d) < tiny) ? copysignf(tiny, d) : d;
  return 1.0f / dd;
}

// Margins of the box tests (binary and wide8).  They have to cover the rounding of the slab arithmetic (Ize 2013) AND of the
// triangle test: Moeller-Trumbore accepts a ray that passes an edge on the outside by up to ~3 eps |o - v0| / cos(incidence),
// and a flat axis-aligned triangle has a zero-thickness box, so a margin that only covers the slab arithmetic lets a grazing
// hit depend on the tree.  Measured (r01_experiments.md #42, #44): far side x (1 + 4e-7) only: 1 pixel in 1e5 differs between
// trees; 1e-6 / 2e-6: 1 in 1e6; with the values below (incidence cosines down to ~0.05 covered) none in 2e6 pixels x 2 frames.
#define VKRT_BOX_PAD_ABS 2.0e-5f   // slabs widened by this x max(|t0|, |t1|) per axis
#define VKRT_BOX_PAD_REL 1.00004f  // far side (and the current best t) scaled by this
#define VKRT_BOX_PAD_REL2 1.00008f // wide8: the far side carries the relative margin of both sides -- tn (1 - 4e-5) <= tf (1 + 4e-5) is
                                   // tn <= tf (1 + 8e-5) for the tn > 0 that matter (tn >= tmin); near planes are left unscaled

// conservative slab test against one child box; returns hit and entry distance
VKRT_DEV bool box_test(f3 o, f3 id, float lox, float loy, float loz, float hix, float hiy, float hiz, float tmin, float tmax,
                       float& tnear)
{
  float t0x = (lox - o.x) * id.x, t1x = (hix - o.x) * id.x;
  float t0y = (loy - o.y) * id.y, t1y = (hiy - o.y) * id.y;
  float t0z = (loz - o.z) * id.z, t1z = (hiz - o.z) * id.z;
  // (round 2, tools/fuzz_parity.py: the triangle test's error in t scales with the distance to the triangle in ALL axes and can
  //  carry a hit across the tmin / tmax clamp: every pad also covers the largest origin-to-box distance of any axis, and the clamp
  //  is relaxed by the largest pad; same formula as the oracle's tree walk)
  const float m = fmaxf(fmaxf(fmaxf(fabsf(lox - o.x), fabsf(hix - o.x)), fmaxf(fabsf(loy - o.y), fabsf(hiy - o.y))), fmaxf(fabsf(loz - o.z), fabsf(hiz - o.z)));
  const float cx = m * fminf(fabsf(id.x), 1.0e4f), cy = m * fminf(fabsf(id.y), 1.0e4f), cz = m * fminf(fabsf(id.z), 1.0e4f);
  float px = VKRT_BOX_PAD_ABS * fmaxf(fmaxf(fabsf(t0x), fabsf(t1x)), cx), py = VKRT_BOX_PAD_ABS * fmaxf(fmaxf(fabsf(t0y), fabsf(t1y)), cy),
        pz = VKRT_BOX_PAD_ABS * fmaxf(fmaxf(fabsf(t0z), fabsf(t1z)), cz);
  const float pc = VKRT_BOX_PAD_ABS * fmaxf(cx, fmaxf(cy, cz));
  float tn = fmaxf(fmaxf(fminf(t0x, t1x) - px, fminf(t0y, t1y) - py), fmaxf(fminf(t0z, t1z) - pz, tmin - pc));
  float tf = fminf(fminf(fmaxf(t0x, t1x) + px, fmaxf(t0y, t1y) + py), fminf(fmaxf(t0z, t1z) + pz, tmax + pc));
  tnear = tn;
  return tn <= tf * VKRT_BOX_PAD_REL;
}

// Moeller-Trumbore on (v0,e1,e2); one IEEE division, only for rays inside the triangle.
VKRT_DEV bool tri_test(f3 o, f3 d, f3 v0, f3 e1, f3 e2, float& t, float& u, float& v)
{
  f3 pvec = fcross3(d, e2);
  float det = fdot3(e1, pvec);
  f3 tvec = o - v0;
  float U = fdot3(tvec, pvec);
  f3 qvec = fcross3(tvec, e1);
  float V = fdot3(d, qvec);
  float T = fdot3(e2, qvec);
  bool ok;
  if(det > 0.0f)
    ok = (U >= 0.0f) && (V >= 0.0f) && (U + V <= det);
  else if(det < 0.0f)
    ok = (U <= 0.0f) && (V <= 0.0f) && (U + V >= det);
  else
    ok = false;
  if(!ok)
    return false;
  float inv = 1.0f / det;
  t = T * inv;
  u = U * inv;
  v = V * inv;
  return true;
}

// ---- watertight alternative (VKRT_OPT_WATERTIGHT) -----------------------------------------------------------------------------
// The ray/triangle test of Woop, Benthin, Wald, "Watertight Ray/Triangle Intersection" (JCGT 2013), on records that hold the
// exact vertices (p0, p1, p2): translate to the ray origin, shear so that the ray runs along +z of a permuted frame, evaluate the
// three 2D edge functions; a shared edge gets the same two sheared end points in both triangles, so its edge function is the
// exact negative in the neighbour (products and differences only, NO fused multiply-add here: -ffp-contract=off) and a ray
// cannot pass between them; exact zeros are re-evaluated in double.  This is what the Vulkan specification asks of traceRayEXT
// (raytrace.rgen:64-75).  Same operation order in oracle/oracle.cpp (isect_tri_wt).  No back-face culling (gl_RayFlagsNoneEXT).
// Its reach outside the exact triangle is ~1e-7 of the distance for every triangle shape (vertices near the ray origin are
// exact after the translation), where Moeller-Trumbore's grows with 1 / sin of the corner angle (tri_prep.h).
struct WtRay
{
  int kz;             // dominant axis of the direction; kx = (kz + 1) % 3, ky = (kz + 2) % 3 (no swap: both windings are accepted)
  float Sx, Sy, Sz;   // shear d[kx] / d[kz], d[ky] / d[kz] and scale 1 / d[kz]
};
VKRT_DEV WtRay wt_prepare(f3 d)
{
  WtRay R;
  const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
  R.kz = (ax > ay) ? (ax > az ? 0 : 2) : (ay > az ? 1 : 2);
  const float dz = R.kz == 0 ? d.x : (R.kz == 1 ? d.y : d.z);
  const float dx = R.kz == 0 ? d.y : (R.kz == 1 ? d.z : d.x);
  const float dy = R.kz == 0 ? d.z : (R.kz == 1 ? d.x : d.y);
  R.Sx = dx / dz; R.Sy = dy / dz; R.Sz = 1.0f / dz;
  return R;
}
// (x, y, z) of v in the permuted frame
VKRT_DEV f3 wt_permute(int kz, f3 v) { return kz == 0 ? mk3(v.y, v.z, v.x) : (kz == 1 ? mk3(v.z, v.x, v.y) : v); }
VKRT_DEV bool tri_test_wt(const WtRay& R, f3 o, f3 p0, f3 p1, f3 p2, float& t, float& u, float& v)
{
  const f3 A = wt_permute(R.kz, p0 - o), B = wt_permute(R.kz, p1 - o), C = wt_permute(R.kz, p2 - o);
  const float Ax = A.x - R.Sx * A.z, Ay = A.y - R.Sy * A.z;
  const float Bx = B.x - R.Sx * B.z, By = B.y - R.Sy * B.z;
  const float Cx = C.x - R.Sx * C.z, Cy = C.y - R.Sy * C.z;
  float U = Cx * By - Cy * Bx, V = Ax * Cy - Ay * Cx, W = Bx * Ay - By * Ax;
  if(U == 0.0f || V == 0.0f || W == 0.0f)
  {  // on an edge in single precision: decide it in double (products of two floats are exact there)
    U = (float)((double)Cx * (double)By - (double)Cy * (double)Bx);
    V = (float)((double)Ax * (double)Cy - (double)Ay * (double)Cx);
    W = (float)((double)Bx * (double)Ay - (double)By * (double)Ax);
  }
  if((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f))
    return false;
  const float det = (U + V) + W;
  if(det == 0.0f)
    return false;
  // Distance: from the triangle's plane (unnormalised geometric normal), not the paper's barycentric average of the sheared vertex
  // depths.  With vertices hundreds of units apart that average carries their depth range times the rounding of (U, V, W) -- 0.008
  // in t on a 1320-unit sliver of the round-3 campaign (seed 31004365), enough to pass tmin = 0.001 and re-hit the very surface a
  // bounce ray starts from, a "hit" the box tests rightly prune: tree-dependent.  The plane form is exact to ~eps |p0 - o| / cos
  // and gives 0 for an origin in the plane.  Evaluated on the translated, permuted vertices already at hand: with d' the permuted
  // direction, N.d' = d'z (Nx Sx + Ny Sy + Nz) and 1 / d'z = Sz.  Watertightness is a property of the (U, V, W) decision above
  // and is not touched.
  const f3 N = cross3(B - A, C - A);
  const float den = (N.x * R.Sx + N.y * R.Sy) + N.z;
  if(den == 0.0f)
    return false;
  const float inv = 1.0f / det;
  t = (dot3(N, A) * R.Sz) / den;
  u = V * inv;   // barycentric weights: U -> p0, V -> p1, W -> p2; (u, v) weigh p1 and p2 like Moeller-Trumbore's
  v = W * inv;
  return true;   // (a NaN anywhere leaves t NaN: every caller's `t > tmin` rejects it)
}

// Compile-time "triangle mode" TM of every traversal: bit 0 = the watertight test (VKRT_OPT_WATERTIGHT), bit 1 = the any-hit alpha /
// dissolve stage (VKRT_OPT_ANYHIT_DISSOLVE), bit 2 = the records may carry the stage's flag in their id word although this walk
// treats every triangle as opaque (the ray-cast G-buffer on a scene built for the stage): the id is masked, no hit is ignored.
// TM = 0 is the product default and compiles to exactly the code it was before.
#define VKRT_TM_WATERTIGHT 1
#define VKRT_TM_DISSOLVE 2
#define VKRT_TM_MASKID 4

// per-ray constants + one entry point on a 48-byte record (a, b, c)
template <bool WT> struct TriRay;
template <> struct TriRay<false>
{
  VKRT_DEV void set(f3) {}
  VKRT_DEV bool hit(f3 o, f3 d, float4 a, float4 b, float4 c, float& t, float& u, float& v) const
  {
    return tri_test(o, d, mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x), t, u, v);
  }
};
template <> struct TriRay<true>
{
  WtRay R;
  VKRT_DEV void set(f3 d) { R = wt_prepare(d); }
  VKRT_DEV bool hit(f3 o, f3, float4 a, float4 b, float4 c, float& t, float& u, float& v) const
  {
    return tri_test_wt(R, o, mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x), t, u, v);
  }
};

// ---- any-hit alpha / dissolve stage (reference raytrace_rahit_todo.glsl:23-37; never compiled there, all rays gl_RayFlagsOpaqueEXT) ----
// With VKRT_OPT_ANYHIT_DISSOLVE the builders set bit 31 of the id word of every triangle whose material has dissolve < 1
// (dissolve = pbrBaseColorFactor.a: the glTF stand-in for the OBJ material's `dissolve`, "illum == 4" = dissolve < 1); a candidate hit
// on such a triangle is ignored when dissolve == 0 and otherwise when rnd(tea(gid, seed)) > dissolve, seed = the payload's seed when
// the ray is traced.  (The GLSL draws rnd(prd.seed) per invocation; Vulkan defines neither the order nor the number of any-hit
// invocations, so the decision is made a pure function of (ray, triangle): results stay independent of the tree and of the
// schedule.  prd.seed is not advanced.)  Same function in oracle/oracle.cpp (dissolveIgnores).
template <int TM>
VKRT_DEV int tri_gid(float idWord)  // the flattened triangle id of a record (bit 31 = dissolve flag when the records can carry it)
{
  return (TM & (VKRT_TM_DISSOLVE | VKRT_TM_MASKID)) ? (__float_as_int(idWord) & 0x7fffffff) : __float_as_int(idWord);
}
template <int TM>
VKRT_DEV bool anyhit_ignores(const DevScene& sc, unsigned slot, float idWord, uint32_t raySeed)
{
  if(!(TM & VKRT_TM_DISSOLVE) || __float_as_int(idWord) >= 0)
    return false;  // opaque triangle (or the stage is not compiled in)
  const float alpha = sc.materials[sc.triShade[slot].w].m.pbrBaseColorFactor[3];
  if(alpha == 0.0f)
    return true;
  uint32_t st = tea((uint32_t)(__float_as_int(idWord) & 0x7fffffff), raySeed);
  return rnd(st) > alpha;
}

// Child order of any-hit walks (round 3, profiles/r03_experiments.md #94).  "Is anything in the way?" has the same answer in any order,
// so the order is a pure cost heuristic.  Front to back is right for closest-hit walks; a shadow ray, though, starts on a surface and
// runs to a light: with the reference's eight fallback lights (hello_vulkan.cpp:247-321: one inside the building, seven far
// outside) seven picks out of eight are stopped by the building's own shell -- the LAST thing a front-to-back walk reaches.  Taking
// the farthest pending child first finds that occluder at once: -6.9 % node visits per ray over the whole frame (shadow rays are 45 %
// of the rays), +2.7 % Mrays/s on the bench scene, pixels identical.  VKRT_OPT_WF_SHARE_FLAGS bit 1 forces it for every any-hit walk.
// Bit 2 makes the choice per ray: far-first only when the ray's END POINT (the light, for a shadow ray) lies outside the bounds of the
// scene's geometry -- whatever stops such a ray encloses the rest -- and front to back otherwise (lights among the geometry, AO rays):
// +3.3 % on the bench scene.  The default (option bit 3, resolved at vkrt_accel_build) turns bit 2 on unless the scene has room-sized
// triangles: those sit in the leaves of the top nodes, a front-to-back walk meets them within a step or two, and on the Sponza-like
// tessellation of the same building far-first costs 2 % instead.
VKRT_DEV bool anyhit_far_first(const DevScene& sc, f3 o, f3 d, float tmax)
{
  if(sc.shareFlags & 2u)
    return true;
  if(!(sc.shareFlags & 4u))
    return false;
  const f3 e = mk3(fmaf(d.x, tmax, o.x), fmaf(d.y, tmax, o.y), fmaf(d.z, tmax, o.z));
  return !(e.x >= sc.sceneLo[0] && e.x <= sc.sceneHi[0] && e.y >= sc.sceneLo[1] && e.y <= sc.sceneHi[1] && e.z >= sc.sceneLo[2] && e.z <= sc.sceneHi[2]);
}

// stk: this lane's LDS stack column (entry k at stk[k * stride]).
template <bool COUNT, int TM = 0>
VKRT_DEV void traverse(const DevScene& sc, f3 o, f3 d, float tmin, float tmax, bool anyHit, int* stk, int stride, RayHit& hit,
                       TravCount& tc, uint32_t raySeed = 0u)
{
  const f3 id = mk3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
  const bool farFirst = anyHit && anyhit_far_first(sc, o, d, tmax);
  TriRay<(TM & VKRT_TM_WATERTIGHT) != 0> tr;
  tr.set(d);
  const float4* __restrict__ nodes = sc.nodes;
  const float4* __restrict__ tris = sc.tris;
  const int cap = (int)sc.stackCap;
  float bestT = tmax, bestU = 0.0f, bestV = 0.0f;
  int bestSlot = -1, bestGid = -1;
  int cur = sc.rootRef;
  int sp = 0;
  // Safety net: a well-formed tree needs at most nodes+leaves steps; a malformed one (builder bug)
  // must still terminate so the grid always drains.
  unsigned steps = sc.stepLimit;
  while(cur != VKRT_TRAV_DONE)
  {
    while(cur >= 0)
    {
      if(--steps == 0u)
      {
        VKRT_TRAV_FAULT(sc);
        cur = VKRT_TRAV_DONE;
        break;
      }
      const float4 q0 = nodes[cur * VKRT_NODE_QUADS + 0];
      const float4 q1 = nodes[cur * VKRT_NODE_QUADS + 1];
      const float4 q2 = nodes[cur * VKRT_NODE_QUADS + 2];
      const float4 q3 = nodes[cur * VKRT_NODE_QUADS + 3];
      if(COUNT) tc.nodes++;
      float tn0, tn1;
      const bool h0 = box_test(o, id, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, tmin, bestT, tn0);
      const bool h1 = box_test(o, id, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, tmin, bestT, tn1);
      const int c0 = __float_as_int(q3.x), c1 = __float_as_int(q3.y);
      if(h0 && h1)
      {
        const bool swap = (tn1 < tn0) != farFirst;  // (any-hit walks: far child first, anyhit_far_first)
        const int nearC = swap ? c1 : c0, farC = swap ? c0 : c1;
        if(sp < cap)
        {
          stk[sp * stride] = farC;
          sp++;
        }
        else
          VKRT_TRAV_FAULT(sc);
        cur = nearC;
      }
      else if(h0)
        cur = c0;
      else if(h1)
        cur = c1;
      else
      {
        if(sp == 0)
          cur = VKRT_TRAV_DONE;
        else
        {
          sp--;
          cur = stk[sp * stride];
        }
      }
    }
    if(cur != VKRT_TRAV_DONE)
    {
      if(--steps == 0u)
      {
        VKRT_TRAV_FAULT(sc);
        break;
      }
      const unsigned code = ~(unsigned)cur;
      const unsigned first = code >> 3, cnt = (code & 7u) + 1u;
      bool done = false;
      for(unsigned k = 0; k < cnt; k++)
      {
        const unsigned s = first + k;
        const float4 a = tris[s * VKRT_TRI_QUADS + 0];
        const float4 b = tris[s * VKRT_TRI_QUADS + 1];
        const float4 c = tris[s * VKRT_TRI_QUADS + 2];
        if(COUNT) tc.tris++;
        float t, u, v;
        if(tr.hit(o, d, a, b, c, t, u, v))
        {
          if(t > tmin)
          {
            if(anyHit)
            {
              if(t < tmax && !anyhit_ignores<TM>(sc, s, c.y, raySeed))
              {
                bestSlot = (int)s;
                bestT = t;
                done = true;
                break;
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
        }
      }
      if(done || sp == 0)
        cur = VKRT_TRAV_DONE;
      else
      {
        sp--;
        cur = stk[sp * stride];
      }
    }
  }
  hit.t = bestT; hit.u = bestU; hit.v = bestV; hit.slot = bestSlot;
}
