// wf_streams.h -- record streams of the wavefront pipeline: layout and accessors shared by the two translation units that touch
// them (wavefront.hip: init / shade / blend and the host-side enqueue; wf_traverse.hip: the traversal kernel).
#pragma once
#include <hip/hip_runtime.h>

#include "device_math.h"
#include "device_scene.h"
#include "traverse_wide.h"  // vkrt_v4f

// ---- streams ---------------------------------------------------------------------------------------------------------
// Six streams [parity][type] of float4 planes with room for `capacity` records (storage: plane() below).  type 0 "C": the path's next ray is a closest-hit
// ray; type 1 "S": a shadow ray, and the segment it belongs to is the last of its sample; type 2 "P" (pair): the shadow ray
// of segment k and the closest-hit ray of segment k + 1, both from the hit point of segment k.
//   plane 0  R0  ray origin.xyz, tmax of the first ray (C: 10000; S, P: lightDist - 0.1)      written by the producer
//   plane 1  R1  direction of the first ray (C: closest-hit ray; S, P: shadow ray), S, P: lightDist (read by the hybrid mode only)
//   plane 2  R2  P: direction of the closest-hit ray of the next segment (prd.rayDirection), -
//   plane 3  H0  written by k_wf_traverse.  C: t, u, v, instance id (-1 = miss).  S: .w = 0 occluded / -1 not.
//                P: .x = 1 occluded / 0 not (stored by the shadow lane), .y .z .w = u, v, instance id of the closest-hit ray
//   plane 4  H1  C, P: triShade record of the closest hit
//   plane 5  S0  path weight.xyz (C: curWeight; S, P: weight after the segment the shadow ray belongs to), seed
//   plane 6  S1  hitValue.xyz (radiance of the current sample), flags
//   plane 7  S2  hitValues.xyz (sum over finished samples), px | lrow << 16
//   plane 8  S3  S, P: clamped contribution of the segment if its light is visible (rgen:99-102), -
// flags: depth[0:8) | smpl[8:24) | isSpecular[25]
#define WF_PLANES 9
#define WF_TYPES 3
enum { WF_R0 = 0, WF_R1, WF_R2, WF_H0, WF_H1, WF_S0, WF_S1, WF_S2, WF_S3 };
enum { WF_C = 0, WF_S = 1, WF_P = 2 };

// Storage.  A path is in exactly one stream, so the C and the S records of a round together never outnumber the paths: the two streams
// share their planes -- C records fill a plane from the front, S records from the back (record i of S lives at capacity - 1 - i; a
// wave still reads 1 KB contiguous) -- and neither of them has an R2 field.  17 planes per parity instead of 27: 544 B per path.
#define WF_SLOTS_CS 8  // R0 R1 H0 H1 S0 S1 S2 S3 (C leaves S3 unused, S leaves H1 unused)
#define WF_SLOTS (WF_SLOTS_CS + WF_PLANES)  // + the nine planes of the pair stream
VKRT_DEV float4* plane(const WfBuffers& B, int parity, int type, int k)
{
  const int slot = type == WF_P ? WF_SLOTS_CS + k : (k > WF_R2 ? k - 1 : k);
  return B.planes + ((size_t)(parity * WF_SLOTS + slot)) * B.capacity;
}
VKRT_DEV unsigned* countOf(const WfBuffers& B, int parity, int type) { return &B.ctrl[parity * 4 + type]; }
// record i of a plane: the plane's base is uniform (kernel arguments, round parity, the workgroup's stream type) and the record's byte
// offset fits 32 bits (capacity < 2^28 paths, checked at launch), so the access is "scalar base + 32-bit lane offset": one address
// VGPR per access instead of two (the shade kernel holds a dozen of them at once)
VKRT_DEV float4* rec(const WfBuffers& B, int parity, int type, int k, unsigned i)
{
  return (float4*)((char*)plane(B, parity, type, k) + (size_t)((type == WF_S ? B.capacity - 1u - i : i) * 16u));
}

// Stream records are written by one kernel and consumed by the next one or two (the traversal kernel reads the ray planes, the shade
// kernel the rest; the direction and seed planes are read by both): no later round reads them again, so they are fetched with non-temporal
// loads (the `nt` policy bit) and do not displace tree nodes and triangles from L2 / Infinity Cache on their way out: +1.5 % ray rate on
// the bench scene, +1.8 % on the Sponza-like one.  The stores stay ordinary -- the next kernel reads the records from the caches; written
// non-temporally they come back from HBM and the frame is 4.5 % slower (profiles/r03_experiments.md #103).
VKRT_DEV float4 wfLoad(const float4* p)
{
  const vkrt_v4f v = __builtin_nontemporal_load((const vkrt_v4f*)p);
  return make_float4(v.x, v.y, v.z, v.w);
}
VKRT_DEV void wfStore(float4* p, float4 v) { *p = v; }

