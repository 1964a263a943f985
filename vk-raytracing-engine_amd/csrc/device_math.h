// device_math.h -- gfx950 device-side vector math, PRNG and samplers of the path tracer.
//
// Restates reference shaders/random.glsl:6-70 and shaders/globals.glsl:4-5 for HIP.  What GLSL
// leaves implementation-defined is fixed by the "vkrt math profile" (DESIGN.md section 3):
// IEEE binary32, no contraction (-ffp-contract=off), source-order evaluation, explicit fmaf only
// where written, sin/cos by Cody-Waite + minimax polynomials, pow(x,5) by multiplication.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VKRT_DEV __device__ __forceinline__

VKRT_DEV unsigned lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

#define VKRT_COUNTER_STRIDE 16  // u64 per counter slot (10 used; two 64-byte lines)

// Traversal work tallies (only maintained by COUNT instantiations).  waveNodeSteps / waveTriSteps are bumped by one
// lane per wavefront per node step / triangle step, so nodes / (64 * waveNodeSteps) is the lane efficiency of that phase.
struct TravCount
{
  unsigned nodes = 0, tris = 0, waveNodeSteps = 0, waveTriSteps = 0;
};

// Block-reduce up to 10 per-lane counters through LDS and add them to this block's counter slot with
// one atomic per non-zero counter.  `dst` = &counters->v[blockIdx.x % SLOTS][0]; red = VKRT_COUNTER_STRIDE*(blockDim/64) u64 of LDS.
VKRT_DEV void blockAddCounters(unsigned long long* dst, const unsigned* vals, int n, unsigned long long* red)
{
  const unsigned lane = lane_id(), wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for(int k = 0; k < n; k++)
  {
    unsigned long long x = vals[k];
#pragma unroll
    for(int off = 32; off > 0; off >>= 1)
      x += __shfl_xor(x, off);
    if(lane == 0)
      red[wave * VKRT_COUNTER_STRIDE + k] = x;
  }
  __syncthreads();
  if(threadIdx.x < (unsigned)n)
  {
    unsigned long long t = 0;
    for(unsigned w = 0; w < nw; w++)
      t += red[w * VKRT_COUNTER_STRIDE + threadIdx.x];
    if(t != 0ull)
      atomicAdd(&dst[threadIdx.x], t);
  }
}

struct f3 { float x, y, z; };
VKRT_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
VKRT_DEV f3 mk3(float s) { return mk3(s, s, s); }
VKRT_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
VKRT_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
VKRT_DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
VKRT_DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
VKRT_DEV f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
VKRT_DEV f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
VKRT_DEV f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
VKRT_DEV float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
VKRT_DEV f3 cross3(f3 a, f3 b)
{
  return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
VKRT_DEV float length3(f3 a) { return sqrtf(dot3(a, a)); }
VKRT_DEV f3 normalize3(f3 a)
{
  float inv = 1.0f / sqrtf(dot3(a, a));
  return a * inv;
}
// GLSL 4.60 section 8.3: min(x,y) = y<x ? y : x, max(x,y) = x<y ? y : x
VKRT_DEV float glsl_min(float x, float y) { return (y < x) ? y : x; }
VKRT_DEV float glsl_max(float x, float y) { return (x < y) ? y : x; }
VKRT_DEV float glsl_clamp(float x, float lo, float hi) { return glsl_min(glsl_max(x, lo), hi); }
VKRT_DEV f3 glsl_mix(f3 a, f3 b, float t) { return a * (1.0f - t) + b * t; }
VKRT_DEV f3 glsl_reflect(f3 I, f3 N) { return I - (2.0f * dot3(N, I)) * N; }

// fused forms used only by the ray/box/triangle tests (driver-defined in the reference)
VKRT_DEV float fdot3(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
VKRT_DEV f3 fcross3(f3 a, f3 b)
{
  return mk3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}

// globals.glsl:4-5
#define VKRT_PI 3.14159265f
#define VKRT_INV_PI (1.0f / 3.14159265f)

// ---- math profile: sin/cos/pow5 ----------------------------------------------------------------
VKRT_DEV void vk_sincos(float x, float* s_out, float* c_out)
{
  const float FOPI = 1.27323954473516f;
  const float DP1 = 0.78515625f, DP2 = 2.4187564849853515625e-4f, DP3 = 3.77489497744594108e-8f;
  float ax = fabsf(x);
  int j = (int)(ax * FOPI);
  j = (j + 1) & ~1;
  float y = (float)j;
  float r = fmaf(y, -DP1, ax);
  r = fmaf(y, -DP2, r);
  r = fmaf(y, -DP3, r);
  float z = r * r;
  float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
  float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f),
                  z * z, fmaf(-0.5f, z, 1.0f));
  int q = (j >> 1) & 3;
  float s = (q & 1) ? pc : ps;
  float c = (q & 1) ? ps : pc;
  // q: 0 (s,c) 1 (c,-s) 2 (-s,-c) 3 (-c,s)
  if(q == 2 || q == 3) s = -s;
  if(q == 1 || q == 2) c = -c;
  if(x < 0.0f) s = -s;
  *s_out = s;
  *c_out = c;
}
VKRT_DEV float vk_pow5(float x) { float x2 = x * x; return (x2 * x2) * x; }

// ---- random.glsl:6-33 ---------------------------------------------------------------------------
VKRT_DEV uint32_t tea(uint32_t val0, uint32_t val1)
{
  uint32_t v0 = val0, v1 = val1, s0 = 0u;
#pragma unroll
  for(uint32_t n = 0; n < 16u; n++)
  {
    s0 += 0x9e3779b9u;
    v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
    v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
  }
  return v0;
}
VKRT_DEV uint32_t lcg(uint32_t& prev)
{
  prev = 1664525u * prev + 1013904223u;
  return prev & 0x00FFFFFFu;
}
VKRT_DEV float rnd(uint32_t& prev) { return (float)lcg(prev) / (float)0x01000000; }

// random.glsl:35-45
VKRT_DEV f3 samplingHemisphere(uint32_t& seed, f3 x, f3 y, f3 z)
{
  float r1 = rnd(seed);
  float r2 = rnd(seed);
  float sq = sqrtf(r1);
  float sn, cs;
  vk_sincos(2 * VKRT_PI * r2, &sn, &cs);
  f3 direction = mk3(cs * sq, sn * sq, sqrtf(1 - r1));
  direction = direction.x * x + direction.y * y + direction.z * z;
  return direction;
}
// random.glsl:47-54
VKRT_DEV void createCoordinateSystem(f3 N, f3& Nt, f3& Nb)
{
  if(fabsf(N.x) > fabsf(N.y))
    Nt = mk3(N.z, 0, -N.x) / sqrtf(N.x * N.x + N.z * N.z);
  else
    Nt = mk3(0, -N.z, N.y) / sqrtf(N.y * N.y + N.z * N.z);
  Nb = cross3(N, Nt);
}
// random.glsl:56-70
VKRT_DEV f3 samplingNDF_GGXTR(uint32_t& seed, float alpha2)
{
  float r1 = rnd(seed);
  float r2 = rnd(seed);
  float cosTheta = sqrtf((1.0f - r2) / ((alpha2 - 1.0f) * r2 + 1.0f));
  float sinTheta = glsl_clamp(sqrtf(1.0f - cosTheta * cosTheta), 0.0f, 1.0f);
  float phi = r1 * 2.0f * VKRT_PI;
  float sinPhi, cosPhi;
  vk_sincos(phi, &sinPhi, &cosPhi);
  return mk3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
}

// rg16f / rgba16f / r16f stores of the hybrid mode: float -> half (RNE) -> float, integer-exact (same routine as the oracle)
VKRT_DEV float quantizeHalf(float f)
{
  const uint32_t x = __float_as_uint(f);
  const uint32_t sign = x & 0x80000000u;
  uint32_t ax = x & 0x7fffffffu;
  if(ax >= 0x7f800000u)
    return f;
  if(ax < 0x38800000u)
  {
    const uint32_t e = ax >> 23;
    if(e < 101u)
      ax = 0u;
    else
    {
      const uint32_t mant = (ax & 0x7fffffu) | 0x800000u;
      const uint32_t shift = 126u - e;
      uint32_t q = mant >> shift;
      const uint32_t rem = mant & ((1u << shift) - 1u), half = 1u << (shift - 1u);
      if(rem > half || (rem == half && (q & 1u))) q++;
      ax = __float_as_uint((float)q * 5.9604644775390625e-8f);
    }
  }
  else
  {
    const uint32_t rem = ax & 0x1fffu;
    ax &= ~0x1fffu;
    if(rem > 0x1000u || (rem == 0x1000u && (ax & 0x2000u))) ax += 0x2000u;
    if(ax >= 0x47800000u) ax = 0x7f800000u;
  }
  return __uint_as_float(sign | ax);
}
