// shade.h -- HIP restatement of the reference hit-group semantics:
//   raytrace.rchit:31-219 (closest hit), raytrace.rmiss:11-19 (miss), gltf.glsl:26-154 (PBR BRDF,
//   point-light NEE) and the texture() fetches they issue (bilinear, REPEAT, LOD 0:
//   hello_vulkan.cpp:448-454).  Quirks kept on purpose:
//   ffnormal unused, ratio from unclamped metalness, emission only at depth 0 or after a specular
//   bounce, directLight re-reads unclamped material, weight may be negative.
#pragma once
#include "device_math.h"
#include "device_scene.h"
#include "traverse.h"

// raycommon.glsl:8-19
struct Payload
{
  f3 hitValue;
  uint32_t seed;
  uint32_t depth;
  f3 rayOrigin, rayDirection, weight;
  bool isSpecular;
  float lightDist;
  f3 shadowRayDir;
};

struct ShadeStats
{
  unsigned hits, diffuse, taps;
  const float* lut;  // 512-entry texel decode table (DevScene::srgbLut or the workgroup's LDS copy, see ldsTexelLut)
};

// Copy the 2-KB texel decode table into LDS (call from every thread of the block, before any divergence).
VKRT_DEV const float* ldsTexelLut(const DevScene& sc, float* lds512)
{
  for(unsigned i = threadIdx.x; i < 512u; i += blockDim.x)
    lds512[i] = sc.srgbLut[i];
  __syncthreads();
  return lds512;
}

struct f4 { float x, y, z, w; };

// Gathers of the hit shader go through buffer descriptors (four SGPRs built from a kernel-argument pointer) with a 32-bit byte
// offset per lane instead of a 64-bit flat address per lane: one VGPR per address instead of two.  The shader holds 16 texel
// addresses + 8 record addresses at its register peak (profiles/r04_experiments.md #114).  Every table is < 4 GiB (vkrt_scene_create refuses larger ones);
// the range check of the descriptor is left open (all ones): indices are validated at upload, as before.
typedef unsigned vkrt_v4u __attribute__((ext_vector_type(4)));
struct BufView { __amdgpu_buffer_rsrc_t r; };
VKRT_DEV BufView bufView(const void* p) { return BufView{__builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)0xffffffffu, 0x00020000)}; }
VKRT_DEV float4 bufLoad4(BufView b, uint32_t byteOffset)
{
  const vkrt_v4u v = __builtin_amdgcn_raw_buffer_load_b128(b.r, (int)byteOffset, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
VKRT_DEV uint32_t bufLoad1(BufView b, uint32_t byteOffset) { return __builtin_amdgcn_raw_buffer_load_b32(b.r, (int)byteOffset, 0, 0); }

// i mod n in [0, n) for n > 0 (REPEAT addressing); power-of-two sizes take the mask path
VKRT_DEV int wrapi(int i, int n)
{
  if((n & (n - 1)) == 0)
    return i & (n - 1);
  int m = i % n;
  return m < 0 ? m + n : m;
}

// texture(): bilinear, REPEAT, LOD 0, RGBA8 UNORM / sRGB (hello_vulkan.cpp:448-454), split in three steps so a hit can
// issue the texel loads of all its textures back to back: footprint (addresses + weights), 4 loads, decode + blend.
struct TexTap
{
  uint32_t i00, i10, i01, i11;  // texel indices into the pool
  float ax, ay;
  uint32_t lutBase;             // 0: sRGB decode, 256: UNORM decode (rgb; alpha is always UNORM)
  bool white;                   // index out of range: 1x1 white dummy (hello_vulkan.cpp:468-472)
};
VKRT_DEV void texFootprint(uint32_t offset, uint32_t width, uint32_t height, bool srgb, bool valid, bool want, float u, float v, TexTap& t)
{
  float fx = u * (float)width - 0.5f;
  float fy = v * (float)height - 0.5f;
  if(!(fabsf(fx) < 1.0e9f)) fx = 0.0f;
  if(!(fabsf(fy) < 1.0e9f)) fy = 0.0f;
  const float flx = floorf(fx), fly = floorf(fy);
  t.ax = fx - flx; t.ay = fy - fly;
  const int w = (int)width, h = (int)height;
  const int x0 = wrapi((int)flx, w), y0 = wrapi((int)fly, h);
  const int x1 = x0 + 1 == w ? 0 : x0 + 1, y1 = y0 + 1 == h ? 0 : y0 + 1;
  const bool live = want && valid;
  const uint32_t r0 = offset + (uint32_t)y0 * width, r1 = offset + (uint32_t)y1 * width;
  t.i00 = live ? r0 + (uint32_t)x0 : 0u; t.i10 = live ? r0 + (uint32_t)x1 : 0u;
  t.i01 = live ? r1 + (uint32_t)x0 : 0u; t.i11 = live ? r1 + (uint32_t)x1 : 0u;
  t.lutBase = srgb ? 0u : 256u;
  t.white = !valid;
}
VKRT_DEV f4 texelDecode(const float* lut, uint32_t p, uint32_t base)
{
  f4 o;
  o.x = lut[base + (p & 255u)]; o.y = lut[base + ((p >> 8) & 255u)]; o.z = lut[base + ((p >> 16) & 255u)];
  o.w = lut[256u + (p >> 24)];
  return o;
}
VKRT_DEV f4 texBlend(const float* lut, const TexTap& tp, uint32_t p00, uint32_t p10, uint32_t p01, uint32_t p11)
{
  f4 r;
  if(tp.white)
  {
    r.x = r.y = r.z = r.w = 1.0f;
    return r;
  }
  const f4 t00 = texelDecode(lut, p00, tp.lutBase), t10 = texelDecode(lut, p10, tp.lutBase);
  const f4 t01 = texelDecode(lut, p01, tp.lutBase), t11 = texelDecode(lut, p11, tp.lutBase);
  const float ax = tp.ax, ay = tp.ay, bx = 1.0f - ax, by = 1.0f - ay;
  r.x = (t00.x * bx + t10.x * ax) * by + (t01.x * bx + t11.x * ax) * ay;
  r.y = (t00.y * bx + t10.y * ax) * by + (t01.y * bx + t11.y * ax) * ay;
  r.z = (t00.z * bx + t10.z * ax) * by + (t01.z * bx + t11.z * ax) * ay;
  r.w = (t00.w * bx + t10.w * ax) * by + (t01.w * bx + t11.w * ax) * ay;
  return r;
}
// one texture by index through the descriptor table at LOD 0 (G-buffer path; the path tracer's hit shader uses DevTexRef)
VKRT_DEV f4 sampleTexLevel(const DevScene& sc, const DevTexture& tx, const uint32_t* levelOffsets, uint32_t level, float u, float v, const float* lut)
{
  const uint32_t ws = tx.width >> level, hs = tx.height >> level;
  const uint32_t w = ws ? ws : 1u, h = hs ? hs : 1u;
  TexTap tp;
  texFootprint(level == 0u ? tx.offset : levelOffsets[level], w, h, (tx.srgb & 1u) != 0u, true, true, u, v, tp);
  return texBlend(lut, tp, sc.texels[tp.i00], sc.texels[tp.i10], sc.texels[tp.i01], sc.texels[tp.i11]);
}

// log2 for the LOD computation, the same operation sequence on both sides (exponent from the bits, 2 atanh((m-1)/(m+1)) series
// on m in (sqrt(1/2), sqrt(2)]): relative error < 1e-7, well inside what the Vulkan spec allows an implementation for lambda.
VKRT_DEV float lodLog2(float x)
{
  const uint32_t b = (uint32_t)__float_as_int(x);
  int e = (int)((b >> 23) & 255u) - 127;
  float m = __int_as_float((int)((b & 0x007fffffu) | 0x3f800000u));
  if(m > 1.41421356f)
  {
    m = m * 0.5f;
    e += 1;
  }
  const float s = (m - 1.0f) / (m + 1.0f), s2 = s * s;
  const float series = 1.0f + s2 * (0.333333333f + s2 * (0.2f + s2 * (0.142857143f + s2 * 0.111111111f)));
  return (float)e + (2.0f * s * series) * 1.44269504f;
}

// Screen-space derivatives of the texture coordinates of one G-buffer pixel (what dFdx / dFdy of the interpolated
// fragTexCoord give the fragment shader's implicit-LOD texture(), frag_shader.frag:96-125).
struct TexGrad
{
  float dudx, dvdx, dudy, dvdy;
  bool on;  // false: LOD 0 (VKRT_OPT_GBUFFER_MIPS = 0)
};

// texture() with implicit derivatives as the sampler of hello_vulkan.cpp:448-454 defines it: linear min / mag filter, linear
// mip filter over the full chain, anisotropy up to 4.  Vulkan 1.3 spec chapter 16 "Image Operations": scale factors rho_x,
// rho_y from the derivatives in texel units, eta = min(rho_max / rho_min, maxAniso), N = ceil(eta), lambda = log2(rho_max /
// eta) clamped to the chain, N bilinear taps along the major axis at u(x - 1/2 + i / (N + 1)), each blended between the two
// levels around lambda.
VKRT_DEV f4 sampleTex(const DevScene& sc, int texIndex, float u, float v, const TexGrad& g, ShadeStats& st)
{
  st.taps++;
  f4 r;
  if(sc.textureCount == 0u || texIndex < 0 || texIndex >= (int)sc.textureCount)
  {
    r.x = r.y = r.z = r.w = 1.0f;
    return r;
  }
  DevTexture tx = sc.textures[texIndex];
  const uint32_t levels = tx.srgb >> 8;
  if(!g.on || levels <= 1u)
    return sampleTexLevel(sc, tx, nullptr, 0u, u, v, st.lut);
  const uint32_t* levelOffsets = sc.texMips + (size_t)texIndex * VKRT_MAX_MIPS;
  const float fw = (float)tx.width, fh = (float)tx.height;
  const float mxu = g.dudx * fw, mxv = g.dvdx * fh, myu = g.dudy * fw, myv = g.dvdy * fh;
  const float rx2 = mxu * mxu + mxv * mxv, ry2 = myu * myu + myv * myv;
  const bool majorX = rx2 >= ry2;
  const float rmax = sqrtf(majorX ? rx2 : ry2), rmin = sqrtf(majorX ? ry2 : rx2);
  float eta = 1.0f;
  if(rmin > 0.0f)
    eta = glsl_min(rmax / rmin, 4.0f);
  else if(rmax > 0.0f)
    eta = 4.0f;
  if(!(eta >= 1.0f))
    eta = 1.0f;
  const int N = (int)ceilf(eta);
  const float scale = rmax / eta;
  float lambda = 0.0f;
  if(scale > 1.0f && scale < 3.0e38f)
    lambda = lodLog2(scale);
  else if(!(scale <= 1.0f))
    lambda = (float)(levels - 1u);  // infinite or NaN footprint: the smallest level
  lambda = glsl_clamp(lambda, 0.0f, (float)(levels - 1u));
  const float fl = floorf(lambda), delta = lambda - fl;
  const uint32_t hi = (uint32_t)fl, lo = hi + 1u < levels ? hi + 1u : levels - 1u;
  const float du = majorX ? g.dudx : g.dudy, dv = majorX ? g.dvdx : g.dvdy;
  r.x = r.y = r.z = r.w = 0.0f;
  for(int i = 1; i <= N; i++)
  {
    const float o = (float)i / (float)(N + 1) - 0.5f;
    float tu = u + o * du, tv = v + o * dv;
    if(!(fabsf(tu) < 1.0e9f) || !(fabsf(tv) < 1.0e9f))
    {
      tu = u;
      tv = v;
    }
    const f4 a = sampleTexLevel(sc, tx, levelOffsets, hi, tu, tv, st.lut), b = sampleTexLevel(sc, tx, levelOffsets, lo, tu, tv, st.lut);
    r.x = r.x + (a.x * (1.0f - delta) + b.x * delta);
    r.y = r.y + (a.y * (1.0f - delta) + b.y * delta);
    r.z = r.z + (a.z * (1.0f - delta) + b.z * delta);
    r.w = r.w + (a.w * (1.0f - delta) + b.w * delta);
  }
  const float inv = 1.0f / (float)N;
  r.x = r.x * inv; r.y = r.y * inv; r.z = r.z * inv; r.w = r.w * inv;
  return r;
}

// gltf.glsl:26-32
VKRT_DEV f3 pbrGetBaseColor(const DevScene& sc, const GltfPBRMaterial& mat, float tu, float tv, const TexGrad& g, ShadeStats& st)
{
  f3 color = mk3(mat.pbrBaseColorFactor[0], mat.pbrBaseColorFactor[1], mat.pbrBaseColorFactor[2]);
  if(mat.pbrBaseColorTexture > -1)
  {
    const f4 t = sampleTex(sc, mat.pbrBaseColorTexture, tu, tv, g, st);
    color = color * mk3(t.x, t.y, t.z);
  }
  return color;
}
// gltf.glsl:34-45
VKRT_DEV void pbrGetMetallicRoughness(const DevScene& sc, const GltfPBRMaterial& mat, float tu, float tv, const TexGrad& g,
                                      float& metallic, float& roughness, ShadeStats& st)
{
  metallic = mat.metallicFactor;
  roughness = mat.roughnessFactor;
  if(mat.metallicRoughnessTexture > -1)
  {
    const f4 t = sampleTex(sc, mat.metallicRoughnessTexture, tu, tv, g, st);
    roughness *= t.y;
    metallic *= t.z;
  }
}
// gltf.glsl:55-66
VKRT_DEV float getNDF_GGXTR(f3 N, f3 H, float alpha)
{
  float a2 = alpha * alpha;
  float NH = dot3(N, H);
  if(NH <= 0.0f)
    return 0.0f;
  float NH2 = NH * NH;
  float d = NH2 * (a2 - 1.0f) + 1.0f;
  return a2 * VKRT_INV_PI / (d * d + 1e-4f);
}
// gltf.glsl:68-78
VKRT_DEV float getG_SchlickGGX(float NV, float k) { return NV / (NV * (1.0f - k) + k); }
VKRT_DEV float getG_Smith(f3 N, f3 V, f3 L, float k)
{
  float NV = fabsf(dot3(N, V));
  float NL = fabsf(dot3(N, L));
  return getG_SchlickGGX(NV, k) * getG_SchlickGGX(NL, k);
}
// gltf.glsl:80-83
VKRT_DEV f3 getF_Schlick(f3 H, f3 V, f3 F0)
{
  return F0 + (mk3(1.0f) - F0) * vk_pow5(1.0f - fabsf(dot3(H, V)));
}
// gltf.glsl:85-96
VKRT_DEV f3 getSpecularBRDF_Cook_Torrance(f3 N, f3 H, f3 V, f3 L, f3 F0, float roughness)
{
  float alpha = roughness * roughness;
  float k = (roughness + 1.0f) * (roughness + 1.0f) / 8.0f;
  float D = getNDF_GGXTR(N, H, alpha);
  float G = getG_Smith(N, V, L, k);
  f3 F = getF_Schlick(H, V, F0);
  float down = 4.0f * fabsf(dot3(V, N)) * fabsf(dot3(L, N)) + 1e-4f;
  return D * F * G / down;
}
// gltf.glsl:98-109
VKRT_DEV f3 getSpecularBRDF_over_pdf_Cook_Torrance(f3 N, f3 H, f3 V, f3 L, f3 F0, float roughness, float ratio)
{
  float k = (roughness + 1.0f) * (roughness + 1.0f) / 8.0f;
  float pdf = (1.0f - ratio) * dot3(N, H) / (4.0f * dot3(L, H) + 1e-4f);
  float G = getG_Smith(N, V, L, k);
  f3 F = getF_Schlick(H, V, F0);
  float down = 4.0f * fabsf(dot3(V, N)) * fabsf(dot3(L, N)) + 1e-4f;
  return (F * G / down) / pdf;
}
// gltf.glsl:111-134.  The GLSL re-evaluates pbrGetBaseColor / pbrGetMetallicRoughness here with the same
// material and texCoord the caller (rchit:111-113) already used; those are pure functions, so the caller's
// values (baseColor and the UNCLAMPED metalness / roughness) are passed in instead of fetching the
// textures a second time.  The reference's texture() calls are still counted (retap).
VKRT_DEV f3 computePBR_BRDF(f3 N, f3 V, f3 L, f3 H, f3 baseColor, float metalness, float roughness)
{
  f3 F0 = mk3(0.04f);
  F0 = glsl_mix(F0, baseColor, metalness);
  f3 F = getF_Schlick(H, V, F0);
  f3 f_cook_torrance = getSpecularBRDF_Cook_Torrance(N, H, V, L, F0, roughness);
  f3 kD = mk3(1.0f) - F;
  kD = kD * (1.0f - metalness);
  f3 f_lambert = baseColor * VKRT_INV_PI;
  f3 diffuse = kD * f_lambert;
  return diffuse + f_cook_torrance;
}
// gltf.glsl:136-154 (non-point lights: 0, with Li = 0 and cosTheta = 0)
VKRT_DEV f3 directLight(const GltfLight& light, f3 P, f3 N, f3 V, f3 baseColor, float metalness, float roughness, unsigned retap,
                        f3& Li, float& cosTheta, ShadeStats& st)
{
  Li = mk3(0.0f);
  cosTheta = 0.0f;
  if(light.type == 0)
  {
    f3 Ldir = mk3(light.position[0], light.position[1], light.position[2]) - P;
    float d = length3(Ldir);
    f3 L = Ldir / d;
    f3 H = normalize3(L + V);
    float attenuation = d * d;
    Li = mk3(light.color[0], light.color[1], light.color[2]) * light.intensity / attenuation;
    cosTheta = glsl_max(dot3(L, N), 0.0f);
    if(cosTheta > 0.0f)
    {
      st.taps += retap;
      return computePBR_BRDF(N, V, L, H, baseColor, metalness, roughness);
    }
  }
  return mk3(0.0f);
}

VKRT_DEV f3 xformPoint(const DevInstance& in, f3 p)
{
  f3 r;
  r.x = ((in.o2w[0] * p.x + in.o2w[1] * p.y) + in.o2w[2] * p.z) + in.o2w[3];
  r.y = ((in.o2w[4] * p.x + in.o2w[5] * p.y) + in.o2w[6] * p.z) + in.o2w[7];
  r.z = ((in.o2w[8] * p.x + in.o2w[9] * p.y) + in.o2w[10] * p.z) + in.o2w[11];
  return r;
}
// vec3(n * gl_WorldToObjectEXT): component j = dot(n, column j of W2O)
VKRT_DEV f3 xformNormal(const DevInstance& in, f3 n)
{
  f3 r;
  r.x = (n.x * in.w2o[0] + n.y * in.w2o[3]) + n.z * in.w2o[6];
  r.y = (n.x * in.w2o[1] + n.y * in.w2o[4]) + n.z * in.w2o[7];
  r.z = (n.x * in.w2o[2] + n.y * in.w2o[5]) + n.z * in.w2o[8];
  return r;
}

// raytrace.rchit:31-219 in three pieces, so that the wavefront shading stage can regroup hits by lobe between them
// (wavefront.hip shadeClosestBlock); run back to back (closestHitShaderInst) they are the shader as written.
//   closestHitFront  rchit:34-125   attribute fetch + interpolation, the four texture() taps, normal mapping, material
//   closestHitLobe   rchit:127-131  ratio from the unclamped metalness, the lobe draw r1 < ratio
//   closestHitTail   rchit:126-216  clamps, diffuse (NEE + cosine sample) or specular (GGX sample) branch, payload outputs
struct HitMid
{
  f3 worldPos, N, tangent, binormal, V, baseColor, emittance;
  float metalU, roughU;  // UNCLAMPED metalness / roughness (ratio and directLight use them as fetched)
  unsigned retap;        // texture() calls computePBR_BRDF would re-issue (counted, not fetched twice)
};

// `ts` = sc.triShade[hit.slot] (vertex indices + material), fetched by the caller (the wavefront traversal kernel
// leaves it in the path record next to the hit).
// `instId` = the instance the hit triangle belongs to (third word of its last triangle-record quad).  hit.slot is not read.
// Reads prd.depth / prd.isSpecular (emission rule, rchit:83); writes nothing to prd.
VKRT_DEV void closestHitFront(const DevScene& sc, const RayHit& hit, const uint32_t instId, const uint4 ts, f3 worldRayDir, const Payload& prd,
                              ShadeStats& st, HitMid& mid)
{
  st.hits++;
  // rchit:34-50 (PrimMeshInfo lookup, three index fetches, vertexOffset, max(0, materialIndex)) is resolved once
  // per triangle at build time into a 16-byte record, so the attribute fetch below is one hop from the hit.
  const uint32_t i0 = ts.x, i1 = ts.y, i2 = ts.z, matIndex = ts.w;
  const f3 b = mk3(1.0f - hit.u - hit.v, hit.u, hit.v);  // rchit:68

  const BufView vPN = bufView(sc.vertexPN), vMat = bufView(sc.shadeMaterials), vInst = bufView(sc.instances);
  const float4 a0 = bufLoad4(vPN, VKRT_VERTEX_BYTES * i0), b0 = bufLoad4(vPN, VKRT_VERTEX_BYTES * i0 + 16u), tq0 = bufLoad4(vPN, VKRT_VERTEX_BYTES * i0 + 32u);
  const float4 a1 = bufLoad4(vPN, VKRT_VERTEX_BYTES * i1), b1 = bufLoad4(vPN, VKRT_VERTEX_BYTES * i1 + 16u), tq1 = bufLoad4(vPN, VKRT_VERTEX_BYTES * i1 + 32u);
  const float4 a2 = bufLoad4(vPN, VKRT_VERTEX_BYTES * i2), b2 = bufLoad4(vPN, VKRT_VERTEX_BYTES * i2 + 16u), tq2 = bufLoad4(vPN, VKRT_VERTEX_BYTES * i2 + 32u);
  // material: four aligned 16-byte loads of the 64-byte hit-shader record (DevShadeMaterial: the factors rchit reads + the four texture
  // references; the 128-byte DevMaterial took eight).  (Staging the material and instance tables in LDS per workgroup -- ds_read_b128
  // instead of lane-loads that nearly always hit L1 -- made the stage 9 % SLOWER: the copy costs more than those hot loads did;
  // profiles/r04_experiments.md #115)
  const uint32_t mo = 64u * matIndex;
  const float4 m0 = bufLoad4(vMat, mo), m1 = bufLoad4(vMat, mo + 16u), ref01 = bufLoad4(vMat, mo + 32u), ref23 = bufLoad4(vMat, mo + 48u);
  const uint32_t dimB = __float_as_uint(ref01.x), baseB = __float_as_uint(ref01.y), dimM = __float_as_uint(ref01.z), baseM = __float_as_uint(ref01.w);
  const uint32_t dimN = __float_as_uint(ref23.x), baseN = __float_as_uint(ref23.y), dimE = __float_as_uint(ref23.z), baseE = __float_as_uint(ref23.w);
  DevInstance in;
  {
    const uint32_t io = 96u * instId;  // 96-byte record: six aligned 16-byte loads
    const float4 q0 = bufLoad4(vInst, io), q1 = bufLoad4(vInst, io + 16u), q2 = bufLoad4(vInst, io + 32u), q3 = bufLoad4(vInst, io + 48u), q4 = bufLoad4(vInst, io + 64u),
                 q5 = bufLoad4(vInst, io + 80u);
    in.o2w[0] = q0.x; in.o2w[1] = q0.y; in.o2w[2] = q0.z; in.o2w[3] = q0.w; in.o2w[4] = q1.x; in.o2w[5] = q1.y; in.o2w[6] = q1.z; in.o2w[7] = q1.w;
    in.o2w[8] = q2.x; in.o2w[9] = q2.y; in.o2w[10] = q2.z; in.o2w[11] = q2.w;
    in.w2o[0] = q3.x; in.w2o[1] = q3.y; in.w2o[2] = q3.z; in.w2o[3] = q3.w; in.w2o[4] = q4.x; in.w2o[5] = q4.y; in.w2o[6] = q4.z; in.w2o[7] = q4.w;
    in.w2o[8] = q5.x; in.primMesh = __float_as_int(q5.y); in.pad[0] = 0; in.pad[1] = 0;
  }
  GltfPBRMaterial mat;  // (the fields this shader reads; a texture index only matters as "> -1": bit 15 of the reference's first word)
  mat.pbrBaseColorFactor[0] = m0.x; mat.pbrBaseColorFactor[1] = m0.y; mat.pbrBaseColorFactor[2] = m0.z; mat.pbrBaseColorFactor[3] = 1.0f;
  mat.metallicFactor = m0.w; mat.roughnessFactor = m1.x;
  mat.emissiveFactor[0] = m1.y; mat.emissiveFactor[1] = m1.z; mat.emissiveFactor[2] = m1.w;
  mat.pbrBaseColorTexture = (dimB & 0x8000u) ? 0 : -1; mat.metallicRoughnessTexture = (dimM & 0x8000u) ? 0 : -1;
  mat.normalTexture = (dimN & 0x8000u) ? 0 : -1; mat.emissiveTexture = (dimE & 0x8000u) ? 0 : -1;
  const float tu = (b0.z * b.x + b1.z * b.y) + b2.z * b.z;
  const float tv = (b0.w * b.x + b1.w * b.y) + b2.w * b.z;

  // The four texture() calls of rchit:83-113 (emissive, normal, base colour, metallic-roughness): footprints first, then
  // all sixteen texel loads, then the decode, so the loads overlap instead of forming four dependent round trips.
  // A tap the reference would not issue reads texel 0 and is discarded.
  const bool emits = prd.depth == 0 || prd.isSpecular;  // rchit:83
  const bool wantE = emits && mat.emissiveTexture > -1, wantN = mat.normalTexture > -1;
  const bool wantB = mat.pbrBaseColorTexture > -1, wantM = mat.metallicRoughnessTexture > -1;
  st.taps += (wantE ? 1u : 0u) + (wantN ? 1u : 0u) + (wantB ? 1u : 0u) + (wantM ? 1u : 0u);
  TexTap tE, tN, tB, tM;
  // (with a footprint pool the tap's texel number is relative to the texture: the pool's record number is added below)
#define VKRT_FOOTPRINT(dim, base, want, tap)                                                                                           \
  texFootprint(sc.texQuads ? 0u : ((base) & 0x7fffffffu), ((dim) & 0x7fffu) + 1u, (((dim) >> 16) & 0x7fffu) + 1u, ((base) >> 31) != 0u, \
               ((dim) >> 31) != 0u, want, tu, tv, tap)
  VKRT_FOOTPRINT(dimE, baseE, wantE, tE);
  VKRT_FOOTPRINT(dimN, baseN, wantN, tN);
  VKRT_FOOTPRINT(dimB, baseB, wantB, tB);
  VKRT_FOOTPRINT(dimM, baseM, wantM, tM);
#undef VKRT_FOOTPRINT
  uint32_t e00, e10, e01, e11, n00, n10, n01, n11, c00, c10, c01, c11, r00, r10, r01, r11;
  if(sc.texQuads)
  {
    // one 16-byte record per tap: the 2x2 footprint at (x0, y0) with the wrap already applied (DevScene::texQuads); the record
    // number is the texture's first record + the level-0 texel number of i00 inside the texture
    const BufView quads = bufView(sc.texQuads);
#define VKRT_QUAD(dim, base, tap, want, a, b, c, d)                                                                                    \
  {                                                                                                                                    \
    const uint32_t rec_ = ((want) && ((dim) >> 31) != 0u) ? tap.i00 + ((base) & 0x7fffffffu) : 0u;                                        \
    const float4 v_ = bufLoad4(quads, 16u * rec_);                                                                                     \
    a = __float_as_uint(v_.x); b = __float_as_uint(v_.y); c = __float_as_uint(v_.z); d = __float_as_uint(v_.w);                        \
  }
    VKRT_QUAD(dimE, baseE, tE, wantE, e00, e10, e01, e11)
    VKRT_QUAD(dimN, baseN, tN, wantN, n00, n10, n01, n11)
    VKRT_QUAD(dimB, baseB, tB, wantB, c00, c10, c01, c11)
    VKRT_QUAD(dimM, baseM, tM, wantM, r00, r10, r01, r11)
#undef VKRT_QUAD
  }
  else
  {
    const BufView tex = bufView(sc.texels);
#define VKRT_TEXEL(i) bufLoad1(tex, 4u * (i))
    e00 = VKRT_TEXEL(tE.i00); e10 = VKRT_TEXEL(tE.i10); e01 = VKRT_TEXEL(tE.i01); e11 = VKRT_TEXEL(tE.i11);
    n00 = VKRT_TEXEL(tN.i00); n10 = VKRT_TEXEL(tN.i10); n01 = VKRT_TEXEL(tN.i01); n11 = VKRT_TEXEL(tN.i11);
    c00 = VKRT_TEXEL(tB.i00); c10 = VKRT_TEXEL(tB.i10); c01 = VKRT_TEXEL(tB.i01); c11 = VKRT_TEXEL(tB.i11);
    r00 = VKRT_TEXEL(tM.i00); r10 = VKRT_TEXEL(tM.i10); r01 = VKRT_TEXEL(tM.i01); r11 = VKRT_TEXEL(tM.i11);
#undef VKRT_TEXEL
  }

  const f3 pos = mk3(a0.x, a0.y, a0.z) * b.x + mk3(a1.x, a1.y, a1.z) * b.y + mk3(a2.x, a2.y, a2.z) * b.z;
  const f3 worldPos = xformPoint(in, pos);
  const f3 nrm = normalize3(mk3(a0.w, b0.x, b0.y) * b.x + mk3(a1.w, b1.x, b1.y) * b.y + mk3(a2.w, b2.x, b2.y) * b.z);
  const f3 worldNrm = normalize3(xformNormal(in, nrm));
  const f3 tag = normalize3(mk3(tq0.x, tq0.y, tq0.z) * b.x + mk3(tq1.x, tq1.y, tq1.z) * b.y + mk3(tq2.x, tq2.y, tq2.z) * b.z);
  f3 worldTag = normalize3(xformNormal(in, tag));
  worldTag = normalize3(worldTag - dot3(worldTag, worldNrm) * worldNrm);
  const f3 worldBin = tq0.w * cross3(worldNrm, worldTag);

  f3 emittance = mk3(0.0f);
  if(emits)  // rchit:83
  {
    emittance = mk3(mat.emissiveFactor[0], mat.emissiveFactor[1], mat.emissiveFactor[2]);
    if(wantE)
    {
      const f4 t = texBlend(st.lut, tE, e00, e10, e01, e11);
      emittance = emittance * mk3(t.x, t.y, t.z);
    }
  }
  f3 tangent = worldTag, binormal = worldBin;
  f3 texNormal = worldNrm;
  if(wantN)  // rchit:100-106
  {
    const f4 t = texBlend(st.lut, tN, n00, n10, n01, n11);
    texNormal = normalize3(mk3(t.x, t.y, t.z) * 2.0f - mk3(1.0f));
    texNormal = normalize3(tangent * texNormal.x + binormal * texNormal.y + worldNrm * texNormal.z);
    createCoordinateSystem(texNormal, tangent, binormal);
  }
  f3 baseColor = mk3(mat.pbrBaseColorFactor[0], mat.pbrBaseColorFactor[1], mat.pbrBaseColorFactor[2]);  // gltf.glsl:26-32
  if(wantB)
  {
    const f4 t = texBlend(st.lut, tB, c00, c10, c01, c11);
    baseColor = baseColor * mk3(t.x, t.y, t.z);
  }
  float metalness = mat.metallicFactor, roughness = mat.roughnessFactor;  // gltf.glsl:34-45
  if(wantM)
  {
    const f4 t = texBlend(st.lut, tM, r00, r10, r01, r11);
    roughness *= t.y;
    metalness *= t.z;
  }

  mid.worldPos = worldPos; mid.N = texNormal; mid.tangent = tangent; mid.binormal = binormal;
  mid.V = normalize3(-worldRayDir);
  mid.baseColor = baseColor; mid.emittance = emittance;
  mid.metalU = metalness; mid.roughU = roughness;
  mid.retap = (mat.pbrBaseColorTexture > -1 ? 1u : 0u) + (mat.metallicRoughnessTexture > -1 ? 1u : 0u);
}

// rchit:127,130-131: true = diffuse lobe.  Advances prd.seed by the one draw.
VKRT_DEV bool closestHitLobe(const HitMid& mid, Payload& prd)
{
  const float ratio = 0.5f * (1.0f - mid.metalU);  // rchit:127 (before the clamps)
  const float r1 = rnd(prd.seed);
  return r1 < ratio;
}

VKRT_DEV void closestHitTail(const DevScene& sc, const PushConstantRay& pc, const HitMid& mid, const bool diffuse, Payload& prd, ShadeStats& st)
{
  const f3 worldPos = mid.worldPos, texNormal = mid.N, tangent = mid.tangent, binormal = mid.binormal, V = mid.V, N = mid.N;
  const f3 baseColor = mid.baseColor;
  f3 emittance = mid.emittance;
  const float metalU = mid.metalU, roughU = mid.roughU;  // unclamped values, as directLight re-derives them
  const unsigned retap = mid.retap;
  const float ratio = 0.5f * (1.0f - metalU);
  const float roughness = glsl_clamp(roughU, 0.01f, 0.99f);  // rchit:128-129
  const float metalness = glsl_clamp(metalU, 0.01f, 0.99f);
  const f3 rayOrigin = worldPos;
  f3 rayDirection;
  float pdf;
  f3 BRDF;
  if(diffuse)
  {
    st.diffuse++;
    prd.isSpecular = false;
    const int random_index = (int)(rnd(prd.seed) * (float)pc.lightsCount);
    const float4* lq = (const float4*)&sc.lights[random_index];  // 32-byte record, two aligned loads
    const float4 l0 = lq[0], l1 = lq[1];
    GltfLight light;
    light.position[0] = l0.x; light.position[1] = l0.y; light.position[2] = l0.z;
    light.color[0] = l0.w; light.color[1] = l1.x; light.color[2] = l1.y;
    light.intensity = l1.z; light.type = __float_as_int(l1.w);
    const f3 lightDir = mk3(light.position[0], light.position[1], light.position[2]) - worldPos;
    const float lightDistance = length3(lightDir);
    const f3 L = normalize3(lightDir);
    prd.lightDist = lightDistance;
    prd.shadowRayDir = L;
    if(dot3(L, texNormal) <= 0)
      emittance = emittance + mk3(0.0f);
    else
    {
      f3 Li;
      float cosTheta;
      const f3 brdf = directLight(light, worldPos, texNormal, V, baseColor, metalU, roughU, retap, Li, cosTheta, st);
      emittance = emittance + (float)pc.lightsCount * brdf * Li * cosTheta;
    }
    rayDirection = normalize3(samplingHemisphere(prd.seed, tangent, binormal, texNormal));
    pdf = ratio * dot3(rayDirection, texNormal) * VKRT_INV_PI;
    BRDF = (1.0f - metalness) * baseColor * VKRT_INV_PI;
  }
  else
  {
    prd.isSpecular = true;
    const float alpha = roughness * roughness;
    const f3 h = samplingNDF_GGXTR(prd.seed, alpha * alpha);
    const f3 H = normalize3(tangent * h.x + binormal * h.y + texNormal * h.z);
    const f3 L = normalize3(glsl_reflect(-V, H));
    rayDirection = L;
    f3 F0 = mk3(0.04f);
    F0 = glsl_mix(F0, baseColor, metalness);
    pdf = 1.0f;
    BRDF = getSpecularBRDF_over_pdf_Cook_Torrance(N, H, V, L, F0, roughness, ratio);
  }
  const float cosTheta = dot3(rayDirection, texNormal);
  prd.rayOrigin = rayOrigin;
  prd.rayDirection = rayDirection;
  prd.hitValue = emittance;
  prd.weight = BRDF * cosTheta / pdf;
}

VKRT_DEV void closestHitShaderInst(const DevScene& sc, const PushConstantRay& pc, const RayHit& hit, const uint32_t instId, const uint4 ts,
                                   f3 worldRayDir, Payload& prd, ShadeStats& st)
{
  HitMid mid;
  closestHitFront(sc, hit, instId, ts, worldRayDir, prd, st, mid);
  const bool diffuse = closestHitLobe(mid, prd);
  closestHitTail(sc, pc, mid, diffuse, prd, st);
}

VKRT_DEV void closestHitShader(const DevScene& sc, const PushConstantRay& pc, const RayHit& hit, const uint4 ts, f3 worldRayDir, Payload& prd,
                               ShadeStats& st)
{
  const uint32_t instId = (uint32_t)__float_as_int(sc.tris[hit.slot * VKRT_TRI_QUADS + 2].z);
  closestHitShaderInst(sc, pc, hit, instId, ts, worldRayDir, prd, st);
}
VKRT_DEV void closestHitShader(const DevScene& sc, const PushConstantRay& pc, const RayHit& hit, f3 worldRayDir, Payload& prd, ShadeStats& st)
{
  closestHitShader(sc, pc, hit, sc.triShade[hit.slot], worldRayDir, prd, st);
}

// raytrace.rmiss:11-19
VKRT_DEV void missShader(const PushConstantRay& pc, Payload& prd)
{
  if(prd.depth == 0)
    prd.hitValue = mk3(pc.clearColor[0], pc.clearColor[1], pc.clearColor[2]) * 0.8f;
  else
    prd.hitValue = mk3(0.01f);
  prd.depth = 100;
}

VKRT_DEV void mat4MulVec4(const float* M, float v0, float v1, float v2, float v3, float* out)
{
#pragma unroll
  for(int i = 0; i < 4; i++)
    out[i] = ((M[0 + i] * v0 + M[4 + i] * v1) + M[8 + i] * v2) + M[12 + i] * v3;
}
