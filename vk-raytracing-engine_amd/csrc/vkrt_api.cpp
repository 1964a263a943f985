// vkrt_api.cpp -- implementation of the C ABI declared in include/vkrt.h (compiled as HIP).
//
// Scene upload mirrors HelloVulkan::loadGltfScene's device-local buffers (hello_vulkan.cpp:353-381);
// vkrt_accel_build stands in for createBottomLevelASGltf/createTopLevelAsGltf (:1001-1047);
// vkrt_pathtrace stands in for HelloVulkan::pathtrace (:1423-1448).  There is no CPU fallback:
// without a HIP device the compute entry points fail with VKRT_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/vkrt.h"
#include "bvh_host.h"
#include "device_scene.h"
#include "kernels.h"
#include "wide_node.h"

#define VKRT_TRI_THRESHOLD_DEFAULT 32
#define VKRT_WF_SHARE_DEFAULT 16
#define VKRT_WF_SHARE_FLAGS_DEFAULT 25
#define VKRT_WF_FRAMES_IN_FLIGHT_DEFAULT 3
#define VKRT_SPLIT_BUDGET_DEFAULT -1  // automatic (round 5): vkrt_accel_build decides per scene
#include "lbvh.h"

namespace {

thread_local std::string g_lastError;

int fail(int code, const char* fmt, ...)
{
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_lastError = buf;
  return code;
}

#define HIP_TRY(expr)                                                                                  \
  do                                                                                                   \
  {                                                                                                    \
    hipError_t e_ = (expr);                                                                            \
    if(e_ != hipSuccess)                                                                               \
      return fail(e_ == hipErrorOutOfMemory ? VKRT_ERR_OUT_OF_MEMORY : VKRT_ERR_HIP, "%s: %s", #expr, \
                  hipGetErrorString(e_));                                                              \
  } while(0)

}  // namespace

struct vkrt_scene
{
  int device = 0;
  int cuCount = 256;
  // host copies (kept like m_gltfScene keeps its vectors)
  std::vector<float> positions;
  std::vector<uint32_t> indices;
  std::vector<vkrt_prim_mesh> primMeshes;
  std::vector<vkrt_node> nodes;
  uint32_t lightCount = 0, materialCount = 0;
  std::vector<float> materialAlpha;  // pbrBaseColorFactor.a per material (the any-hit stage's dissolve), host copy for the host builder
  // device allocations
  std::vector<void*> allocs;
  DevScene dev{};
  void* accelNodes = nullptr;
  void* accelTris = nullptr;
  void* accelShade = nullptr;
  bool built = false;
  vkrt_accel_info info{};
  unsigned int* workCounter = nullptr;
  DevCounters* counters = nullptr;
  hipEvent_t evStart = nullptr, evStop = nullptr;
  bool timed = false;
  // wavefront mode working set
  void* wfMem = nullptr;
  WfBuffers wf{};
  WfAsync wfAsync{};
  std::vector<hipEvent_t> wfEvents;
  WfTiming wfTiming{};
  bool wfTimed = false;
  int splitResolved = 0;  // the budget the last vkrt_accel_build used (VKRT_INFO_SPLIT_BUDGET): what -1 resolved to
  int wfTimingRounds = 0;  // rounds per frame of the last timed call (vkrt_last_trace_timing: which gaps are shade launches)
  // execution options (include/vkrt.h vkrt_option); index = option id
  std::vector<hipEvent_t> wfPool;  // events ordering the lanes of one call (kernels.h WfAsync::pool)
  int opt[VKRT_OPT_LAST + 1] = {0, 1, 1, 3, 64, VKRT_WF_SHARE_DEFAULT, VKRT_TRI_THRESHOLD_DEFAULT, 0, VKRT_WF_SHARE_FLAGS_DEFAULT, 1, 0, 0, 0,
                                VKRT_WF_FRAMES_IN_FLIGHT_DEFAULT, VKRT_SPLIT_BUDGET_DEFAULT};
  bool hasLargeTriangles = false;  // some instanced triangle covers more than 1 % of the largest face of the scene's box (any-hit order heuristic)
  bool wavefront = true;  // execution mode the acceleration structure was built for (opt[VKRT_OPT_MODE] at vkrt_accel_build)
};

namespace {

template <typename T>
int upload(vkrt_scene* s, const T* src, size_t count, const T** dst)
{
  void* p = nullptr;
  const size_t bytes = std::max<size_t>(count * sizeof(T), 16);
  HIP_TRY(hipMalloc(&p, bytes));
  s->allocs.push_back(p);
  if(count)
    HIP_TRY(hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice));
  *dst = (const T*)p;
  return VKRT_OK;
}

int validate(const vkrt_scene_desc* d)
{
  if(!d)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "desc is NULL");
  if(d->struct_size != sizeof(vkrt_scene_desc))
    return fail(VKRT_ERR_INVALID_ARGUMENT, "vkrt_scene_desc.struct_size %u != %zu (ABI mismatch)", d->struct_size,
                sizeof(vkrt_scene_desc));
  if(d->vertex_count && (!d->positions || !d->normals || !d->tangents || !d->texcoords0))
    return fail(VKRT_ERR_INVALID_ARGUMENT, "vertex arrays missing");
  if(d->index_count && !d->indices)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "indices missing");
  if(d->material_count == 0 || !d->materials)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "at least one material is required (nvh::GltfScene appends a default one)");
  if(d->light_count == 0 || !d->lights)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "at least one light is required (hello_vulkan.cpp:247 adds 8 fallback lights)");
  if((d->prim_mesh_count && !d->prim_meshes) || (d->node_count && !d->nodes) || (d->texture_count && !d->textures))
    return fail(VKRT_ERR_INVALID_ARGUMENT, "array pointer missing");
  for(uint32_t i = 0; i < d->prim_mesh_count; i++)
  {
    const vkrt_prim_mesh& p = d->prim_meshes[i];
    if((uint64_t)p.firstIndex + p.indexCount > d->index_count || (uint64_t)p.vertexOffset + p.vertexCount > d->vertex_count)
      return fail(VKRT_ERR_INVALID_ARGUMENT, "primMesh %u: range outside the index/vertex arrays", i);
    if(p.materialIndex >= (int32_t)d->material_count)
      return fail(VKRT_ERR_INVALID_ARGUMENT, "primMesh %u: materialIndex %d out of range", i, p.materialIndex);
    const uint32_t triIdx = (p.indexCount / 3) * 3;
    for(uint32_t k = 0; k < triIdx; k++)
      if(d->indices[p.firstIndex + k] >= p.vertexCount)
        return fail(VKRT_ERR_INVALID_ARGUMENT, "primMesh %u: index %u >= vertexCount %u", i, d->indices[p.firstIndex + k],
                    p.vertexCount);
  }
  for(uint32_t i = 0; i < d->node_count; i++)
  {
    if(d->nodes[i].primMesh < 0 || (uint32_t)d->nodes[i].primMesh >= d->prim_mesh_count)
      return fail(VKRT_ERR_INVALID_ARGUMENT, "node %u: primMesh %d out of range", i, d->nodes[i].primMesh);
    for(int k = 0; k < 16; k++)
      if(!std::isfinite(d->nodes[i].worldMatrix[k]))
        return fail(VKRT_ERR_INVALID_ARGUMENT, "node %u: worldMatrix[%d] is not finite", i, k);
  }
  // the builders compare and quantise boxes: NaN / inf coordinates have no place in an acceleration structure (a Vulkan driver is
  // free to drop such triangles; here they are refused up front)
  for(size_t i = 0; i < (size_t)d->vertex_count * 3; i++)
    if(!std::isfinite(d->positions[i]))
      return fail(VKRT_ERR_INVALID_ARGUMENT, "positions[%zu] (vertex %zu) is not finite", i, i / 3);
  for(uint32_t i = 0; i < d->material_count; i++)
  {
    const GltfPBRMaterial& m = d->materials[i];
    const int32_t t[4] = {m.pbrBaseColorTexture, m.metallicRoughnessTexture, m.normalTexture, m.emissiveTexture};
    for(int k = 0; k < 4; k++)
      if(t[k] >= (int32_t)d->texture_count && d->texture_count > 0)
        return fail(VKRT_ERR_INVALID_ARGUMENT, "material %u: texture index %d out of range", i, t[k]);
  }
  for(uint32_t i = 0; i < d->texture_count; i++)
    if(!d->textures[i].rgba8 || d->textures[i].width == 0 || d->textures[i].height == 0)
      return fail(VKRT_ERR_INVALID_ARGUMENT, "texture %u: empty", i);
  return VKRT_OK;
}

// Mip chain of one RGBA8 image as nvvk::cmdGenerateMipmaps builds it (hello_vulkan.cpp:496): level L is blitted from level
// L - 1 with VK_FILTER_LINEAR to max(1, size / 2).  vkCmdBlitImage semantics: destination texel centre x + 0.5 maps to the
// unnormalised source coordinate (x + 0.5) * srcSize / dstSize, filtered bilinearly around it with clamp-to-edge; an sRGB
// image is filtered in linear space and re-encoded.  Even sizes reduce to the 2x2 box average.
float srgbEncode(float c) { return c <= 0.0031308f ? 12.92f * c : 1.055f * powf(c, 1.0f / 2.4f) - 0.055f; }
void downsampleLevel(const uint32_t* src, uint32_t sw, uint32_t sh, uint32_t* dst, uint32_t dw, uint32_t dh, bool srgb, const float* lut512)
{
  for(uint32_t y = 0; y < dh; y++)
    for(uint32_t x = 0; x < dw; x++)
    {
      const float fu = ((float)x + 0.5f) * ((float)sw / (float)dw) - 0.5f, fv = ((float)y + 0.5f) * ((float)sh / (float)dh) - 0.5f;
      const float flx = floorf(fu), fly = floorf(fv), ax = fu - flx, ay = fv - fly;
      auto cl = [](int v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); };
      const int x0 = cl((int)flx, (int)sw), x1 = cl((int)flx + 1, (int)sw), y0 = cl((int)fly, (int)sh), y1 = cl((int)fly + 1, (int)sh);
      const uint32_t p[4] = {src[(size_t)y0 * sw + x0], src[(size_t)y0 * sw + x1], src[(size_t)y1 * sw + x0], src[(size_t)y1 * sw + x1]};
      uint32_t out = 0;
      for(int c = 0; c < 4; c++)
      {
        const uint32_t base = (srgb && c < 3) ? 0u : 256u;
        const float t00 = lut512[base + ((p[0] >> (8 * c)) & 255u)], t10 = lut512[base + ((p[1] >> (8 * c)) & 255u)];
        const float t01 = lut512[base + ((p[2] >> (8 * c)) & 255u)], t11 = lut512[base + ((p[3] >> (8 * c)) & 255u)];
        float v = (t00 * (1.0f - ax) + t10 * ax) * (1.0f - ay) + (t01 * (1.0f - ax) + t11 * ax) * ay;
        if(srgb && c < 3) v = srgbEncode(v);
        v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
        out |= (uint32_t)(v * 255.0f + 0.5f) << (8 * c);
      }
      dst[(size_t)y * dw + x] = out;
    }
}

// execution mode: wavefront pipeline (default) or the single persistent megakernel
bool useWavefront(const vkrt_scene* s) { return s->opt[VKRT_OPT_MODE] == 1; }

int clampOption(int option, int v)
{
  switch(option)
  {
    case VKRT_OPT_MODE: case VKRT_OPT_BVH_LAYOUT: return v ? 1 : 0;
    case VKRT_OPT_WF_SUBFRAMES: case VKRT_OPT_WF_FRAMES_IN_FLIGHT: return std::max(1, std::min(VKRT_WF_MAX_LANES, v));
    case VKRT_OPT_SPLIT_BUDGET: return std::max(-1, std::min(100, v));  // -1 = automatic (vkrt_accel_build)
    case VKRT_OPT_WF_TRAV_BLOCK: return v == 256 ? 256 : v == 128 ? 128 : 64;
    case VKRT_OPT_WF_SHARE: return std::max(0, std::min(64, v));
    case VKRT_OPT_TRI_THRESHOLD: return std::max(0, std::min(65, v));
    case VKRT_OPT_WF_SHARE_PERIOD: return std::max(0, std::min(255, v));
    case VKRT_OPT_WF_SHARE_FLAGS: return v & 31;
    case VKRT_OPT_GBUFFER_MIPS: case VKRT_OPT_WATERTIGHT: case VKRT_OPT_SKIP_DEAD_SHADOW_RAYS: case VKRT_OPT_ANYHIT_DISSOLVE: return v ? 1 : 0;
  }
  return v;
}

// Initial option values: the process-wide test hooks documented in include/vkrt.h, read once per scene handle.
void optionsFromEnvironment(vkrt_scene* s)
{
  const char* e;
  if((e = getenv("VKRT_MODE")) && !strcmp(e, "mega")) s->opt[VKRT_OPT_MODE] = 0;
  if((e = getenv("VKRT_BVH")) && !strcmp(e, "bvh2")) s->opt[VKRT_OPT_BVH_LAYOUT] = 0;
  const struct { const char* name; int option; } ints[] = {{"VKRT_WF_SUBFRAMES", VKRT_OPT_WF_SUBFRAMES}, {"VKRT_WF_TRAV_BLOCK", VKRT_OPT_WF_TRAV_BLOCK},
                                                          {"VKRT_WF_SHARE", VKRT_OPT_WF_SHARE}, {"VKRT_TRI_THRESHOLD", VKRT_OPT_TRI_THRESHOLD},
                                                          {"VKRT_WF_SHARE_PERIOD", VKRT_OPT_WF_SHARE_PERIOD}, {"VKRT_WF_SHARE_FLAGS", VKRT_OPT_WF_SHARE_FLAGS},
                                                          {"VKRT_GBUFFER_MIPS", VKRT_OPT_GBUFFER_MIPS}, {"VKRT_WATERTIGHT", VKRT_OPT_WATERTIGHT},
                                                          {"VKRT_SKIP_DEAD_SHADOW_RAYS", VKRT_OPT_SKIP_DEAD_SHADOW_RAYS},
                                                          {"VKRT_ANYHIT_DISSOLVE", VKRT_OPT_ANYHIT_DISSOLVE}, {"VKRT_WF_FRAMES_IN_FLIGHT", VKRT_OPT_WF_FRAMES_IN_FLIGHT},
                                                          {"VKRT_SPLIT_BUDGET", VKRT_OPT_SPLIT_BUDGET}};
  for(const auto& k : ints)
    if((e = getenv(k.name)))
      s->opt[k.option] = clampOption(k.option, atoi(e));
}

void freeAccel(vkrt_scene* s)
{
  if(s->accelNodes) (void)hipFree(s->accelNodes);
  if(s->accelTris) (void)hipFree(s->accelTris);
  if(s->accelShade) (void)hipFree(s->accelShade);
  s->accelNodes = s->accelTris = s->accelShade = nullptr;
  s->built = false;
}

int setDevice(const vkrt_scene* s)
{
  HIP_TRY(hipSetDevice(s->device));
  return VKRT_OK;
}

// Working set of the wavefront pipeline for `paths` path records (whole 8x8 tiles of the shard) in each of `groups` frame groups
// (frames in flight) + the internal streams of the lanes.  Growing it is the only place a trace call may synchronise with the
// host (vkrt_reserve).
int ensureWorkingSet(vkrt_scene* s, uint32_t paths, int groups, hipStream_t stream)
{
  groups = std::max(1, std::min(VKRT_WF_MAX_LANES, groups));
  // wf_streams.h rec(): a record's byte offset inside a plane is a 32-bit lane offset (16 B x slot); 2^28 path records and more
  // (16384 x 16384 pixels in one shard, ~146 GB of streams) would wrap it silently
  if(paths >= (1u << 28))
    return fail(VKRT_ERR_UNSUPPORTED, "wavefront mode: %u path records in one shard (limit 2^28 - 1: split the launch into shards)", paths);
  if(!s->wfMem || s->wf.capacity < paths || (int)s->wf.groups < groups)
  {
    HIP_TRY(hipStreamSynchronize(stream));  // an earlier frame may still be using the smaller buffer
    if(s->wfMem) (void)hipFree(s->wfMem);
    s->wfMem = nullptr;
    paths = std::max(paths, s->wf.capacity);
    groups = std::max(groups, (int)s->wf.groups);
    s->wf = WfBuffers{};
    HIP_TRY(hipMalloc(&s->wfMem, vkrt_wf_state_bytes(paths, groups)));
    vkrt_wf_carve(s->wfMem, paths, groups, &s->wf);
  }
  if(s->wfAsync.count == 0)
  {
    HIP_TRY(hipEventCreateWithFlags(&s->wfAsync.fork, hipEventDisableTiming));
    for(int j = 0; j < VKRT_WF_MAX_LANES; j++)
    {
      HIP_TRY(hipStreamCreateWithFlags(&s->wfAsync.streams[j], hipStreamNonBlocking));
      HIP_TRY(hipEventCreateWithFlags(&s->wfAsync.join[j], hipEventDisableTiming));
    }
    s->wfAsync.count = VKRT_WF_MAX_LANES;
  }
  return VKRT_OK;
}

// events that order the lanes of a call of `frames` frames (created once, kept; no device synchronisation)
int ensureEventPool(vkrt_scene* s, int frames)
{
  const size_t want = (size_t)vkrt_wf_pool_events(frames, VKRT_WF_MAX_LANES);
  while(s->wfPool.size() < want)
  {
    hipEvent_t e;
    HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    s->wfPool.push_back(e);
  }
  s->wfAsync.pool = s->wfPool.data();
  s->wfAsync.poolSize = (int)s->wfPool.size();
  return VKRT_OK;
}

// frames a call keeps in flight: the option, but never more than the call has
int framesInFlight(const vkrt_scene* s, int frames) { return std::max(1, std::min(s->opt[VKRT_OPT_WF_FRAMES_IN_FLIGHT], frames)); }
#define VKRT_FRAMES_PER_BATCH 32  // a longer call is rendered as batches of this many frames (bounds the event pool)

}  // namespace

extern "C" {

int vkrt_abi_version(void) { return VKRT_ABI_VERSION; }
const char* vkrt_last_error(void) { return g_lastError.c_str(); }

int vkrt_device_count(void)
{
  int n = 0;
  if(hipGetDeviceCount(&n) != hipSuccess)
    return 0;
  return n;
}

int vkrt_scene_create(const vkrt_scene_desc* d, int device, vkrt_scene** out)
{
  if(!out)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "out is NULL");
  *out = nullptr;
  int rc = validate(d);
  if(rc != VKRT_OK)
    return rc;
  const int ndev = vkrt_device_count();
  if(ndev <= 0)
    return fail(VKRT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
  if(device < 0 || device >= ndev)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "device %d out of range [0,%d)", device, ndev);
  vkrt_scene* s = new vkrt_scene();
  s->device = device;
  optionsFromEnvironment(s);
  auto bail = [&](int code) {
    vkrt_scene_destroy(s);
    return code;
  };
  if((rc = setDevice(s)) != VKRT_OK)
    return bail(rc);
  hipDeviceProp_t prop;
  if(hipGetDeviceProperties(&prop, device) == hipSuccess)
    s->cuCount = prop.multiProcessorCount;

  s->positions.assign(d->positions, d->positions + 3 * (size_t)d->vertex_count);
  s->indices.assign(d->indices, d->indices + d->index_count);
  s->primMeshes.assign(d->prim_meshes, d->prim_meshes + d->prim_mesh_count);
  s->nodes.assign(d->nodes, d->nodes + d->node_count);
  s->lightCount = d->light_count;
  s->materialCount = d->material_count;

  DevScene& D = s->dev;
  if((rc = upload(s, d->positions, 3 * (size_t)d->vertex_count, &D.positions)) != VKRT_OK) return bail(rc);
  if((rc = upload(s, d->indices, (size_t)d->index_count, &D.indices)) != VKRT_OK) return bail(rc);
  // (normals / uv travel inside vertexPN; the ePrimLookup indirection of hello_vulkan.cpp:363-368 is resolved per triangle
  // at build time into triShade, so neither is uploaded in its raw form)
  // interleaved (position, normal, uv, tangent) records for the closest-hit attribute fetch (rchit:41-66)
  {
    std::vector<float> pn((size_t)d->vertex_count * 4 * VKRT_VERTEX_QUADS);
    for(uint32_t v = 0; v < d->vertex_count; v++)
    {
      float* o = &pn[(size_t)v * 4 * VKRT_VERTEX_QUADS];
      o[0] = d->positions[3 * (size_t)v]; o[1] = d->positions[3 * (size_t)v + 1]; o[2] = d->positions[3 * (size_t)v + 2];
      o[3] = d->normals[3 * (size_t)v]; o[4] = d->normals[3 * (size_t)v + 1]; o[5] = d->normals[3 * (size_t)v + 2];
      o[6] = d->texcoords0[2 * (size_t)v]; o[7] = d->texcoords0[2 * (size_t)v + 1];
      for(int k = 0; k < 4; k++) o[8 + k] = d->tangents[4 * (size_t)v + k];
    }
    const float* pnDev = nullptr;
    if((rc = upload(s, pn.data(), pn.size(), &pnDev)) != VKRT_OK) return bail(rc);
    D.vertexPN = (const float4*)pnDev;
  }
  if((rc = upload(s, d->lights, (size_t)d->light_count, &D.lights)) != VKRT_OK) return bail(rc);
  // instances: object->world rows + inverse (gl_ObjectToWorldEXT / gl_WorldToObjectEXT, rchit:72-76)
  std::vector<DevInstance> inst(d->node_count);
  for(uint32_t n = 0; n < d->node_count; n++)
  {
    DevInstance& in = inst[n];
    memset(&in, 0, sizeof in);
    for(int r = 0; r < 3; r++)
      for(int c = 0; c < 4; c++)
        in.o2w[r * 4 + c] = d->nodes[n].worldMatrix[c * 4 + r];
    vkrt::invert3x3_rows(in.o2w, in.w2o);
    in.primMesh = d->nodes[n].primMesh;
  }
  if((rc = upload(s, inst.data(), inst.size(), &D.instances)) != VKRT_OK) return bail(rc);
  // textures: RGBA8 pool + table + sRGB decode table
  float lut[512];
  for(int i = 0; i < 256; i++)
  {
    const float c = (float)i / 255.0f;
    lut[i] = (c <= 0.04045f) ? c / 12.92f : powf((c + 0.055f) / 1.055f, 2.4f);
    lut[256 + i] = c;
  }
  std::vector<DevTexture> table(d->texture_count);
  std::vector<uint32_t> mipTable((size_t)d->texture_count * VKRT_MAX_MIPS, 0u);
  std::vector<uint32_t> pool;
  // footprint pool (DevScene::texQuads): 16 bytes per level-0 texel; built when it stays below 2 GiB and VKRT_TEX_QUADS=0 is not set
  // (test hook: the path tracer then gathers the four texels of a tap one by one, as it did up to round 3; same values either way)
  std::vector<uint32_t> quadFirst(d->texture_count, 0u);
  uint64_t quadTotal = 1;  // record 0 = white dummy (hits without a texture read it and discard it)
  for(uint32_t t = 0; t < d->texture_count; t++)
  {
    quadFirst[t] = (uint32_t)std::min<uint64_t>(quadTotal, 0xffffffffull);
    quadTotal += (uint64_t)d->textures[t].width * d->textures[t].height;
  }
  const char* quadEnv = getenv("VKRT_TEX_QUADS");
  const bool wantQuads = quadTotal * 16ull < (2ull << 30) && !(quadEnv && atoi(quadEnv) == 0);
  std::vector<uint32_t> quads;
  if(wantQuads)
  {
    quads.assign((size_t)quadTotal * 4, 0xffffffffu);
    for(uint32_t t = 0; t < d->texture_count; t++)
    {
      const vkrt_texture& tx = d->textures[t];
      const uint32_t* src = (const uint32_t*)tx.rgba8;
      uint32_t* dst = &quads[(size_t)quadFirst[t] * 4];
      for(uint32_t y = 0; y < tx.height; y++)
      {
        const uint32_t y1 = y + 1 == tx.height ? 0 : y + 1;
        for(uint32_t x = 0; x < tx.width; x++)
        {
          const uint32_t x1 = x + 1 == tx.width ? 0 : x + 1;
          uint32_t* q = dst + ((size_t)y * tx.width + x) * 4;
          memcpy(&q[0], &src[(size_t)y * tx.width + x], 4); memcpy(&q[1], &src[(size_t)y * tx.width + x1], 4);
          memcpy(&q[2], &src[(size_t)y1 * tx.width + x], 4); memcpy(&q[3], &src[(size_t)y1 * tx.width + x1], 4);
        }
      }
    }
  }
  for(uint32_t t = 0; t < d->texture_count; t++)
  {
    const vkrt_texture& tx = d->textures[t];
    const size_t n = (size_t)tx.width * tx.height;
    size_t at = pool.size();
    pool.resize(at + n);
    memcpy(&pool[at], tx.rgba8, n * 4);
    // the mip chain behind level 0 (sampled by the hybrid G-buffer's implicit-LOD texture(); the path tracer reads level 0)
    uint32_t levels = 1, w = tx.width, h = tx.height;
    mipTable[(size_t)t * VKRT_MAX_MIPS] = (uint32_t)at;
    while((w > 1 || h > 1) && levels < VKRT_MAX_MIPS)
    {
      const uint32_t dw = std::max(1u, w / 2), dh = std::max(1u, h / 2);
      const size_t to = pool.size();
      pool.resize(to + (size_t)dw * dh);
      downsampleLevel(&pool[at], w, h, &pool[to], dw, dh, tx.is_srgb != 0, lut);
      mipTable[(size_t)t * VKRT_MAX_MIPS + levels] = (uint32_t)to;
      at = to; w = dw; h = dh;
      levels++;
    }
    table[t] = DevTexture{mipTable[(size_t)t * VKRT_MAX_MIPS], tx.width, tx.height, (tx.is_srgb ? 1u : 0u) | (levels << 8)};
  }
  if(pool.empty())
    pool.push_back(0xffffffffu);  // shading always issues its texel loads (to texel 0 when a material has no texture)
  // materials, with the descriptors of their textures folded in (DevMaterial)
  std::vector<DevMaterial> mats(d->material_count);
  std::vector<DevShadeMaterial> shadeMats(d->material_count);
  for(uint32_t i = 0; i < d->material_count; i++)
  {
    memset(&mats[i], 0, sizeof(DevMaterial));
    mats[i].m = d->materials[i];
    DevShadeMaterial& sm = shadeMats[i];
    memset(&sm, 0, sizeof(sm));
    for(int k = 0; k < 3; k++)
    {
      sm.f[k] = d->materials[i].pbrBaseColorFactor[k];
      sm.f[5 + k] = d->materials[i].emissiveFactor[k];
    }
    sm.f[3] = d->materials[i].metallicFactor;
    sm.f[4] = d->materials[i].roughnessFactor;
    s->materialAlpha.push_back(d->materials[i].pbrBaseColorFactor[3]);
    const int idx[4] = {mats[i].m.pbrBaseColorTexture, mats[i].m.metallicRoughnessTexture, mats[i].m.normalTexture, mats[i].m.emissiveTexture};
    for(int k = 0; k < 4; k++)
    {
      DevTexRef& r = mats[i].tex[k];
      r = DevTexRef{0u, 1u | (1u << 16), 0u, 0u};
      if(idx[k] >= 0 && (uint32_t)idx[k] < d->texture_count)
      {
        const DevTexture& t = table[(size_t)idx[k]];
        if(t.width == 0u || t.height == 0u || t.width > 32768u || t.height > 32768u)
          return bail(fail(VKRT_ERR_UNSUPPORTED, "texture %d is %ux%u (supported: 1..32768 per side)", idx[k], t.width, t.height));
        r = DevTexRef{t.offset, t.width | (t.height << 16), 1u | ((t.srgb & 1u) ? 2u : 0u), wantQuads ? quadFirst[(size_t)idx[k]] : 0u};
      }
      // (the reference order of DevShadeMaterial is VKRT_TEXREF_*: BASE, MR, NORMAL, EMISSIVE = the order of idx[])
      const uint32_t base = wantQuads ? r.quads : r.offset;
      if(base >> 31)
        return bail(fail(VKRT_ERR_UNSUPPORTED, "texture pool of 2^31 records and more is not supported"));
      sm.ref[2 * k] = ((r.wh & 0xffffu) - 1u) | (idx[k] > -1 ? 1u << 15 : 0u) | (((r.wh >> 16) - 1u) << 16) | ((r.flags & 1u) << 31);
      sm.ref[2 * k + 1] = base | ((r.flags & 2u) << 30);
    }
  }
  if((rc = upload(s, mats.data(), mats.size(), &D.materials)) != VKRT_OK) return bail(rc);
  if((rc = upload(s, shadeMats.data(), shadeMats.size(), &D.shadeMaterials)) != VKRT_OK) return bail(rc);
  if((rc = upload(s, table.data(), table.size(), &D.textures)) != VKRT_OK) return bail(rc);
  if((rc = upload(s, mipTable.data(), mipTable.size(), &D.texMips)) != VKRT_OK) return bail(rc);
  if((rc = upload(s, pool.data(), pool.size(), &D.texels)) != VKRT_OK) return bail(rc);
  D.texQuads = nullptr;
  if(wantQuads)
  {
    const uint32_t* qd = nullptr;
    if((rc = upload(s, quads.data(), quads.size(), &qd)) != VKRT_OK) return bail(rc);
    D.texQuads = (const uint4*)qd;
  }
  // the hit shader addresses its tables with 32-bit byte offsets (shade.h BufView): refuse what does not fit
  if((uint64_t)pool.size() * 4 >= (4ull << 30) || (uint64_t)d->vertex_count * VKRT_VERTEX_BYTES >= (4ull << 30) || (uint64_t)d->material_count * 128 >= (4ull << 30) ||
     (uint64_t)d->node_count * 96 >= (4ull << 30))
    return bail(fail(VKRT_ERR_UNSUPPORTED, "scene tables of 4 GiB and more are not supported (texel pool %zu texels, %u vertices)", pool.size(), d->vertex_count));
  const float* lutDev = nullptr;
  if((rc = upload(s, lut, 512, &lutDev)) != VKRT_OK) return bail(rc);
  D.srgbLut = lutDev;
  D.textureCount = d->texture_count;
  D.rootRef = VKRT_TRAV_DONE;
  D.stackCap = 1;
  D.layout = 0;
  D.stepLimit = 64;
  D.triThreshold = 0;
  D.shareMinIdle = 0;
  D.sharePeriodMask = 0;
  D.shareFlags = 0;
  D.gbufferMips = 1;

  void* p = nullptr;
  if(hipMalloc(&p, 64) != hipSuccess) return bail(fail(VKRT_ERR_OUT_OF_MEMORY, "hipMalloc(work counter)"));
  s->allocs.push_back(p);
  s->workCounter = (unsigned int*)p;
  if(hipMalloc(&p, sizeof(DevCounters)) != hipSuccess) return bail(fail(VKRT_ERR_OUT_OF_MEMORY, "hipMalloc(counters)"));
  s->allocs.push_back(p);
  s->counters = (DevCounters*)p;
  D.faults = &s->counters->v[0][10];
  if(hipMemset(s->counters, 0, sizeof(DevCounters)) != hipSuccess) return bail(fail(VKRT_ERR_HIP, "hipMemset(counters)"));
  if(hipEventCreate(&s->evStart) != hipSuccess || hipEventCreate(&s->evStop) != hipSuccess)
    return bail(fail(VKRT_ERR_HIP, "hipEventCreate"));
  *out = s;
  return VKRT_OK;
}

void vkrt_scene_destroy(vkrt_scene* s)
{
  if(!s)
    return;
  (void)hipSetDevice(s->device);
  freeAccel(s);
  for(void* p : s->allocs)
    (void)hipFree(p);
  if(s->wfMem) (void)hipFree(s->wfMem);
  if(s->wfAsync.count)
  {
    (void)hipEventDestroy(s->wfAsync.fork);
    for(int j = 0; j < s->wfAsync.count; j++)
    {
      (void)hipStreamDestroy(s->wfAsync.streams[j]);
      (void)hipEventDestroy(s->wfAsync.join[j]);
    }
  }
  for(hipEvent_t e : s->wfEvents) (void)hipEventDestroy(e);
  for(hipEvent_t e : s->wfPool) (void)hipEventDestroy(e);
  if(s->evStart) (void)hipEventDestroy(s->evStart);
  if(s->evStop) (void)hipEventDestroy(s->evStop);
  delete s;
}

int vkrt_scene_set_option(vkrt_scene* s, int option, int value)
{
  if(!s)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "scene is NULL");
  if(option < VKRT_OPT_MODE || option > VKRT_OPT_LAST)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "unknown option %d", option);
  if(clampOption(option, value) != value)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "option %d: value %d out of range", option, value);
  s->opt[option] = value;
  return VKRT_OK;
}

int vkrt_scene_get_option(const vkrt_scene* s, int option, int* value)
{
  if(!s || !value)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "NULL argument");
  if(option == VKRT_INFO_ANYHIT_ORDER)
  {
    *value = s->built ? (int)(s->dev.shareFlags & 6u) : 0;
    return VKRT_OK;
  }
  if(option == VKRT_INFO_SPLIT_BUDGET)
  {
    *value = s->built ? s->splitResolved : 0;
    return VKRT_OK;
  }
  if(option < VKRT_OPT_MODE || option > VKRT_OPT_LAST)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "unknown option %d", option);
  *value = s->opt[option];
  return VKRT_OK;
}

int vkrt_reserve(vkrt_scene* s, const vkrt_shard* shard, void* hip_stream)
{
  if(!s)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "NULL argument");
  // frame groups: what a vkrt_pathtrace_frames call would keep in flight under the current options
  return vkrt_reserve_frames(s, shard, (uint32_t)std::max(1, s->opt[VKRT_OPT_WF_FRAMES_IN_FLIGHT]), hip_stream);
}

int vkrt_reserve_frames(vkrt_scene* s, const vkrt_shard* shard, uint32_t frames_per_call, void* hip_stream)
{
  if(!s || !shard)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "NULL argument");
  if(frames_per_call == 0)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "frames_per_call is 0");
  if(shard->full_width == 0 || shard->full_height == 0)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "empty launch size");
  if(shard->shard_count > 1 && (shard->strip_rows == 0 || shard->shard_index >= shard->shard_count))
    return fail(VKRT_ERR_INVALID_ARGUMENT, "bad shard");
  int rc = setDevice(s);
  if(rc != VKRT_OK)
    return rc;
  const uint64_t tiles = (uint64_t)((shard->full_width + 7) / 8) * ((vkrt_shard_rows(shard) + 7) / 8);
  if(tiles * 64 >= 0xFFFFFFFFull)
    return fail(VKRT_ERR_UNSUPPORTED, "launch too large");
  // the mode that counts is the one the acceleration structure was built for (vkrt_pathtrace and vkrt_hybrid_trace look at
  // s->wavefront, not at an option that may have changed since); before a build, the option is the best guess there is
  const bool wavefront = s->built ? s->wavefront : useWavefront(s);
  if(tiles == 0 || !wavefront)
    return VKRT_OK;  // the megakernel keeps its state in registers / LDS
  // frame groups: what a call of frames_per_call frames keeps in flight (the balanced share of the option, wavefront.hip)
  int rc2 = ensureWorkingSet(s, (uint32_t)tiles * 64u, framesInFlight(s, (int)std::min<uint32_t>(frames_per_call, VKRT_FRAMES_PER_BATCH)), (hipStream_t)hip_stream);
  if(rc2 == VKRT_OK)
    rc2 = ensureEventPool(s, (int)std::min<uint32_t>(frames_per_call, VKRT_FRAMES_PER_BATCH));
  return rc2;
}

static int accelBuildOnce(vkrt_scene* s, uint32_t flags, void* hip_stream);

// VKRT_OPT_SPLIT_BUDGET = -1: the library decides.  Triangle pre-splitting pays where large triangles are not aligned with the axes
// (+14 % ... +97 % on a rotated building) and costs 1-8 % elsewhere (profiles/r05_split_rotated.jsonl), and the SAH cost of the finished
// tree tells the two apart: a 30 % budget lowers it by 18-30 % in the first case and by at most 5 % -- or raises it -- in the second.
// So: build with a 30 % budget, build without, keep the split tree only when its cost is below 0.9 of the unsplit one (one more build
// in that case; device builds are ~13 ms each for 262 k triangles).  Pixels do not depend on the outcome.
int vkrt_accel_build(vkrt_scene* s, uint32_t flags, void* hip_stream)
{
  if(!s)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "scene is NULL");
  const bool deviceBuild = flags == 0 || (flags & (VKRT_BUILD_LBVH_GPU | VKRT_BUILD_PLOC_GPU)) != 0;
  if(s->opt[VKRT_OPT_SPLIT_BUDGET] >= 0 || !deviceBuild)
  {
    const int keep = s->opt[VKRT_OPT_SPLIT_BUDGET];
    if(keep < 0) s->opt[VKRT_OPT_SPLIT_BUDGET] = 0;  // (the host builder does not split)
    const int rc = accelBuildOnce(s, flags, hip_stream);
    s->splitResolved = s->opt[VKRT_OPT_SPLIT_BUDGET];
    s->opt[VKRT_OPT_SPLIT_BUDGET] = keep;
    return rc;
  }
  const auto t0 = std::chrono::steady_clock::now();
  auto once = [&](int budget, float* sah) {
    s->opt[VKRT_OPT_SPLIT_BUDGET] = budget;
    const int rc = accelBuildOnce(s, flags, hip_stream);
    if(rc == VKRT_OK && sah) *sah = s->info.sah_cost;
    return rc;
  };
  float sahSplit = 0.0f, sahPlain = 0.0f;
  int rc = once(30, &sahSplit);
  if(rc == VKRT_OK) rc = once(0, &sahPlain);
  int resolved = 0;
  if(rc == VKRT_OK && sahSplit < 0.9f * sahPlain)
  {
    rc = once(30, nullptr);
    resolved = 30;
  }
  s->opt[VKRT_OPT_SPLIT_BUDGET] = -1;
  s->splitResolved = resolved;
  if(rc == VKRT_OK)
    s->info.build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();  // all the builds it took
  return rc;
}

static int accelBuildOnce(vkrt_scene* s, uint32_t flags, void* hip_stream)
{
  if(!s)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "scene is NULL");
  if(flags == 0)
    flags = VKRT_BUILD_DEFAULT;
  const bool wantPloc = (flags & VKRT_BUILD_PLOC_GPU) != 0;
  const bool wantLbvh = (flags & VKRT_BUILD_LBVH_GPU) != 0 || wantPloc, wantSah = (flags & VKRT_BUILD_SAH_HOST) != 0;
  if(wantLbvh == wantSah || (wantPloc && (flags & VKRT_BUILD_LBVH_GPU) != 0) || (flags & ~7u) != 0)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "build_flags must select exactly one of VKRT_BUILD_LBVH_GPU / VKRT_BUILD_PLOC_GPU / VKRT_BUILD_SAH_HOST");
  int rc = setDevice(s);
  if(rc != VKRT_OK)
    return rc;
  hipStream_t stream = (hipStream_t)hip_stream;
  HIP_TRY(hipStreamSynchronize(stream));  // no trace may be reading the old tree
  freeAccel(s);
  const auto t0 = std::chrono::steady_clock::now();
  s->info = vkrt_accel_info{};
  s->info.build_flags = wantPloc ? VKRT_BUILD_PLOC_GPU : wantLbvh ? VKRT_BUILD_LBVH_GPU : VKRT_BUILD_SAH_HOST;
  s->wavefront = useWavefront(s);
  const bool watertight = s->opt[VKRT_OPT_WATERTIGHT] != 0;
  if((watertight || s->opt[VKRT_OPT_ANYHIT_DISSOLVE] != 0) && useWavefront(s) && s->opt[VKRT_OPT_WF_TRAV_BLOCK] != 64)
    return fail(VKRT_ERR_UNSUPPORTED, "VKRT_OPT_WATERTIGHT / VKRT_OPT_ANYHIT_DISSOLVE are built for the default 64-thread traversal workgroups "
                "(VKRT_OPT_WF_TRAV_BLOCK = %d)", s->opt[VKRT_OPT_WF_TRAV_BLOCK]);
  // per instance: is its material non-opaque (dissolve = pbrBaseColorFactor.a < 1)?  The stage is compiled into the traversal only when
  // the scene has such an instance: without one no record carries the flag, the stage could never ignore a hit, and the flag test
  // and seed load per ray would cost 3.8 % for nothing (profiles/r03_options/ANYHIT_DISSOLVE.json)
  std::vector<uint8_t> instDissolves;
  bool dissolve = false;
  if(s->opt[VKRT_OPT_ANYHIT_DISSOLVE] != 0)
    for(const vkrt_node& n : s->nodes)
    {
      const int32_t m = std::max(0, s->primMeshes[(size_t)n.primMesh].materialIndex);
      const bool nonOpaque = s->materialAlpha[(size_t)m] < 1.0f && s->primMeshes[(size_t)n.primMesh].indexCount >= 3u;
      instDissolves.push_back(nonOpaque ? 1 : 0);
      dissolve = dissolve || nonOpaque;
    }
  s->dev.watertight = watertight ? 1u : 0u;
  s->dev.dissolve = dissolve ? 1u : 0u;
  const std::vector<uint8_t>* dissolvePtr = dissolve ? &instDissolves : nullptr;

  if(wantSah)
  {
    std::vector<vkrt::FlatTri> tris;
    vkrt::flatten_instances(s->positions.data(), s->indices.data(), s->primMeshes.data(), s->nodes.data(),
                            (uint32_t)s->nodes.size(), tris);
    // wide8 (compressed 8-wide) is the trace-optimised layout; the megakernel and VKRT_OPT_BVH_LAYOUT = 0 keep BVH2
    const bool wide = useWavefront(s) && s->opt[VKRT_OPT_BVH_LAYOUT] == 1;
    std::vector<float> packed;
    const void* nodeData = nullptr;
    size_t nodeBytesUsed = 0;
    vkrt::BuiltBvh bvh;
    vkrt::BuiltWide8 w8;
    if(wide)
    {
      vkrt::build_wide8_host(tris, w8, watertight);
      vkrt::pack_triangles(tris, w8.triOrder, packed, watertight, dissolvePtr);
      nodeData = w8.nodes.data();
      nodeBytesUsed = w8.nodes.size() * sizeof(uint32_t);
    }
    else
    {
      vkrt::build_sah_host(tris, 4, bvh, watertight);
      vkrt::pack_triangles(tris, bvh.triOrder, packed, watertight, dissolvePtr);
      nodeData = bvh.nodes.data();
      nodeBytesUsed = bvh.nodes.size() * sizeof(float);
    }
    std::vector<uint32_t> shadeRec;
    vkrt::pack_tri_shade(tris, wide ? w8.triOrder : bvh.triOrder, s->indices.data(), s->primMeshes.data(), s->nodes.data(), shadeRec);
    HIP_TRY(hipMalloc(&s->accelShade, std::max<size_t>(shadeRec.size() * 4, 16)));
    if(!shadeRec.empty())
      HIP_TRY(hipMemcpyAsync(s->accelShade, shadeRec.data(), shadeRec.size() * 4, hipMemcpyHostToDevice, stream));
    s->dev.triShade = (const uint4*)s->accelShade;
    const size_t nodeBytes = std::max<size_t>(nodeBytesUsed, VKRT_WNODE_MIN_ALLOC);
    const size_t triBytes = std::max<size_t>(packed.size() * sizeof(float), 48);
    HIP_TRY(hipMalloc(&s->accelNodes, nodeBytes));
    HIP_TRY(hipMalloc(&s->accelTris, triBytes));
    if(nodeBytesUsed)
      HIP_TRY(hipMemcpyAsync(s->accelNodes, nodeData, nodeBytesUsed, hipMemcpyHostToDevice, stream));
    if(!packed.empty())
      HIP_TRY(hipMemcpyAsync(s->accelTris, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    s->dev.nodes = (const float4*)s->accelNodes;
    s->dev.tris = (const float4*)s->accelTris;
    s->dev.triCount = (uint32_t)tris.size();
    s->info.triangle_count = (uint32_t)tris.size();
    s->info.node_bytes = nodeBytesUsed;
    s->info.triangle_bytes = packed.size() * sizeof(float);
    if(wide)
    {
      s->dev.layout = 1;
      s->dev.rootRef = tris.empty() ? VKRT_TRAV_DONE : 0;
      s->dev.stackCap = 2 * (w8.maxDepth + 1);  // at most one pending group per level (uint2 entries = 2 words)
      s->info.node_count = w8.nodeCount;
      s->info.max_depth = w8.maxDepth;
      s->info.sah_cost = w8.sahCost;
    }
    else
    {
      s->dev.layout = 0;
      s->dev.rootRef = bvh.rootRef;
      s->dev.stackCap = bvh.maxDepth + 2;
      s->info.node_count = (uint32_t)(bvh.nodes.size() / 16);
      s->info.max_depth = bvh.maxDepth;
      s->info.sah_cost = bvh.sahCost;
    }
  }
  else
  {
    const bool wide = useWavefront(s) && s->opt[VKRT_OPT_BVH_LAYOUT] == 1;
    vkrt::LbvhResult r;
    // GPU radix-tree build (Morton codes, sort, Karras hierarchy, bottom-up fit).  For the trace-optimised layout the
    // binary tree keeps one triangle per leaf and is collapsed into wide8 nodes by the same SAH-optimal DP as the SAH
    // path -- on the device too (wide_collapse.hip); nothing but four statistics words comes back to the host.
    // VKRT_BUILD_PLOC_GPU: same pipeline with the radix tree replaced by locally-ordered clustering (ploc.hip)
    rc = vkrt::build_lbvh_device(s->dev, (uint32_t)s->nodes.size(), s->primMeshes, s->nodes, stream, r, wide ? 1u : 4u, wide, wantPloc, watertight, dissolve,
                                 (unsigned)s->opt[VKRT_OPT_SPLIT_BUDGET]);
    if(rc != VKRT_OK)
      return fail(rc, "%s build failed: %s", wantPloc ? "PLOC" : "LBVH", r.error.c_str());
    s->info.triangle_count = r.uniqueTris;
    s->dev.triCount = r.triCount;  // slots: a pre-split triangle occupies one per reference
    if(wide && r.hasWide)
    {
      (void)hipFree(r.nodes); (void)hipFree(r.tris); (void)hipFree(r.triShade);
      s->accelNodes = r.wide.nodes;
      s->accelTris = r.wide.tris;
      s->accelShade = r.wide.triShade;
      s->dev.nodes = (const float4*)s->accelNodes;
      s->dev.tris = (const float4*)s->accelTris;
      s->dev.triShade = (const uint4*)s->accelShade;
      s->dev.layout = 1;
      s->dev.rootRef = 0;
      s->dev.stackCap = 2 * (r.wide.maxDepth + 1);  // at most one pending group per level (uint2 entries = 2 words)
      s->info.node_count = r.wide.nodeCount;
      s->info.max_depth = r.wide.maxDepth;
      s->info.sah_cost = r.wide.sahCost;
      s->info.node_bytes = (uint64_t)r.wide.nodeCount * VKRT_WNODE_BYTES;
      s->info.triangle_bytes = (uint64_t)r.triCount * 48;
    }
    else if(wide && r.triCount > 0)
    {
      // fallback (a single triangle, or a radix tree too deep for the device collapse's level budget): download the binary
      // tree (device layout == host layout of BuiltBvh) and the sorted triangle records, collapse on the host
      vkrt::BuiltBvh b2;
      b2.nodes.resize((size_t)r.nodeCount * 16);
      std::vector<float> trisHost((size_t)r.triCount * 12);
      hipError_t e = hipSuccess;
      if(r.nodeCount) e = hipMemcpy(b2.nodes.data(), r.nodes, b2.nodes.size() * 4, hipMemcpyDeviceToHost);
      if(e == hipSuccess) e = hipMemcpy(trisHost.data(), r.tris, trisHost.size() * 4, hipMemcpyDeviceToHost);
      (void)hipFree(r.nodes); (void)hipFree(r.tris); (void)hipFree(r.triShade);
      if(e != hipSuccess)
        return fail(VKRT_ERR_HIP, "LBVH download: %s", hipGetErrorString(e));
      b2.rootRef = r.rootRef;
      b2.maxDepth = r.maxDepth;
      b2.triOrder.resize(r.triCount);
      std::vector<vkrt::FlatTri> tris(r.triCount);
      for(uint32_t k = 0; k < r.triCount; k++)
      {
        b2.triOrder[k] = k;
        const float* t = &trisHost[(size_t)k * 12];
        vkrt::FlatTri& ft = tris[k];
        for(int c = 0; c < 3; c++)
        {
          ft.v0[c] = t[c];
          // records hold (v0, e1, e2) or, watertight, (p0, p1, p2): the other form is re-derived (p1 = v0 + e1 is not the exact vertex,
          // but the boxes below only grow by it)
          ft.e1[c] = watertight ? t[3 + c] - t[c] : t[3 + c];
          ft.e2[c] = watertight ? t[6 + c] - t[c] : t[6 + c];
          ft.p1[c] = watertight ? t[3 + c] : t[c] + t[3 + c];
          ft.p2[c] = watertight ? t[6 + c] : t[c] + t[6 + c];
        }
        memcpy(&ft.gid, &t[9], 4); memcpy(&ft.inst, &t[10], 4); memcpy(&ft.prim, &t[11], 4);
        ft.gid &= 0x7fffffffu;  // (bit 31 = the any-hit stage's flag; pack_triangles sets it again)
      }
      vkrt::BuiltWide8 w8;
      vkrt::collapse_wide8(b2, tris, w8, watertight);
      std::vector<float> packed;
      std::vector<uint32_t> shadeRec;
      vkrt::pack_triangles(tris, w8.triOrder, packed, watertight, dissolvePtr);
      vkrt::pack_tri_shade(tris, w8.triOrder, s->indices.data(), s->primMeshes.data(), s->nodes.data(), shadeRec);
      HIP_TRY(hipMalloc(&s->accelNodes, std::max<size_t>(w8.nodes.size() * 4, VKRT_WNODE_MIN_ALLOC)));
      HIP_TRY(hipMalloc(&s->accelTris, std::max<size_t>(packed.size() * 4, 48)));
      HIP_TRY(hipMalloc(&s->accelShade, std::max<size_t>(shadeRec.size() * 4, 16)));
      HIP_TRY(hipMemcpy(s->accelNodes, w8.nodes.data(), w8.nodes.size() * 4, hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(s->accelTris, packed.data(), packed.size() * 4, hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(s->accelShade, shadeRec.data(), shadeRec.size() * 4, hipMemcpyHostToDevice));
      s->dev.nodes = (const float4*)s->accelNodes;
      s->dev.tris = (const float4*)s->accelTris;
      s->dev.triShade = (const uint4*)s->accelShade;
      s->dev.layout = 1;
      s->dev.rootRef = 0;
      s->dev.stackCap = 2 * (w8.maxDepth + 1);  // at most one pending group per level (uint2 entries = 2 words)
      s->info.node_count = w8.nodeCount;
      s->info.max_depth = w8.maxDepth;
      s->info.sah_cost = w8.sahCost;
      s->info.node_bytes = w8.nodes.size() * 4;
      s->info.triangle_bytes = packed.size() * 4;
    }
    else
    {
      s->accelNodes = r.nodes;
      s->accelTris = r.tris;
      s->accelShade = r.triShade;
      s->dev.triShade = (const uint4*)r.triShade;
      s->dev.nodes = (const float4*)r.nodes;
      s->dev.tris = (const float4*)r.tris;
      s->dev.rootRef = r.rootRef;
      s->dev.layout = 0;
      s->dev.stackCap = r.maxDepth + 2;
      s->info.node_count = r.nodeCount;
      s->info.max_depth = r.maxDepth;
      s->info.sah_cost = r.sahCost;
      s->info.node_bytes = (uint64_t)r.nodeCount * 64;
      s->info.triangle_bytes = (uint64_t)r.triCount * 48;
    }
  }
  // world-space bounds of the instanced geometry, conservatively from the corners of every node's local box (the any-hit order
  // heuristic asks whether a ray ends outside them; nothing else reads them)
  {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    std::vector<float> meshBox(s->primMeshes.size() * 6);
    for(size_t m = 0; m < s->primMeshes.size(); m++)
    {
      const vkrt_prim_mesh& pm = s->primMeshes[m];
      float* b = &meshBox[m * 6];
      for(int k = 0; k < 3; k++) { b[k] = INFINITY; b[3 + k] = -INFINITY; }
      for(uint32_t v = 0; v < pm.vertexCount; v++)
        for(int k = 0; k < 3; k++)
        {
          const float x = s->positions[3 * (size_t)(pm.vertexOffset + v) + k];
          b[k] = std::min(b[k], x); b[3 + k] = std::max(b[3 + k], x);
        }
    }
    for(const vkrt_node& n : s->nodes)
    {
      const float* b = &meshBox[(size_t)n.primMesh * 6];
      if(!(b[0] <= b[3]))
        continue;
      for(int c = 0; c < 8; c++)
      {
        const float p[3] = {(c & 1) ? b[3] : b[0], (c & 2) ? b[4] : b[1], (c & 4) ? b[5] : b[2]};
        for(int k = 0; k < 3; k++)
        {
          const float w = n.worldMatrix[k] * p[0] + n.worldMatrix[4 + k] * p[1] + n.worldMatrix[8 + k] * p[2] + n.worldMatrix[12 + k];
          lo[k] = std::min(lo[k], w); hi[k] = std::max(hi[k], w);
        }
      }
    }
    for(int k = 0; k < 3; k++)
    {
      const float pad = 1e-3f * std::max(1e-6f, hi[k] - lo[k]);
      s->dev.sceneLo[k] = lo[k] - pad; s->dev.sceneHi[k] = hi[k] + pad;
    }
    // "large" = a triangle of more than 1 % of the largest face of the scene's box (world space, every instance)
    const double ex = (double)hi[0] - lo[0], ey = (double)hi[1] - lo[1], ez = (double)hi[2] - lo[2];
    const double face = std::max(ex * ey, std::max(ey * ez, ez * ex));
    double maxArea = 0.0;
    for(const vkrt_node& n : s->nodes)
    {
      const vkrt_prim_mesh& pm = s->primMeshes[(size_t)n.primMesh];
      const float* M = n.worldMatrix;
      for(uint32_t t = 0; t + 3 <= pm.indexCount; t += 3)
      {
        double w[3][3];
        for(int v = 0; v < 3; v++)
        {
          const float* p = &s->positions[3 * (size_t)(s->indices[pm.firstIndex + t + v] + pm.vertexOffset)];
          for(int k = 0; k < 3; k++) w[v][k] = (double)M[k] * p[0] + (double)M[4 + k] * p[1] + (double)M[8 + k] * p[2] + (double)M[12 + k];
        }
        const double a[3] = {w[1][0] - w[0][0], w[1][1] - w[0][1], w[1][2] - w[0][2]}, b[3] = {w[2][0] - w[0][0], w[2][1] - w[0][1], w[2][2] - w[0][2]};
        const double cx = a[1] * b[2] - a[2] * b[1], cy = a[2] * b[0] - a[0] * b[2], cz = a[0] * b[1] - a[1] * b[0];
        maxArea = std::max(maxArea, 0.5 * std::sqrt(cx * cx + cy * cy + cz * cz));
      }
    }
    s->hasLargeTriangles = face > 0.0 && maxArea > 0.01 * face;
  }
  s->info.reference_count = s->dev.triCount;
  s->dev.stepLimit = 4u * (s->info.node_count + s->dev.triCount) + 64u;
  s->dev.triThreshold = 0;
  s->dev.shareMinIdle = 0;
  s->dev.sharePeriodMask = 0;
  // order of any-hit walks (bits 1, 2: every layout and mode; traverse.h anyhit_far_first).  Bit 3 = automatic: far-first for rays that
  // end outside the scene bounds unless the scene has LARGE triangles -- room-sized polygons sit in the leaves of the top nodes and
  // stop such rays within a step or two of a front-to-back walk, which the far-first order then only delays (measured on the two
  // tessellations of the atrium, profiles/r03_experiments.md #94: +3.3 % on the uniform one, -2 % on the Sponza-like one).  Round 5: that
  // holds while the large triangles enter the tree WHOLE; pre-split into references (VKRT_OPT_SPLIT_BUDGET) they are spread over the
  // tree like any other geometry and far-first wins again -- the Sponza-like building rotated 20 / 45 / 35+20 degrees with the automatic
  // budget: -2.2 % / -2.9 % / -2.0 % traversal time, the rotated uniform one -2.5 % ... -3.1 % (profiles/r05_experiments.md #144)
  s->dev.shareFlags = (uint32_t)s->opt[VKRT_OPT_WF_SHARE_FLAGS] & 6u;
  const bool largeWhole = s->hasLargeTriangles && s->info.reference_count == s->info.triangle_count;
  if((s->opt[VKRT_OPT_WF_SHARE_FLAGS] & 8) && !largeWhole)
    s->dev.shareFlags |= 4u;
  if(s->dev.layout == 1)
  {
    // triangle postponing (traverse_wide.h): lanes with pending triangles before a wave tests them; 0 = immediate
    s->dev.triThreshold = (uint32_t)s->opt[VKRT_OPT_TRI_THRESHOLD];
    if(s->dev.triThreshold != 0u)
      s->dev.stackCap += 2 * VKRT_W8_POSTPONE_ROOM;  // room for parked triangle groups (uint2 entries)
    // work sharing inside a traversal wave (traverse_share.h): minimum number of idle lanes before they take over subtrees
    s->dev.shareMinIdle = (uint32_t)s->opt[VKRT_OPT_WF_SHARE];
    s->dev.sharePeriodMask = (uint32_t)s->opt[VKRT_OPT_WF_SHARE_PERIOD];
    s->dev.shareFlags |= (uint32_t)s->opt[VKRT_OPT_WF_SHARE_FLAGS] & 17u;  // bit 0: child donation, bit 4: triangle-group donation
  }
  // LDS budget: stackCap * 256 lanes * 4 B must fit a workgroup (160 KiB per CU on gfx950)
  if((size_t)s->dev.stackCap * 256 * 4 > 64 * 1024)
    return fail(VKRT_ERR_UNSUPPORTED, "BVH depth %u needs a %zu-byte LDS stack per workgroup (limit 64 KiB)", s->info.max_depth,
                (size_t)s->dev.stackCap * 256 * 4);
  s->info.build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  s->built = true;
  return VKRT_OK;
}

int vkrt_accel_get_info(const vkrt_scene* s, vkrt_accel_info* out)
{
  if(!s || !out)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "NULL argument");
  if(!s->built)
    return fail(VKRT_ERR_NOT_BUILT, "vkrt_accel_build has not been called");
  *out = s->info;
  return VKRT_OK;
}

uint32_t vkrt_shard_rows(const vkrt_shard* sh)
{
  if(!sh)
    return 0;
  if(sh->strip_rows == 0 || sh->shard_count <= 1)
    return (sh->shard_count <= 1 || sh->shard_index == 0) ? sh->full_height : 0;
  const uint32_t strips = (sh->full_height + sh->strip_rows - 1) / sh->strip_rows;
  uint32_t rows = 0;
  for(uint32_t s = sh->shard_index; s < strips; s += sh->shard_count)
  {
    const uint32_t y0 = s * sh->strip_rows;
    rows += std::min(sh->strip_rows, sh->full_height - y0);
  }
  return rows;
}

int vkrt_pathtrace(vkrt_scene* s, const PushConstantRay* pc, const GlobalUniforms* cam, const vkrt_trace_opts* opts,
                   const vkrt_shard* shard, float* image, void* hip_stream)
{
  return vkrt_pathtrace_frames(s, pc, cam, opts, shard, image, 1u, hip_stream);
}

int vkrt_pathtrace_frames(vkrt_scene* s, const PushConstantRay* pc, const GlobalUniforms* cam, const vkrt_trace_opts* opts,
                          const vkrt_shard* shard, float* image, uint32_t n_frames, void* hip_stream)
{
  if(!s || !pc || !cam || !shard)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "NULL argument");
  if(n_frames == 0u)
    return VKRT_OK;
  if(n_frames > 65536u || (int64_t)pc->frame + (int64_t)n_frames > 0x7fffffffll)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "n_frames %u out of range", n_frames);
  if(!image && vkrt_shard_rows(shard) != 0u)  // (a shard without rows -- more ranks than strips -- has no image to pass)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "NULL image");
  if(!s->built)
    return fail(VKRT_ERR_NOT_BUILT, "vkrt_pathtrace before vkrt_accel_build");
  if(shard->full_width == 0 || shard->full_height == 0)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "empty launch size");
  if(shard->shard_count > 1 && (shard->strip_rows == 0 || shard->shard_index >= shard->shard_count))
    return fail(VKRT_ERR_INVALID_ARGUMENT, "bad shard (strip_rows %u, %u of %u)", shard->strip_rows, shard->shard_index,
                shard->shard_count);
  if(pc->lightsCount < 0 || (uint32_t)pc->lightsCount > s->lightCount)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "PushConstantRay.lightsCount %d outside [0,%u]", pc->lightsCount, s->lightCount);
  if(pc->samples < 0 || pc->depth < 0 || pc->depth > 99)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "samples/depth out of range");
  int rc = setDevice(s);
  if(rc != VKRT_OK)
    return rc;
  hipStream_t stream = (hipStream_t)hip_stream;
  TraceParams P;
  memset(&P, 0, sizeof P);
  P.sc = s->dev;
  P.pc = *pc;
  memcpy(P.viewInverse, cam->viewInverse.m, sizeof P.viewInverse);
  memcpy(P.projInverse, cam->projInverse.m, sizeof P.projInverse);
  P.seed = opts ? opts->seed : 0u;
  P.flags = (opts ? opts->flags : 0u) & VKRT_TRACE_PUBLIC_FLAGS;
  if(s->opt[VKRT_OPT_SKIP_DEAD_SHADOW_RAYS])
    P.flags |= VKRT_FLAG_SKIP_DEAD_SHADOW;  // internal bit (device_scene.h)
  P.fullW = shard->full_width;
  P.fullH = shard->full_height;
  const bool sharded = shard->shard_count > 1;
  P.stripRows = sharded ? shard->strip_rows : 0u;
  P.shardCount = sharded ? shard->shard_count : 1u;
  P.shardIndex = sharded ? shard->shard_index : 0u;
  P.localRows = vkrt_shard_rows(shard);
  P.image = image;
  P.workCounter = s->workCounter;
  P.counters = s->counters;
  if(P.localRows == 0)
    return VKRT_OK;
  P.tilesX = (P.fullW + 7) / 8;
  const uint64_t tiles = (uint64_t)P.tilesX * ((P.localRows + 7) / 8);
  if(tiles * 64 >= 0xFFFFFFFFull)
    return fail(VKRT_ERR_UNSUPPORTED, "launch too large");
  P.tileCount = (uint32_t)tiles;
  P.tileFirst = 0;

  const bool count = (P.flags & VKRT_TRACE_COUNT_TRAVERSAL) != 0;
  if(s->wavefront)
  {
    if(P.fullW > 65535u || P.localRows > 65535u || pc->samples > 65535)
      return fail(VKRT_ERR_UNSUPPORTED, "wavefront mode packs pixel coordinates / sample index in 16 bits");
    const uint32_t seedStep = (P.flags & VKRT_TRACE_SAME_SEED_EVERY_FRAME) ? 0u : 1u;
    if((rc = ensureWorkingSet(s, P.tileCount * 64u, framesInFlight(s, (int)n_frames), stream)) != VKRT_OK)
      return rc;
    if((rc = ensureEventPool(s, (int)std::min<uint32_t>(n_frames, VKRT_FRAMES_PER_BATCH))) != VKRT_OK)
      return rc;
    WfTiming* timing = nullptr;
    if(P.flags & VKRT_TRACE_TIME_KERNELS)
    {
      const size_t wantEv = (size_t)2 * (2 * (size_t)pc->samples * pc->depth) * std::min<uint32_t>(n_frames, 8u);
      while(s->wfEvents.size() < wantEv && s->wfEvents.size() < 8192)
      {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        s->wfEvents.push_back(e);
      }
      s->wfTiming.events = s->wfEvents.data();
      s->wfTiming.capacity = (int)s->wfEvents.size();
      timing = &s->wfTiming;
    }
    s->wfTimed = timing != nullptr;
    s->wfTimingRounds = (pc->samples > 0 && pc->depth > 0) ? pc->samples * (pc->depth + 1) : 0;
    HIP_TRY(hipEventRecord(s->evStart, stream));
    WfOptions wo;
    wo.subframes = s->opt[VKRT_OPT_WF_SUBFRAMES];
    wo.travBlock = (s->dev.watertight || s->dev.dissolve) ? 64 : s->opt[VKRT_OPT_WF_TRAV_BLOCK];  // (the non-default triangle modes exist for the default workgroup only)
    wo.inFlight = framesInFlight(s, (int)n_frames);
    if(timing)
      timing->used = 0;  // one timing record per call: the batches of a long call append to it
    for(uint32_t first = 0; first < n_frames; first += VKRT_FRAMES_PER_BATCH)
    {
      TraceParams Pb = P;
      Pb.pc.frame = P.pc.frame + (int)first;
      Pb.seed = P.seed + first * seedStep;
      HIP_TRY(vkrt_launch_wavefront(Pb, s->wf, wo, (int)std::min<uint32_t>(VKRT_FRAMES_PER_BATCH, n_frames - first), seedStep, count, stream, timing, &s->wfAsync));
    }
    HIP_TRY(hipEventRecord(s->evStop, stream));
    s->timed = true;
    return VKRT_OK;
  }
  const size_t lds = (size_t)P.sc.stackCap * 256 * sizeof(int);
  int perCU = 0;
  HIP_TRY(vkrt_pathtrace_occupancy(lds, &perCU));
  if(perCU < 1)
    return fail(VKRT_ERR_UNSUPPORTED, "path-trace kernel does not fit a CU with %zu B of LDS", lds);
  const uint64_t wantBlocks = (tiles * 64 + 255) / 256;
  const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(wantBlocks, (uint64_t)s->cuCount * perCU));

  HIP_TRY(hipEventRecord(s->evStart, stream));
  for(uint32_t k = 0; k < n_frames; k++)  // the megakernel renders the frames of a call one after another
  {
    TraceParams Pk = P;
    Pk.pc.frame = P.pc.frame + (int)k;
    Pk.seed = P.seed + ((P.flags & VKRT_TRACE_SAME_SEED_EVERY_FRAME) ? 0u : k);
    HIP_TRY(hipMemsetAsync(s->workCounter, 0, sizeof(unsigned int), stream));
    HIP_TRY(vkrt_launch_pathtrace(Pk, grid, count, stream));
  }
  HIP_TRY(hipEventRecord(s->evStop, stream));
  s->timed = true;
  s->wfTimed = false;
  return VKRT_OK;
}

namespace {
// launch geometry shared by the hybrid entry points (same shard semantics as vkrt_pathtrace)
int fillParams(vkrt_scene* s, const PushConstantRay* pc, const GlobalUniforms* cam, const vkrt_trace_opts* opts, const vkrt_shard* shard,
               TraceParams& P)
{
  if(!s->built)
    return fail(VKRT_ERR_NOT_BUILT, "acceleration structure not built");
  if(shard->full_width == 0 || shard->full_height == 0)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "empty launch size");
  if(shard->shard_count > 1 && (shard->strip_rows == 0 || shard->shard_index >= shard->shard_count))
    return fail(VKRT_ERR_INVALID_ARGUMENT, "bad shard");
  memset(&P, 0, sizeof P);
  P.sc = s->dev;
  if(pc) P.pc = *pc;
  memcpy(P.viewInverse, cam->viewInverse.m, sizeof P.viewInverse);
  memcpy(P.projInverse, cam->projInverse.m, sizeof P.projInverse);
  P.seed = opts ? opts->seed : 0u;
  P.flags = (opts ? opts->flags : 0u) & VKRT_TRACE_PUBLIC_FLAGS;  // (the hybrid passes never skip shadow rays: hitDists needs their result)
  P.fullW = shard->full_width;
  P.fullH = shard->full_height;
  const bool sharded = shard->shard_count > 1;
  P.stripRows = sharded ? shard->strip_rows : 0u;
  P.shardCount = sharded ? shard->shard_count : 1u;
  P.shardIndex = sharded ? shard->shard_index : 0u;
  P.localRows = vkrt_shard_rows(shard);
  P.counters = s->counters;
  P.tilesX = (P.fullW + 7) / 8;
  const uint64_t tiles = (uint64_t)P.tilesX * ((P.localRows + 7) / 8);
  if(tiles * 64 >= 0xFFFFFFFFull)
    return fail(VKRT_ERR_UNSUPPORTED, "launch too large");
  P.tileCount = (uint32_t)tiles;
  P.tileFirst = 0;
  return VKRT_OK;
}
}  // namespace

namespace {
int gbufferImpl(vkrt_scene* s, const float clearColor[4], int lightsCount, const GlobalUniforms* cam, const float* viewMatrix, const vkrt_shard* shard,
                const vkrt_gbuffer* out, const vkrt_nrd_planes* nrd, void* hip_stream);
int hybridImpl(vkrt_scene* s, const PushConstantRay* pc, const GlobalUniforms* cam, const vkrt_trace_opts* opts, const vkrt_shard* shard,
               const vkrt_gbuffer* g, const vkrt_nrd_planes* nrd, float* accum, void* hip_stream);
}  // namespace

int vkrt_gbuffer_raycast(vkrt_scene* s, const float clearColor[4], int lightsCount, const GlobalUniforms* cam, const vkrt_shard* shard,
                         const vkrt_gbuffer* out, void* hip_stream)
{
  return gbufferImpl(s, clearColor, lightsCount, cam, nullptr, shard, out, nullptr, hip_stream);
}

int vkrt_gbuffer_raycast_nrd(vkrt_scene* s, const float clearColor[4], int lightsCount, const GlobalUniforms* cam, const float viewMatrix[16],
                             const vkrt_shard* shard, const vkrt_gbuffer* out, const vkrt_nrd_planes* nrd, void* hip_stream)
{
  if(s && clearColor && cam && shard && out && nrd && vkrt_shard_rows(shard) == 0u)
    return VKRT_OK;  // a shard without rows
  if(!nrd || !viewMatrix || !nrd->normalRoughness || !nrd->viewZ || !nrd->diffRadianceHitDist)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "NULL NRD plane / view matrix");
  return gbufferImpl(s, clearColor, lightsCount, cam, viewMatrix, shard, out, nrd, hip_stream);
}

int vkrt_hybrid_trace(vkrt_scene* s, const PushConstantRay* pc, const GlobalUniforms* cam, const vkrt_trace_opts* opts, const vkrt_shard* shard,
                      const vkrt_gbuffer* g, float* accum, void* hip_stream)
{
  return hybridImpl(s, pc, cam, opts, shard, g, nullptr, accum, hip_stream);
}

int vkrt_hybrid_trace_nrd(vkrt_scene* s, const PushConstantRay* pc, const GlobalUniforms* cam, const vkrt_trace_opts* opts, const vkrt_shard* shard,
                          const vkrt_gbuffer* g, const vkrt_nrd_planes* nrd, float* accum, void* hip_stream)
{
  if(s && pc && cam && shard && g && nrd && vkrt_shard_rows(shard) == 0u)
    return VKRT_OK;  // a shard without rows
  if(!nrd || !nrd->viewZ || !nrd->diffRadianceHitDist)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "NULL NRD plane");
  return hybridImpl(s, pc, cam, opts, shard, g, nrd, accum, hip_stream);
}

namespace {
int gbufferImpl(vkrt_scene* s, const float clearColor[4], int lightsCount, const GlobalUniforms* cam, const float* viewMatrix, const vkrt_shard* shard,
                const vkrt_gbuffer* out, const vkrt_nrd_planes* nrd, void* hip_stream)
{
  if(s && clearColor && cam && shard && out && vkrt_shard_rows(shard) == 0u)
    return VKRT_OK;  // a shard without rows: nothing to write, no planes to pass
  if(!s || !clearColor || !cam || !shard || !out || !out->color || !out->position || !out->normal || !out->roughMetal)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "NULL argument");
  if(lightsCount < 0 || (uint32_t)lightsCount > s->lightCount)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "lightsCount %d outside [0,%u]", lightsCount, s->lightCount);
  int rc = setDevice(s);
  if(rc != VKRT_OK)
    return rc;
  TraceParams P;
  if((rc = fillParams(s, nullptr, cam, nullptr, shard, P)) != VKRT_OK)
    return rc;
  if(P.localRows == 0)
    return VKRT_OK;
  P.sc.gbufferMips = (uint32_t)s->opt[VKRT_OPT_GBUFFER_MIPS];
  NrdPlanes np{};
  if(nrd)
  {
    np.normRough = nrd->normalRoughness; np.viewZ = nrd->viewZ; np.radHitD = nrd->diffRadianceHitDist;
    memcpy(np.viewMatrix, viewMatrix, sizeof np.viewMatrix);
  }
  HIP_TRY(vkrt_launch_gbuffer(P, clearColor, lightsCount, out->color, out->position, out->normal, out->roughMetal, nrd ? &np : nullptr,
                              (hipStream_t)hip_stream));
  return VKRT_OK;
}

int hybridImpl(vkrt_scene* s, const PushConstantRay* pc, const GlobalUniforms* cam, const vkrt_trace_opts* opts, const vkrt_shard* shard,
               const vkrt_gbuffer* g, const vkrt_nrd_planes* nrd, float* accum, void* hip_stream)
{
  if(s && pc && cam && shard && g && vkrt_shard_rows(shard) == 0u)
    return VKRT_OK;
  if(!s || !pc || !cam || !shard || !g || !accum || !g->color || !g->position || !g->normal || !g->roughMetal)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "NULL argument");
  if(pc->lightsCount < 0 || (uint32_t)pc->lightsCount > s->lightCount)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "PushConstantRay.lightsCount %d outside [0,%u]", pc->lightsCount, s->lightCount);
  if(pc->depth < 0 || pc->depth > 99)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "depth out of range");
  int rc = setDevice(s);
  if(rc != VKRT_OK)
    return rc;
  TraceParams P;
  if((rc = fillParams(s, pc, cam, opts, shard, P)) != VKRT_OK)
    return rc;
  if(P.localRows == 0)
    return VKRT_OK;
  hipStream_t stream = (hipStream_t)hip_stream;
  HIP_TRY(hipEventRecord(s->evStart, stream));
  NrdPlanes np{};
  if(nrd)
  {
    np.viewZ = nrd->viewZ; np.radHitD = nrd->diffRadianceHitDist;
  }
  // The GI paths (raytraceHybrid.rgen:172-282) run on the path tracer's wavefront streams when the scene was built for them:
  // k_hybrid does the shadow and AO rays and leaves the per-pixel state for k_hy_gi_init (wavefront.hip).  The megakernel mode
  // keeps the whole rgen in k_hybrid.
  const bool giOnStreams = s->wavefront && pc->useGI == 1 && P.fullW <= 65535u && P.localRows <= 65535u;
  if(giOnStreams)
  {
    if((rc = ensureWorkingSet(s, P.tileCount * 64u, 1, stream)) != VKRT_OK)
      return rc;
    P.tileFirst = 0;
    HIP_TRY(vkrt_launch_hybrid(P, g->color, g->position, g->normal, g->roughMetal, accum, nrd ? &np : nullptr, vkrt_wf_hybrid_tmp(s->wf), stream));
    HybridGi G{(const float4*)g->color, (const float4*)g->position, (const float4*)g->normal, (const float2*)g->roughMetal, (float4*)accum,
               nrd ? (float4*)nrd->diffRadianceHitDist : nullptr, nrd ? nrd->viewZ : nullptr};
    HIP_TRY(vkrt_launch_hybrid_gi(P, s->wf, G, (s->dev.watertight || s->dev.dissolve) ? 64u : (unsigned)s->opt[VKRT_OPT_WF_TRAV_BLOCK], stream));
  }
  else
    HIP_TRY(vkrt_launch_hybrid(P, g->color, g->position, g->normal, g->roughMetal, accum, nrd ? &np : nullptr, nullptr, stream));
  HIP_TRY(hipEventRecord(s->evStop, stream));
  s->timed = true;
  s->wfTimed = false;
  return VKRT_OK;
}
}  // namespace

int vkrt_post(int device, const PushConstantPost* pc, uint32_t n, const float* mainImg, const float* rtImg, float* out, void* hip_stream)
{
  if(!pc || !mainImg || !out || (pc->rtMode == 0 && !rtImg))
    return fail(VKRT_ERR_INVALID_ARGUMENT, "NULL argument");
  if(vkrt_device_count() <= 0)
    return fail(VKRT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
  HIP_TRY(hipSetDevice(device));
  if(n == 0)
    return VKRT_OK;
  HIP_TRY(vkrt_launch_post(pc->rtMode, pc->viewAccumulated, pc->useGI, n, mainImg, rtImg, out, (hipStream_t)hip_stream));
  return VKRT_OK;
}

int vkrt_counters_reset(vkrt_scene* s, void* hip_stream)
{
  if(!s)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "scene is NULL");
  int rc = setDevice(s);
  if(rc != VKRT_OK)
    return rc;
  HIP_TRY(hipMemsetAsync(s->counters, 0, sizeof(DevCounters), (hipStream_t)hip_stream));
  return VKRT_OK;
}

int vkrt_counters_read(vkrt_scene* s, vkrt_counters* out)
{
  if(!s || !out)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "NULL argument");
  int rc = setDevice(s);
  if(rc != VKRT_OK)
    return rc;
  HIP_TRY(hipDeviceSynchronize());
  DevCounters h;
  HIP_TRY(hipMemcpy(&h, s->counters, sizeof h, hipMemcpyDeviceToHost));
  unsigned long long t[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for(int sl = 0; sl < VKRT_COUNTER_SLOTS; sl++)
    for(int k = 0; k < 12; k++) t[k] += h.v[sl][k];
  out->rays_closest = t[0]; out->rays_shadow = t[1]; out->hits = t[2]; out->diffuse_hits = t[3];
  out->tex_taps = t[4]; out->pixels = t[5]; out->nodes_visited = t[6]; out->tris_tested = t[7];
  out->wave_node_steps = t[8]; out->wave_tri_steps = t[9]; out->traversal_faults = t[10]; out->pair_records = t[11];
  return VKRT_OK;
}

int vkrt_last_trace_ms(vkrt_scene* s, float* ms)
{
  if(!s || !ms)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "NULL argument");
  if(!s->timed)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "no trace has been launched on this scene");
  HIP_TRY(hipEventSynchronize(s->evStop));
  HIP_TRY(hipEventElapsedTime(ms, s->evStart, s->evStop));
  return VKRT_OK;
}

int vkrt_last_trace_timing(vkrt_scene* s, vkrt_trace_timing* out)
{
  if(!s || !out)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "NULL argument");
  if(!s->timed)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "no trace has been launched on this scene");
  memset(out, 0, sizeof *out);
  HIP_TRY(hipEventSynchronize(s->evStop));
  HIP_TRY(hipEventElapsedTime(&out->total_ms, s->evStart, s->evStop));
  out->mode = s->wavefront ? 1u : 0u;
  if(s->wfTimed)
  {
    for(int k = 0; k < s->wfTiming.used; k++)
    {
      float ms = 0;
      HIP_TRY(hipEventElapsedTime(&ms, s->wfEvents[2 * k], s->wfEvents[2 * k + 1]));
      out->traverse_ms += ms;
    }
    out->traverse_launches = (uint32_t)s->wfTiming.used;
    // the shade launch of round r sits between the traversal launches of rounds r and r + 1 on the same stream
    const int rounds = s->wfTimingRounds;
    for(int k = 0; k + 1 < s->wfTiming.used; k++)
    {
      if(rounds > 0 && (k + 1) % rounds == 0)
        continue;  // a frame ends here: its last shade launch is followed by the next frame's begin
      float ms = 0;
      HIP_TRY(hipEventElapsedTime(&ms, s->wfEvents[2 * k + 1], s->wfEvents[2 * k + 2]));
      out->shade_ms += ms;
      out->shade_launches++;
    }
  }
  else if(!s->wavefront)
  {
    out->traverse_ms = out->total_ms;
    out->traverse_launches = 1;
  }
  return VKRT_OK;
}

int vkrt_debug_trace_rays(vkrt_scene* s, uint32_t n, const float* origins, const float* directions, float tmin, float tmax,
                          int any_hit, float* t, float* u, float* v, int32_t* gid)
{
  if(!s || (n && (!origins || !directions || !t || !u || !v || !gid)))
    return fail(VKRT_ERR_INVALID_ARGUMENT, "NULL argument");
  if(!s->built)
    return fail(VKRT_ERR_NOT_BUILT, "vkrt_debug_trace_rays before vkrt_accel_build");
  if(n == 0)
    return VKRT_OK;
  int rc = setDevice(s);
  if(rc != VKRT_OK)
    return rc;
  float *dO = nullptr, *dD = nullptr, *dT = nullptr, *dU = nullptr, *dV = nullptr;
  int* dG = nullptr;
  const size_t b3 = (size_t)n * 3 * sizeof(float), b1 = (size_t)n * sizeof(float);
  hipError_t e = hipSuccess;
  auto tryHip = [&](hipError_t x) { if(e == hipSuccess) e = x; };
  tryHip(hipMalloc((void**)&dO, b3)); tryHip(hipMalloc((void**)&dD, b3)); tryHip(hipMalloc((void**)&dT, b1));
  tryHip(hipMalloc((void**)&dU, b1)); tryHip(hipMalloc((void**)&dV, b1)); tryHip(hipMalloc((void**)&dG, b1));
  if(e == hipSuccess) tryHip(hipMemcpy(dO, origins, b3, hipMemcpyHostToDevice));
  if(e == hipSuccess) tryHip(hipMemcpy(dD, directions, b3, hipMemcpyHostToDevice));
  if(e == hipSuccess) tryHip(vkrt_launch_trace_rays(s->dev, n, dO, dD, tmin, tmax, any_hit, dT, dU, dV, dG, nullptr));
  if(e == hipSuccess) tryHip(hipDeviceSynchronize());
  if(e == hipSuccess) tryHip(hipMemcpy(t, dT, b1, hipMemcpyDeviceToHost));
  if(e == hipSuccess) tryHip(hipMemcpy(u, dU, b1, hipMemcpyDeviceToHost));
  if(e == hipSuccess) tryHip(hipMemcpy(v, dV, b1, hipMemcpyDeviceToHost));
  if(e == hipSuccess) tryHip(hipMemcpy(gid, dG, b1, hipMemcpyDeviceToHost));
  (void)hipFree(dO); (void)hipFree(dD); (void)hipFree(dT); (void)hipFree(dU); (void)hipFree(dV); (void)hipFree(dG);
  if(e != hipSuccess)
    return fail(VKRT_ERR_HIP, "vkrt_debug_trace_rays: %s", hipGetErrorString(e));
  return VKRT_OK;
}

int vkrt_debug_check_accel(vkrt_scene* s, vkrt_accel_check* out)
{
  if(!s || !out)
    return fail(VKRT_ERR_INVALID_ARGUMENT, "NULL argument");
  if(!s->built)
    return fail(VKRT_ERR_NOT_BUILT, "vkrt_accel_build has not run");
  int rc = setDevice(s);
  if(rc != VKRT_OK)
    return rc;
  HIP_TRY(hipDeviceSynchronize());
  memset(out, 0, sizeof *out);
  const uint32_t T = s->dev.triCount, N = s->info.node_count;
  out->layout = s->dev.layout;
  std::vector<float> tris((size_t)T * 12);
  if(T)
    HIP_TRY(hipMemcpy(tris.data(), s->accelTris, tris.size() * 4, hipMemcpyDeviceToHost));
  std::vector<uint32_t> seen(T, 0u);
  std::vector<uint8_t> reached(N, 0);
  struct Bound { float lo[3], hi[3]; };
  std::vector<Bound> chain;  // boxes of the slots on the path from the root
  // slots per triangle id: a triangle with one slot must lie inside every box above that slot (its vertices do: convexity); the slots
  // of a pre-split triangle (VKRT_OPT_SPLIT_BUDGET) are checked together at the end: `reach[slot]` = intersection of the chain
  auto gidOf = [&](uint32_t slot) { uint32_t g; memcpy(&g, &tris[(size_t)slot * 12 + 9], 4); return g & 0x7fffffffu; };
  auto vertexOf = [&](uint32_t slot, int v, float p[3]) {
    const float* t = &tris[(size_t)slot * 12];  // (v0, e1, e2) or, watertight, (p0, p1, p2)
    for(int k = 0; k < 3; k++)
      p[k] = v == 0 ? t[k] : (s->dev.watertight ? t[3 * v + k] : (v == 1 ? t[k] + t[3 + k] : t[k] + t[6 + k]));
  };
  const uint32_t gidCount = s->info.triangle_count;
  std::vector<uint32_t> slotsOfGid(gidCount, 0u);
  for(uint32_t k = 0; k < T; k++)
  {
    const uint32_t g = gidOf(k);
    if(g < gidCount) slotsOfGid[g]++; else out->bad_references++;
  }
  std::vector<Bound> reach(T);
  for(Bound& b : reach)
    for(int k = 0; k < 3; k++) { b.lo[k] = -INFINITY; b.hi[k] = INFINITY; }
  auto checkTriangle = [&](uint32_t slot) {
    out->triangles_referenced++;
    if(slot >= T) { out->bad_references++; return; }
    if(seen[slot]++) { out->triangles_repeated++; return; }
    const uint32_t g = gidOf(slot);
    const bool single = g < gidCount && slotsOfGid[g] == 1u;
    for(const Bound& b : chain)
      for(int k = 0; k < 3; k++)
      {
        reach[slot].lo[k] = std::max(reach[slot].lo[k], b.lo[k]);
        reach[slot].hi[k] = std::min(reach[slot].hi[k], b.hi[k]);
      }
    if(!single)
      return;
    for(int v = 0; v < 3; v++)
    {
      float p[3];
      vertexOf(slot, v, p);
      for(const Bound& b : chain)
        for(int k = 0; k < 3; k++)
          if(!(p[k] >= b.lo[k] && p[k] <= b.hi[k])) { out->box_violations++; break; }
    }
  };
  if(s->dev.layout == 1u)
  {
    std::vector<uint32_t> nodes((size_t)N * 20);
    if(N)
      HIP_TRY(hipMemcpy(nodes.data(), s->accelNodes, nodes.size() * 4, hipMemcpyDeviceToHost));
    struct Item { uint32_t node, depth; size_t chainLen; Bound b; bool hasBound; };
    std::vector<Item> stack;
    if(s->dev.rootRef != VKRT_TRAV_DONE && N)
      stack.push_back(Item{0u, 1u, 0, Bound{}, false});
    while(!stack.empty())
    {
      const Item it = stack.back();
      stack.pop_back();
      chain.resize(it.chainLen);
      if(it.hasBound) chain.push_back(it.b);
      if(it.node >= N || reached[it.node]++) { out->bad_references++; continue; }
      out->nodes_reached++;
      out->max_depth = std::max(out->max_depth, it.depth);
      const uint32_t* n = &nodes[(size_t)it.node * 20];
      float org[3];
      memcpy(org, n, 12);
      const uint32_t ew = n[3], imask = ew >> 24, childBase = n[4], triBase = n[5];
      double cell[3];
      for(int k = 0; k < 3; k++) cell[k] = std::ldexp(1.0, (int)((ew >> (8 * k)) & 255u) - 127);
      for(int slot = 0; slot < 8; slot++)
      {
        const uint32_t meta = (n[6 + (slot >> 2)] >> (8 * (slot & 3))) & 255u;
        if(meta == 0u)
        {
          if(imask & (1u << slot)) out->bad_references++;
          continue;
        }
        Bound b;
        for(int k = 0; k < 3; k++)
        {
          const uint32_t ql = (n[8 + 2 * k + (slot >> 2)] >> (8 * (slot & 3))) & 255u;
          const uint32_t qh = (n[14 + 2 * k + (slot >> 2)] >> (8 * (slot & 3))) & 255u;
          b.lo[k] = (float)((double)org[k] + ql * cell[k]);  // (exact in double; the kernel pads its own float arithmetic)
          b.hi[k] = (float)((double)org[k] + qh * cell[k]);
          if((double)b.lo[k] > (double)org[k] + ql * cell[k]) b.lo[k] = std::nextafter(b.lo[k], -INFINITY);
          if((double)b.hi[k] < (double)org[k] + qh * cell[k]) b.hi[k] = std::nextafter(b.hi[k], INFINITY);
        }
        const bool internal = (imask >> slot) & 1u;
        if(internal)
        {
          if(meta != (0x20u | (24u + (uint32_t)slot))) out->bad_references++;
          const uint32_t rank = (uint32_t)__builtin_popcount(imask & ((1u << slot) - 1u));
          stack.push_back(Item{childBase + rank, it.depth + 1u, chain.size(), b, true});
        }
        else
        {
          const uint32_t unary = meta >> 5, off = meta & 31u;
          const uint32_t cnt = (uint32_t)__builtin_popcount(unary);
          if(unary != (1u << cnt) - 1u || cnt == 0u || off + cnt > 24u) out->bad_references++;
          chain.push_back(b);
          for(uint32_t t = 0; t < cnt; t++) checkTriangle(triBase + off + t);
          chain.pop_back();
        }
      }
    }
  }
  else
  {
    std::vector<float> nodes((size_t)N * 16);
    if(N)
      HIP_TRY(hipMemcpy(nodes.data(), s->accelNodes, nodes.size() * 4, hipMemcpyDeviceToHost));
    struct Item { int32_t ref; uint32_t depth; size_t chainLen; Bound b; bool hasBound; };
    std::vector<Item> stack;
    if(s->dev.rootRef != VKRT_TRAV_DONE)
      stack.push_back(Item{s->dev.rootRef, 1u, 0, Bound{}, false});
    while(!stack.empty())
    {
      const Item it = stack.back();
      stack.pop_back();
      chain.resize(it.chainLen);
      if(it.hasBound) chain.push_back(it.b);
      if(it.ref < 0)
      {  // leaf: ~(first slot << 3 | count - 1)
        const uint32_t code = (uint32_t)~it.ref, first = code >> 3, cnt = (code & 7u) + 1u;
        out->max_depth = std::max(out->max_depth, it.depth - 1u);
        for(uint32_t t = 0; t < cnt; t++) checkTriangle(first + t);
        continue;
      }
      if((uint32_t)it.ref >= N || reached[(uint32_t)it.ref]++) { out->bad_references++; continue; }
      out->nodes_reached++;
      out->max_depth = std::max(out->max_depth, it.depth);
      const float* n = &nodes[(size_t)it.ref * 16];
      for(int c = 0; c < 2; c++)
      {
        Bound b;
        for(int k = 0; k < 3; k++) { b.lo[k] = n[6 * c + k]; b.hi[k] = n[6 * c + 3 + k]; }
        int32_t ref;
        memcpy(&ref, &n[12 + c], 4);
        stack.push_back(Item{ref, it.depth + 1u, chain.size(), b, true});
      }
    }
  }
  for(uint32_t t = 0; t < T; t++)
    if(!seen[t]) out->triangles_missing++;
  // pre-split triangles: every lattice point of the triangle must be reachable through at least one of its slots
  {
    std::vector<uint32_t> first(gidCount + 1, 0u), fill(gidCount, 0u), bySlot(T);
    for(uint32_t g = 0; g < gidCount; g++) first[g + 1] = first[g] + slotsOfGid[g];
    for(uint32_t k = 0; k < T; k++)
    {
      const uint32_t g = gidOf(k);
      if(g < gidCount) bySlot[first[g] + fill[g]++] = k;
    }
    for(uint32_t g = 0; g < gidCount; g++)
    {
      if(slotsOfGid[g] == 0u) { out->triangles_uncovered++; continue; }  // an instanced triangle the tree does not hold at all
      if(slotsOfGid[g] == 1u) continue;
      out->triangles_split++;
      const uint32_t s0 = bySlot[first[g]];
      float v[3][3];
      for(int c = 0; c < 3; c++) vertexOf(s0, c, v[c]);
      bool covered = true;
      const int n = 8;  // lattice: barycentric (i, j, n - i - j) / n
      for(int i = 0; i <= n && covered; i++)
        for(int j = 0; i + j <= n && covered; j++)
        {
          double p[3];
          for(int k = 0; k < 3; k++) p[k] = ((double)v[0][k] * (n - i - j) + (double)v[1][k] * i + (double)v[2][k] * j) / n;
          bool inSome = false;
          for(uint32_t q = first[g]; q < first[g + 1] && !inSome; q++)
          {
            if(!seen[bySlot[q]]) continue;
            const Bound& b = reach[bySlot[q]];
            bool in = true;
            // (the lattice point is rounded from double; a piece's box is padded by 8 ulp of the coordinates: allow as much)
            for(int k = 0; k < 3; k++)
            {
              const double tol = 1e-6 * std::max(1.0, std::fabs(p[k]));
              in = in && p[k] >= (double)b.lo[k] - tol && p[k] <= (double)b.hi[k] + tol;
            }
            inSome = in;
          }
          covered = inSome;
        }
      if(!covered) out->triangles_uncovered++;
    }
  }
  return VKRT_OK;
}

int vkrt_debug_eval_math(int device, int op, uint32_t n, const float* a, const float* b, float* out)
{
  if(n && (!a || !b || !out))
    return fail(VKRT_ERR_INVALID_ARGUMENT, "NULL argument");
  if(vkrt_device_count() <= 0)
    return fail(VKRT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
  if(n == 0)
    return VKRT_OK;
  HIP_TRY(hipSetDevice(device));
  float *dA = nullptr, *dB = nullptr, *dC = nullptr;
  const size_t bytes = (size_t)n * sizeof(float);
  hipError_t e = hipSuccess;
  auto tryHip = [&](hipError_t x) { if(e == hipSuccess) e = x; };
  tryHip(hipMalloc((void**)&dA, bytes)); tryHip(hipMalloc((void**)&dB, bytes)); tryHip(hipMalloc((void**)&dC, bytes));
  if(e == hipSuccess) tryHip(hipMemcpy(dA, a, bytes, hipMemcpyHostToDevice));
  if(e == hipSuccess) tryHip(hipMemcpy(dB, b, bytes, hipMemcpyHostToDevice));
  if(e == hipSuccess) tryHip(vkrt_launch_eval_math(op, n, dA, dB, dC, nullptr));
  if(e == hipSuccess) tryHip(hipDeviceSynchronize());
  if(e == hipSuccess) tryHip(hipMemcpy(out, dC, bytes, hipMemcpyDeviceToHost));
  (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC);
  if(e != hipSuccess)
    return fail(VKRT_ERR_HIP, "vkrt_debug_eval_math: %s", hipGetErrorString(e));
  return VKRT_OK;
}

}  // extern "C"
