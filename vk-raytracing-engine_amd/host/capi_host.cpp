// capi_host.cpp -- C wrappers over the host layer so the Python test harness can drive it with ctypes
// (glTF ingest, camera matrices, config parsing, and an end-to-end render through HelloVkrt).
#include <cstring>
#include <string>

#include "hello_vkrt.h"
#include "strip_gather.h"

using namespace vkrt_host;

namespace {
thread_local std::string g_err;
}

extern "C" {

const char* vkrt_host_last_error() { return g_err.c_str(); }

void* vkrt_host_load_gltf(const char* path)
{
  try { return new GltfScene(loadGltf(path)); }
  catch(const std::exception& e) { g_err = e.what(); return nullptr; }
}
void vkrt_host_free_scene(void* s) { delete (GltfScene*)s; }

// counts: [vertices, indices, primMeshes, nodes, materials, lights, textures]
void vkrt_host_scene_counts(const void* s_, uint32_t* counts)
{
  const GltfScene* s = (const GltfScene*)s_;
  counts[0] = s->vertexCount(); counts[1] = (uint32_t)s->m_indices.size(); counts[2] = (uint32_t)s->m_primMeshes.size();
  counts[3] = (uint32_t)s->m_nodes.size(); counts[4] = (uint32_t)s->m_materials.size(); counts[5] = (uint32_t)s->m_lights.size();
  counts[6] = (uint32_t)s->m_textures.size();
}
void vkrt_host_scene_copy(const void* s_, float* pos, float* nrm, float* tan, float* uv, uint32_t* idx, vkrt_prim_mesh* pm,
                          vkrt_node* nodes, GltfPBRMaterial* mats, GltfLight* lights)
{
  const GltfScene* s = (const GltfScene*)s_;
  memcpy(pos, s->m_positions.data(), s->m_positions.size() * 4);
  memcpy(nrm, s->m_normals.data(), s->m_normals.size() * 4);
  memcpy(tan, s->m_tangents.data(), s->m_tangents.size() * 4);
  memcpy(uv, s->m_texcoords0.data(), s->m_texcoords0.size() * 4);
  memcpy(idx, s->m_indices.data(), s->m_indices.size() * 4);
  memcpy(pm, s->m_primMeshes.data(), s->m_primMeshes.size() * sizeof(vkrt_prim_mesh));
  memcpy(nodes, s->m_nodes.data(), s->m_nodes.size() * sizeof(vkrt_node));
  memcpy(mats, s->m_materials.data(), s->m_materials.size() * sizeof(GltfPBRMaterial));
  memcpy(lights, s->m_lights.data(), s->m_lights.size() * sizeof(GltfLight));
}
void vkrt_host_texture_info(const void* s_, uint32_t i, uint32_t* whs)
{
  const GltfScene* s = (const GltfScene*)s_;
  whs[0] = s->m_textures[i].width; whs[1] = s->m_textures[i].height; whs[2] = s->m_textures[i].srgb ? 1 : 0;
}
void vkrt_host_texture_copy(const void* s_, uint32_t i, uint8_t* rgba)
{
  const GltfScene* s = (const GltfScene*)s_;
  memcpy(rgba, s->m_textures[i].rgba.data(), s->m_textures[i].rgba.size());
}

// HelloVulkan::updateUniformBuffer for an explicit look-at camera
void vkrt_host_global_uniforms(const float* eye, const float* center, const float* up, float fov, int width, int height,
                               GlobalUniforms* out)
{
  CameraManipulator cam;
  cam.setLookat(Vec3{eye[0], eye[1], eye[2]}, Vec3{center[0], center[1], center[2]}, Vec3{up[0], up[1], up[2]});
  cam.setFov(fov);
  *out = makeGlobalUniforms(cam, width, height);
}

// parse config text; out: [scene, vsync, width, height, samples, depth, frames, seed, nscenes, framesPerCall, watertight, anyHitDissolve,
// skipDeadShadowRays] (the three library options: -1 = key absent); path of the selected scene
int vkrt_host_parse_config(const char* text, int* out, char* scenePath, int cap)
{
  try
  {
    const AppConfig c = parseConfig(text);
    out[0] = c.scene; out[1] = c.vsync; out[2] = c.width; out[3] = c.height; out[4] = c.samples; out[5] = c.depth;
    out[6] = c.frames; out[7] = c.seed; out[8] = (int)c.scenes.size();
    out[9] = c.framesPerCall; out[10] = c.watertight; out[11] = c.anyHitDissolve; out[12] = c.skipDeadShadowRays;
    strncpy(scenePath, c.scenePath().c_str(), (size_t)cap - 1);
    scenePath[cap - 1] = 0;
    return 0;
  }
  catch(const std::exception& e) { g_err = e.what(); return 1; }
}

// End-to-end through the HelloVulkan-shaped class: load, build, `frames` x (updateUniformBuffer,
// updateFrame, pathtrace), download.  seed for frame f = seed0 + f.  Returns 0 on success.
int vkrt_host_render_gltf(const char* path, int device, int width, int height, int samples, int depth, int frames, uint32_t seed0,
                          const float* eye, const float* center, const float* up, float fov, uint32_t buildFlags, float* rgbaOut)
{
  try
  {
    HelloVkrt vk(device);
    vk.setup(width, height);
    vk.CameraManip.setLookat(Vec3{eye[0], eye[1], eye[2]}, Vec3{center[0], center[1], center[2]}, Vec3{up[0], up[1], up[2]});
    vk.CameraManip.setFov(fov);
    vk.loadGltfScene(path);
    vk.createOffscreenRender();
    vk.initRayTracing();
    vk.m_buildFlags = buildFlags;
    vk.createBottomLevelASGltf();
    vk.createTopLevelAsGltf();
    vk.m_pcRay.samples = samples;
    vk.m_pcRay.depth = depth;
    const float clear[4] = {1, 1, 1, 1};
    for(int f = 0; f < frames; f++)
    {
      vk.updateUniformBuffer();
      vk.updateFrame();
      vk.m_seed = seed0 + (uint32_t)f;
      vk.pathtrace(clear);
    }
    std::vector<float> img;
    vk.downloadImage(img);
    memcpy(rgbaOut, img.data(), img.size() * sizeof(float));
    return 0;
  }
  catch(const std::exception& e) { g_err = e.what(); return 1; }
}

// The hybrid sequence of the reference's frame loop (main.cpp:510-561: rasterizeGltf -> raytraceRasterizedScene -> drawPost)
// through HelloVkrt for ONE rank of a `world`-rank job (setShard): displayOut receives the rank's display strips (rows of its
// shard x width x rgba32f, after post.frag).  world = 1: the whole image.  GI on, shadows and AO on.
int vkrt_host_render_gltf_hybrid(const char* path, int device, int width, int height, int depth, int frames, uint32_t seed0, const float* eye,
                                 const float* center, const float* up, float fov, uint32_t rank, uint32_t world, float* displayOut)
{
  try
  {
    HelloVkrt vk(device);
    vk.setup(width, height);
    vk.setShard(rank, world);
    vk.CameraManip.setLookat(Vec3{eye[0], eye[1], eye[2]}, Vec3{center[0], center[1], center[2]}, Vec3{up[0], up[1], up[2]});
    vk.CameraManip.setFov(fov);
    vk.loadGltfScene(path);
    vk.createOffscreenRender();
    vk.initRayTracing();
    vk.createBottomLevelASGltf();
    vk.createTopLevelAsGltf();
    vk.m_pcRay.samples = 1;
    vk.m_pcRay.depth = depth;
    vk.m_pcRay.useShadows = 1; vk.m_pcRay.useAO = 1; vk.m_pcRay.useGI = 1;
    vk.m_pcPost.rtMode = 0;
    const float clear[4] = {1, 1, 1, 1};
    for(int f = 0; f < frames; f++)
    {
      vk.updateUniformBuffer();
      vk.updateFrame();
      vk.m_seed = seed0 + (uint32_t)f;
      vk.rasterizeGltf(clear);
      vk.raytraceRasterizedScene();
    }
    std::vector<float> display;
    vk.drawPost(display);
    memcpy(displayOut, display.data(), display.size() * sizeof(float));
    return 0;
  }
  catch(const std::exception& e) { g_err = e.what(); return 1; }
}

int vkrt_host_decode_png(const uint8_t* data, uint64_t size, uint32_t* wh, uint8_t* rgbaOut, uint64_t cap)
{
  TextureImage t;
  std::string why;
  if(!decodePngMemory(data, (size_t)size, t, why)) { g_err = why; return 1; }
  wh[0] = t.width; wh[1] = t.height;
  if(rgbaOut && cap >= t.rgba.size()) memcpy(rgbaOut, t.rgba.data(), t.rgba.size());
  return 0;
}

/* PNG or JPEG by signature */
int vkrt_host_decode_image(const uint8_t* data, uint64_t size, uint32_t* wh, uint8_t* rgbaOut, uint64_t cap)
{
  TextureImage t;
  std::string why;
  if(!decodeImageMemory(data, (size_t)size, t, why)) { g_err = why; return 1; }
  wh[0] = t.width; wh[1] = t.height;
  if(rgbaOut && cap >= t.rgba.size()) memcpy(rgbaOut, t.rgba.data(), t.rgba.size());
  return 0;
}

int vkrt_host_write_png(const char* path, const float* displayRgba, int w, int h)
{
  try
  {
    writePNG(path, std::vector<float>(displayRgba, displayRgba + (size_t)w * h * 4), w, h);
    return 0;
  }
  catch(const std::exception& e)
  {
    g_err = e.what();
    return 1;
  }
}

// ---- multi-GPU strip layout (strip_gather.h) for the harness: row map on the host, un-interleave kernel on the device ----
uint32_t vkrt_host_strip_rows(uint32_t height, uint32_t stripRows, uint32_t world, uint32_t rank)
{
  vkrt_host::StripLayout L;
  L.width = 1; L.height = height; L.stripRows = stripRows; L.world = world;
  return L.rowsOf(rank);
}
void vkrt_host_strip_source(uint32_t height, uint32_t stripRows, uint32_t world, uint32_t y, uint32_t* rank, uint32_t* local)
{
  vkrt_host::StripLayout L;
  L.width = 1; L.height = height; L.stripRows = stripRows; L.world = world;
  L.source(y, *rank, *local);
}
// gathered: device [world][capRows][width] rgba32f; full: device [height][width] rgba32f
int vkrt_host_unpack_strips(const float* gathered, float* full, uint32_t width, uint32_t height, uint32_t stripRows, uint32_t world, void* hipStream)
{
  vkrt_host::StripLayout L;
  L.width = width; L.height = height; L.stripRows = stripRows; L.world = world;
  const hipError_t e = vkrt_host::unpackStrips(gathered, full, L, (hipStream_t)hipStream);
  if(e != hipSuccess)
  {
    g_err = std::string("unpackStrips: ") + hipGetErrorString(e);
    return 1;
  }
  return 0;
}

}  // extern "C"
