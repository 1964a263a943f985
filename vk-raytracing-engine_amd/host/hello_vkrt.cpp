// hello_vkrt.cpp -- see hello_vkrt.h.  Thin: every compute step is a C-ABI call into libvkrt.so.
#include "hello_vkrt.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <zlib.h>
#include <sstream>
#include <stdexcept>

#include "json_mini.h"

namespace vkrt_host {

void HelloVkrt::check(int rc, const char* what) const
{
  if(rc != VKRT_OK)
    throw std::runtime_error(std::string(what) + ": " + vkrt_last_error());
}

void HelloVkrt::setup(int width, int height)
{
  m_size.width = width;
  m_size.height = height;
  CameraManip.setWindowSize(width, height);
}

void HelloVkrt::loadGltfScene(const std::string& filename) { loadScene(loadGltf(filename)); }

void HelloVkrt::loadScene(const GltfScene& scene)
{
  if(m_scene)
  {
    vkrt_scene_destroy(m_scene);
    m_scene = nullptr;
  }
  m_gltfScene = scene;
  std::vector<vkrt_texture> tex;
  const vkrt_scene_desc d = m_gltfScene.desc(tex);
  check(vkrt_scene_create(&d, m_device, &m_scene), "vkrt_scene_create");
  m_pcRay.lightsCount = (int32_t)m_gltfScene.m_lights.size();  // hello_vulkan.cpp:323-324
}

void HelloVkrt::initRayTracing()
{
  // hello_vulkan.cpp:911-915
  m_pcRay.samples = 1;
  m_pcRay.depth = 3;
  m_pcRay.useShadows = 1;
  m_pcRay.useAO = 1;
  m_pcRay.useGI = 0;
}

void HelloVkrt::createBottomLevelASGltf()
{
  if(!m_scene)
    throw std::runtime_error("createBottomLevelASGltf before loadGltfScene");
  m_blasRequested = true;  // one BLAS per primMesh in the reference; built together with the TLAS here
}

void HelloVkrt::createTopLevelAsGltf()
{
  if(!m_scene || !m_blasRequested)
    throw std::runtime_error("createTopLevelAsGltf before createBottomLevelASGltf");
  // per-handle options the acceleration-structure build consumes (include/vkrt.h): the triangle test, the any-hit stage
  // (only what was specified: an unspecified option keeps the value vkrt_scene_create read from the environment)
  if(m_watertight >= 0)
    check(vkrt_scene_set_option(m_scene, VKRT_OPT_WATERTIGHT, m_watertight ? 1 : 0), "vkrt_scene_set_option(VKRT_OPT_WATERTIGHT)");
  if(m_anyHitDissolve >= 0)
    check(vkrt_scene_set_option(m_scene, VKRT_OPT_ANYHIT_DISSOLVE, m_anyHitDissolve ? 1 : 0), "vkrt_scene_set_option(VKRT_OPT_ANYHIT_DISSOLVE)");
  if(m_skipDeadShadowRays >= 0)
    check(vkrt_scene_set_option(m_scene, VKRT_OPT_SKIP_DEAD_SHADOW_RAYS, m_skipDeadShadowRays ? 1 : 0), "vkrt_scene_set_option(VKRT_OPT_SKIP_DEAD_SHADOW_RAYS)");
  check(vkrt_accel_build(m_scene, m_buildFlags, nullptr), "vkrt_accel_build");
}

void HelloVkrt::setShard(uint32_t rank, uint32_t world)
{
  if(world == 0 || rank >= world)
    throw std::runtime_error("setShard: bad rank / world");
  m_shard = vkrt_shard{0, 0, world > 1 ? 16u : 0u, world, world > 1 ? rank : 0u};
}

void HelloVkrt::createOffscreenRender()
{
  float** planes[6] = {&m_offscreenColor, &m_positionTexture, &m_normalTexture, &m_roughnessTexture, &m_accumulatedTexture, &m_displayImage};
  m_shard.full_width = (uint32_t)m_size.width;
  m_shard.full_height = (uint32_t)m_size.height;
  const size_t px = (size_t)m_size.width * (m_shard.shard_count > 1 ? vkrt_shard_rows(&m_shard) : (uint32_t)m_size.height);
  const size_t bytes[6] = {px * 16, px * 16, px * 16, px * 8, px * 16, px * 16};
  if(hipSetDevice(m_device) != hipSuccess)
    throw std::runtime_error("createOffscreenRender: hipSetDevice failed");
  for(int k = 0; k < 6; k++)
  {
    if(*planes[k]) (void)hipFree(*planes[k]);
    *planes[k] = nullptr;
    if(hipMalloc((void**)planes[k], bytes[k]) != hipSuccess)
      throw std::runtime_error("createOffscreenRender: hipMalloc failed");
    (void)hipMemset(*planes[k], 0, bytes[k]);
  }
}

void HelloVkrt::updateUniformBuffer() { m_hostUBO = makeGlobalUniforms(CameraManip, m_size.width, m_size.height); }

void HelloVkrt::resetFrame() { m_pcRay.frame = -1; }

void HelloVkrt::updateFrame()
{
  const vkrt_mat4 m = CameraManip.getMatrix();
  const float fov = CameraManip.getFov();
  if(!m_refValid || memcmp(&m_refCamMatrix, &m, sizeof m) != 0 || m_refFov != fov)
  {
    resetFrame();
    m_refCamMatrix = m;
    m_refFov = fov;
    m_refValid = true;
  }
  m_pcRay.frame++;
}

void HelloVkrt::pathtrace(const float clearColor[4])
{
  if(m_stopAtMaxFrames && m_pcRay.frame >= m_maxFrames)
    return;
  if(!m_scene || !m_offscreenColor)
    throw std::runtime_error("pathtrace before scene/offscreen image creation");
  for(int k = 0; k < 4; k++) m_pcRay.clearColor[k] = clearColor[k];
  const vkrt_trace_opts opts{m_seed, m_traceFlags};
  const vkrt_shard shard = launchShard();
  check(vkrt_pathtrace(m_scene, &m_pcRay, &m_hostUBO, &opts, &shard, m_offscreenColor, nullptr), "vkrt_pathtrace");
}

void HelloVkrt::pathtraceFrames(const float clearColor[4], int n, bool seedPerFrame)
{
  if(n <= 0)
    return;
  if(m_stopAtMaxFrames)  // hello_vulkan.cpp:1426: frames >= m_maxFrames are not rendered
  {
    if(m_pcRay.frame >= m_maxFrames)
      return;
    n = std::min(n, m_maxFrames - m_pcRay.frame);
  }
  if(!m_scene || !m_offscreenColor)
    throw std::runtime_error("pathtraceFrames before scene/offscreen image creation");
  for(int k = 0; k < 4; k++) m_pcRay.clearColor[k] = clearColor[k];
  const vkrt_trace_opts opts{m_seed, m_traceFlags | (seedPerFrame ? 0u : (uint32_t)VKRT_TRACE_SAME_SEED_EVERY_FRAME)};
  const vkrt_shard shard = launchShard();
  check(vkrt_pathtrace_frames(m_scene, &m_pcRay, &m_hostUBO, &opts, &shard, m_offscreenColor, (uint32_t)n, nullptr), "vkrt_pathtrace_frames");
  m_pcRay.frame += n - 1;
  if(seedPerFrame)
    m_seed += (uint32_t)(n - 1);
}

// the launch geometry of this rank: the whole image, or its strips of the full-size launch (setShard)
vkrt_shard HelloVkrt::launchShard() const
{
  vkrt_shard shard = m_shard;
  shard.full_width = (uint32_t)m_size.width;
  shard.full_height = (uint32_t)m_size.height;
  return shard;
}

void HelloVkrt::rasterizeGltf(const float clearColor[4])
{
  if(!m_scene || !m_offscreenColor)
    throw std::runtime_error("rasterizeGltf before scene/offscreen image creation");
  // sharded like the path tracer: every plane holds this rank's strips stacked from row 0 (the ABI's shard semantics)
  const vkrt_shard shard = launchShard();
  const vkrt_gbuffer g{m_offscreenColor, m_positionTexture, m_normalTexture, m_roughnessTexture};
  check(vkrt_gbuffer_raycast(m_scene, clearColor, m_pcRay.lightsCount /* m_pcRaster.lightsCount, :323 */, &m_hostUBO, &shard, &g, nullptr),
        "vkrt_gbuffer_raycast");
}

void HelloVkrt::raytraceRasterizedScene()
{
  if(m_stopAtMaxFrames && m_pcRay.frame >= m_maxFrames)
    return;
  const vkrt_trace_opts opts{m_seed, m_traceFlags};
  const vkrt_shard shard = launchShard();
  const vkrt_gbuffer g{m_offscreenColor, m_positionTexture, m_normalTexture, m_roughnessTexture};
  check(vkrt_hybrid_trace(m_scene, &m_pcRay, &m_hostUBO, &opts, &shard, &g, m_accumulatedTexture, nullptr), "vkrt_hybrid_trace");
}

const float* HelloVkrt::drawPostDevice()
{
  m_pcPost.aspectRatio = (float)m_size.width / (float)m_size.height;
  m_pcPost.useGI = m_pcRay.useGI;
  const vkrt_shard shard = launchShard();
  const uint32_t n = (uint32_t)((size_t)m_size.width * vkrt_shard_rows(&shard));  // per pixel: the strips of this rank, or the whole image
  check(vkrt_post(m_device, &m_pcPost, n, m_offscreenColor, m_accumulatedTexture, m_displayImage, nullptr), "vkrt_post");
  return m_displayImage;
}

void HelloVkrt::drawPost(std::vector<float>& displayRgba)
{
  const vkrt_shard shard = launchShard();
  const uint32_t n = (uint32_t)((size_t)m_size.width * vkrt_shard_rows(&shard));
  drawPostDevice();
  displayRgba.resize((size_t)n * 4);
  if(hipDeviceSynchronize() != hipSuccess ||
     hipMemcpy(displayRgba.data(), m_displayImage, displayRgba.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
    throw std::runtime_error("drawPost: hipMemcpy failed");
}

void HelloVkrt::onResize(int w, int h)
{
  setup(w, h);
  resetFrame();
  createOffscreenRender();
}

void HelloVkrt::destroyResources()
{
  if(m_scene) vkrt_scene_destroy(m_scene);
  m_scene = nullptr;
  float** planes[6] = {&m_offscreenColor, &m_positionTexture, &m_normalTexture, &m_roughnessTexture, &m_accumulatedTexture, &m_displayImage};
  for(int k = 0; k < 6; k++)
  {
    if(*planes[k]) (void)hipFree(*planes[k]);
    *planes[k] = nullptr;
  }
}

void HelloVkrt::downloadImage(std::vector<float>& rgba) const
{
  rgba.resize((size_t)m_size.width * m_size.height * 4);
  if(hipDeviceSynchronize() != hipSuccess ||
     hipMemcpy(rgba.data(), m_offscreenColor, rgba.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
    throw std::runtime_error("downloadImage: hipMemcpy failed");
}

vkrt_counters HelloVkrt::counters()
{
  vkrt_counters c{};
  check(vkrt_counters_read(m_scene, &c), "vkrt_counters_read");
  return c;
}
vkrt_accel_info HelloVkrt::accelInfo() const
{
  vkrt_accel_info i{};
  check(vkrt_accel_get_info(m_scene, &i), "vkrt_accel_get_info");
  return i;
}
float HelloVkrt::lastTraceMs()
{
  float ms = 0;
  check(vkrt_last_trace_ms(m_scene, &ms), "vkrt_last_trace_ms");
  return ms;
}

// ---- config.json ---------------------------------------------------------------------------------------
AppConfig parseConfig(const std::string& text)
{
  const Json j = Json::parse(text);
  AppConfig c;
  // mandatory keys, exactly the reference's (main.cpp:139-144)
  for(const char* k : {"scenes", "scene", "vsync", "width", "height"})
    if(!j.has(k))
      throw std::runtime_error(std::string("config.json: missing key \"") + k + "\"");
  for(size_t i = 0; i < j["scenes"].size(); i++) c.scenes.push_back(j["scenes"][i].string());
  c.scene = j["scene"].integer();
  c.vsync = j["vsync"].boolean();
  c.width = j["width"].integer();
  c.height = j["height"].integer();
  if(c.scene < 0 || (size_t)c.scene >= c.scenes.size())
    throw std::runtime_error("config.json: \"scene\" index out of range");
  if(c.width <= 0 || c.height <= 0)
    throw std::runtime_error("config.json: bad width/height");
  // optional keys (stand-ins for the ImGui panel state)
  c.samples = j["samples"].integer(c.samples);
  c.depth = j["depth"].integer(c.depth);
  c.frames = j["frames"].integer(c.frames);
  c.seed = j["seed"].integer(c.seed);
  c.seedPerFrame = j["seedPerFrame"].boolean(c.seedPerFrame);
  c.build = j["build"].string(c.build);
  c.mode = j["mode"].string(c.mode);  // "pathtrace" (rtMode 1) or "hybrid" (rtMode 0, the reference's start-up mode)
  c.useShadows = j["useShadows"].boolean(c.useShadows);
  c.useAO = j["useAO"].boolean(c.useAO);
  c.useGI = j["useGI"].boolean(c.useGI);
  if(j.has("watertight")) c.watertight = j["watertight"].boolean(false) ? 1 : 0;
  if(j.has("anyHitDissolve")) c.anyHitDissolve = j["anyHitDissolve"].boolean(false) ? 1 : 0;
  if(j.has("skipDeadShadowRays")) c.skipDeadShadowRays = j["skipDeadShadowRays"].boolean(false) ? 1 : 0;
  c.framesPerCall = j["framesPerCall"].integer(c.framesPerCall);
  if(c.framesPerCall < 1)
    throw std::runtime_error("config.json: \"framesPerCall\" must be >= 1");
  c.output = j["output"].string("");
  if(j.has("clearColor"))
    for(int k = 0; k < 4; k++) c.clearColor[k] = (float)j["clearColor"][(size_t)k].number(1.0);
  if(j.has("camera"))
  {
    const Json& cam = j["camera"];
    c.hasCamera = true;
    auto v3 = [&](const char* k, Vec3 d) {
      if(!cam.has(k)) return d;
      return Vec3{(float)cam[k][0].number(), (float)cam[k][1].number(), (float)cam[k][2].number()};
    };
    c.eye = v3("eye", c.eye);
    c.center = v3("center", c.center);
    c.up = v3("up", c.up);
    c.fov = (float)cam["fov"].number(c.fov);
  }
  return c;
}

AppConfig loadConfig(const std::string& path)
{
  std::ifstream f(path);
  if(!f)
    throw std::runtime_error("cannot open " + path);
  std::stringstream ss;
  ss << f.rdbuf();
  return parseConfig(ss.str());
}

// ---- image writers ---------------------------------------------------------------------------------------
void writePPM(const std::string& path, const std::vector<float>& rgba, int w, int h)
{
  std::ofstream f(path, std::ios::binary);
  f << "P6\n" << w << " " << h << "\n255\n";
  std::vector<unsigned char> row((size_t)w * 3);
  for(int y = 0; y < h; y++)
  {
    for(int x = 0; x < w; x++)
      for(int c = 0; c < 3; c++)
      {
        float v = rgba[((size_t)y * w + x) * 4 + c];
        v = std::pow(std::fmax(v, 0.0f), 1.0f / 2.2f);  // post.frag:39
        row[(size_t)x * 3 + c] = (unsigned char)(std::fmin(v, 1.0f) * 255.0f + 0.5f);
      }
    f.write((const char*)row.data(), (std::streamsize)row.size());
  }
}

// 8-bit RGBA PNG of a display image (values already through post.frag's gamma, [0,1]); zlib does deflate + CRC
void writePNG(const std::string& path, const std::vector<float>& displayRgba, int w, int h)
{
  std::vector<unsigned char> raw((size_t)h * ((size_t)w * 4 + 1));
  for(int y = 0; y < h; y++)
  {
    unsigned char* row = &raw[(size_t)y * ((size_t)w * 4 + 1)];
    row[0] = 0;  // filter: none
    for(int x = 0; x < w * 4; x++)
    {
      const float v = displayRgba[(size_t)y * w * 4 + x];
      row[1 + x] = (unsigned char)(std::fmin(std::fmax(v, 0.0f), 1.0f) * 255.0f + 0.5f);
    }
  }
  uLongf zlen = compressBound((uLong)raw.size());
  std::vector<unsigned char> z(zlen);
  if(compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK)
    throw std::runtime_error("writePNG: deflate failed");
  std::ofstream f(path, std::ios::binary);
  auto be32 = [](unsigned char* p, uint32_t v) { p[0] = (unsigned char)(v >> 24); p[1] = (unsigned char)(v >> 16); p[2] = (unsigned char)(v >> 8); p[3] = (unsigned char)v; };
  auto chunk = [&](const char* tag, const unsigned char* data, size_t n) {
    unsigned char hd[8];
    be32(hd, (uint32_t)n);
    memcpy(hd + 4, tag, 4);
    f.write((const char*)hd, 8);
    if(n) f.write((const char*)data, (std::streamsize)n);
    uLong c = crc32(0L, (const Bytef*)tag, 4);
    if(n) c = crc32(c, data, (uInt)n);
    unsigned char cr[4];
    be32(cr, (uint32_t)c);
    f.write((const char*)cr, 4);
  };
  static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  f.write((const char*)sig, 8);
  unsigned char ihdr[13];
  be32(ihdr, (uint32_t)w); be32(ihdr + 4, (uint32_t)h);
  ihdr[8] = 8; ihdr[9] = 6; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;  // 8-bit RGBA, deflate, no filter method / interlace
  chunk("IHDR", ihdr, 13);
  chunk("IDAT", z.data(), zlen);
  chunk("IEND", nullptr, 0);
  if(!f)
    throw std::runtime_error("writePNG: cannot write " + path);
}

void writePFM(const std::string& path, const std::vector<float>& rgba, int w, int h)
{
  std::ofstream f(path, std::ios::binary);
  f << "PF\n" << w << " " << h << "\n-1.0\n";
  std::vector<float> row((size_t)w * 3);
  for(int y = h - 1; y >= 0; y--)  // PFM stores the bottom row first
  {
    for(int x = 0; x < w; x++)
      for(int c = 0; c < 3; c++) row[(size_t)x * 3 + c] = rgba[((size_t)y * w + x) * 4 + c];
    f.write((const char*)row.data(), (std::streamsize)(row.size() * sizeof(float)));
  }
}

}  // namespace vkrt_host
