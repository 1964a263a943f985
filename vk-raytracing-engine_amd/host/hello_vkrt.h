// hello_vkrt.h -- C++ host mirror of the reference renderer object for the ray-tracing path.
//
// Same member names, argument meaning and call order as the slice of `class HelloVulkan`
// (reference hello_vulkan.h:56-208) that the path tracer uses, so a maintainer can swap the object
// under main.cpp's init sequence (main.cpp:226-240) and frame loop (main.cpp:504-508):
//
//   reference                                         here
//   loadGltfScene(filename)      hello_vulkan.cpp:327  loadGltfScene(filename)   -> vkrt_scene_create
//   initRayTracing()             :901                  initRayTracing()          (defaults :911-918)
//   createBottomLevelASGltf()    :1001                 createBottomLevelASGltf() -> vkrt_accel_build
//   createTopLevelAsGltf()       :1031                 createTopLevelAsGltf()    (instances are part of the same build)
//   createOffscreenRender()      :637                  createOffscreenRender()   (rgba32f image in HBM)
//   updateUniformBuffer(cmdBuf)  :61                   updateUniformBuffer()
//   updateFrame()/resetFrame()   :1501-1521            updateFrame()/resetFrame()
//   pathtrace(cmdBuf,clearColor) :1423                 pathtrace(clearColor)     -> vkrt_pathtrace
//   destroyResources()           :518                  destroyResources()
//
// Errors: the reference asserts / throws std::runtime_error (:336,:341,:1243); this class throws
// std::runtime_error carrying vkrt_last_error().  Everything below the class is the C ABI (include/vkrt.h).
#pragma once
#include <string>
#include <vector>

#include "../../include/vkrt.h"
#include "camera.h"
#include "gltf_scene.h"

namespace vkrt_host {

class HelloVkrt
{
public:
  explicit HelloVkrt(int device = 0) : m_device(device) {}
  ~HelloVkrt() { destroyResources(); }
  HelloVkrt(const HelloVkrt&) = delete;
  HelloVkrt& operator=(const HelloVkrt&) = delete;

  void setup(int width, int height);                      // AppBaseVk::setup + window size
  void loadGltfScene(const std::string& filename);        // hello_vulkan.cpp:327-394
  void loadScene(const GltfScene& scene);                 // same upload from an in-memory scene
  void initRayTracing();                                  // :901-919
  void createBottomLevelASGltf();                         // :1001-1011
  void createTopLevelAsGltf();                            // :1031-1047
  void createOffscreenRender();                           // :637-665 (colour image only)
  void updateUniformBuffer();                             // :61-102
  void resetFrame();                                      // :1501-1504
  void updateFrame();                                     // :1506-1521
  void pathtrace(const float clearColor[4]);              // :1423-1448
  // n iterations of the frame loop with the camera at rest (main.cpp:503-508: updateFrame() -> pathtrace(), n times) as ONE library
  // call (vkrt_pathtrace_frames): call it after updateFrame() like pathtrace(); it renders frames m_pcRay.frame .. + n - 1 with seeds
  // m_seed .. + n - 1 (m_seed for all of them when seedPerFrame is false) and leaves m_pcRay.frame / m_seed at the LAST frame's
  // values, as n - 1 further updateFrame() calls would.  m_stopAtMaxFrames cuts the count like the reference's early return does.
  void pathtraceFrames(const float clearColor[4], int n, bool seedPerFrame = true);
  // hybrid mode (rtMode == 0): main.cpp:510-561 = rasterizeGltf -> raytraceRasterizedScene -> drawPost
  void rasterizeGltf(const float clearColor[4]);          // :583-615 (ray-cast G-buffer: no raster path from HIP)
  void raytraceRasterizedScene();                         // :1450-1473
  void drawPost(std::vector<float>& displayRgba);         // :882-897 + post.frag (composite + gamma), downloaded
  const float* drawPostDevice();                          // the same, left on the device (rows of this rank's shard): what a gather sends
  void onResize(int w, int h);                            // :620-626
  // multi-GPU (one process per GPU): this object renders only the strips of `rank` (vkrt_shard, 16-row strips dealt round-robin);
  // call before createOffscreenRender.  Both modes: pathtrace() and the hybrid sequence rasterizeGltf / raytraceRasterizedScene /
  // drawPost work on the rank's strips (every plane holds them stacked from row 0).  Gathered by StripGather (strip_gather.h).
  void setShard(uint32_t rank, uint32_t world);
  const vkrt_shard& shard() const { return m_shard; }
  const float* offscreenDevice() const { return m_offscreenColor; }
  void destroyResources();                                // :518-578

  // read back m_offscreenColor (rgba32f, row 0 = top) -- what drawPost samples (:882-897)
  void downloadImage(std::vector<float>& rgba) const;
  vkrt_counters counters();
  vkrt_accel_info accelInfo() const;
  float lastTraceMs();

  // public state, as in the reference class
  PushConstantRay m_pcRay{{1, 1, 1, 1}, -1, 0, 1, 3, 1, 1, 0};  // frame = -1 until the first updateFrame
  bool m_stopAtMaxFrames{false};  // hello_vulkan.h:156
  int m_maxFrames{1};             // hello_vulkan.h:157
  CameraManipulator CameraManip;  // the nvh global of the reference
  GlobalUniforms m_hostUBO{};
  GltfScene m_gltfScene;
  struct { int width = 1280, height = 720; } m_size;
  uint32_t m_seed = 0;            // replaces int(clockARB()) (raytrace.rgen:27); advanced every frame
  uint32_t m_buildFlags = VKRT_BUILD_DEFAULT;
  // library options without a counterpart in the reference class (include/vkrt.h vkrt_option); applied by createTopLevelAsGltf.
  // Tri-state: -1 = not specified, the handle keeps what vkrt_scene_create read from the environment (VKRT_WATERTIGHT, ...);
  // 0 / 1 = set.  Precedence: this member (config.json) over the environment over the library default.
  int m_watertight = -1;          // VKRT_OPT_WATERTIGHT: the watertight triangle test a Vulkan driver runs, instead of Moeller-Trumbore
  int m_anyHitDissolve = -1;      // VKRT_OPT_ANYHIT_DISSOLVE: the any-hit stage hello_vulkan.cpp:1185-1191 keeps commented out
  int m_skipDeadShadowRays = -1;  // VKRT_OPT_SKIP_DEAD_SHADOW_RAYS: same pixels, fewer shadow rays
  uint32_t m_traceFlags = 0;
  PushConstantPost m_pcPost{1.0f, 0, 0, 0};  // rtMode 0 = hybrid (hello_vulkan.cpp:917), 1 = path tracer

private:
  void check(int rc, const char* what) const;
  vkrt_shard launchShard() const;
  int m_device;
  vkrt_scene* m_scene = nullptr;
  float* m_offscreenColor = nullptr;  // device rgba32f
  // hybrid planes (createOffscreenRender :637-826): position, normal, rough/metal, accumulation, display
  float* m_positionTexture = nullptr;
  float* m_normalTexture = nullptr;
  float* m_roughnessTexture = nullptr;
  float* m_accumulatedTexture = nullptr;
  float* m_displayImage = nullptr;
  bool m_blasRequested = false;
  vkrt_shard m_shard{0, 0, 0, 1, 0};  // whole image unless setShard was called
  // updateFrame()'s function-local statics in the reference (:1508-1509)
  vkrt_mat4 m_refCamMatrix{};
  float m_refFov = 60.0f;
  bool m_refValid = false;
};

// config.json of the reference (main.cpp:136-145): the five mandatory keys plus optional ones that
// stand in for the ImGui panel (main.cpp:67-105,448-465).
struct AppConfig
{
  std::vector<std::string> scenes;
  int scene = 0;
  bool vsync = false;
  int width = 1280, height = 720;
  // optional
  int samples = 1, depth = 3, frames = 1, seed = 0;
  bool seedPerFrame = true;
  float clearColor[4] = {1, 1, 1, 1};  // main.cpp:247
  bool hasCamera = false;
  Vec3 eye{0, 0, 15}, center{0, 0, 0}, up{0, 1, 0};
  float fov = 60.0f;
  std::string build = "ploc";  // "ploc" (device, default) | "lbvh" (device, fastest build) | "sah" (host)
  std::string mode = "pathtrace";
  bool useShadows = true, useAO = true, useGI = false;  // hello_vulkan.cpp:913-915
  int watertight = -1, anyHitDissolve = -1, skipDeadShadowRays = -1;  // library options (include/vkrt.h): -1 = key absent (environment / library default)
  int framesPerCall = 1;  // > 1: the frame loop hands that many progressive frames to the library at once (HelloVkrt::pathtraceFrames)
  std::string output;
  std::string scenePath() const { return scenes.at((size_t)scene); }
};
AppConfig parseConfig(const std::string& jsonText);
AppConfig loadConfig(const std::string& path);

// post.frag:39,58 -- gamma 1/2.2 on all four channels; writers for review images
void writePPM(const std::string& path, const std::vector<float>& rgba, int w, int h);
void writePNG(const std::string& path, const std::vector<float>& displayRgba, int w, int h);  // display image (post.frag output)
void writePFM(const std::string& path, const std::vector<float>& rgba, int w, int h);

}  // namespace vkrt_host
