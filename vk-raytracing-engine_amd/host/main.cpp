// main.cpp -- `vkrt_render`: head-less stand-in for the reference's application shell
// (reference main.cpp:117-245 init order, :441-630 frame loop) for the path-tracer mode.
// No window, no ImGui, no swapchain: reads config.json, renders `frames` progressive frames and
// writes the rgba32f image (PFM) and a gamma-2.2 preview (PPM, post.frag:39).
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <string>

#include "hello_vkrt.h"

using namespace vkrt_host;

int main(int argc, char** argv)
{
  std::string cfgPath = "config.json", outOverride;
  int device = 0;
  for(int i = 1; i < argc; i++)
  {
    if(!strcmp(argv[i], "--config") && i + 1 < argc) cfgPath = argv[++i];
    else if(!strcmp(argv[i], "--output") && i + 1 < argc) outOverride = argv[++i];
    else if(!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
    else if(!strcmp(argv[i], "-h") || !strcmp(argv[i], "--help"))
    {
      printf("usage: vkrt_render [--config config.json] [--output prefix] [--device N]\n");
      return 0;
    }
    else cfgPath = argv[i];
  }
  try
  {
    const AppConfig cfg = loadConfig(cfgPath);
    const size_t slash = cfgPath.find_last_of("/\\");
    const std::string cfgDir = slash == std::string::npos ? "." : cfgPath.substr(0, slash);
    std::string scenePath = cfg.scenePath();
    if(!scenePath.empty() && scenePath[0] != '/') scenePath = cfgDir + "/" + scenePath;

    HelloVkrt helloVk(device);
    helloVk.setup(cfg.width, cfg.height);                       // main.cpp:209-216
    if(cfg.hasCamera)
    {
      helloVk.CameraManip.setLookat(cfg.eye, cfg.center, cfg.up);
      helloVk.CameraManip.setFov(cfg.fov);
    }
    const auto t0 = std::chrono::steady_clock::now();
    helloVk.loadGltfScene(scenePath);                           // main.cpp:226
    if(!helloVk.m_gltfScene.warnings.empty()) fprintf(stderr, "warning: %s\n", helloVk.m_gltfScene.warnings.c_str());
    helloVk.createOffscreenRender();                            // main.cpp:228
    helloVk.initRayTracing();                                   // main.cpp:235
    helloVk.m_buildFlags = cfg.build == "lbvh" ? VKRT_BUILD_LBVH_GPU : VKRT_BUILD_SAH_HOST;
    helloVk.createBottomLevelASGltf();                          // main.cpp:236
    helloVk.createTopLevelAsGltf();                             // main.cpp:237
    helloVk.m_pcRay.samples = cfg.samples;
    helloVk.m_pcRay.depth = cfg.depth;
    const vkrt_accel_info ai = helloVk.accelInfo();
    const double loadMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    printf("scene %s: %u triangles, %u BVH nodes (depth %u, SAH %.1f), %zu lights; load+build %.1f ms\n", scenePath.c_str(),
           ai.triangle_count, ai.node_count, ai.max_depth, ai.sah_cost, helloVk.m_gltfScene.m_lights.size(), loadMs);

    const bool hybrid = cfg.mode == "hybrid";
    helloVk.m_pcPost.rtMode = hybrid ? 0 : 1;
    helloVk.m_pcRay.useShadows = cfg.useShadows; helloVk.m_pcRay.useAO = cfg.useAO; helloVk.m_pcRay.useGI = cfg.useGI;
    double traceMs = 0;
    for(int f = 0; f < cfg.frames; f++)                         // main.cpp:441 loop body
    {
      helloVk.updateUniformBuffer();                            // main.cpp:503
      helloVk.updateFrame();                                    // main.cpp:504
      helloVk.m_seed = (uint32_t)cfg.seed + (cfg.seedPerFrame ? (uint32_t)f : 0u);
      if(!hybrid)
        helloVk.pathtrace(cfg.clearColor);                      // main.cpp:507
      else
      {
        helloVk.rasterizeGltf(cfg.clearColor);                  // main.cpp:513
        helloVk.raytraceRasterizedScene();                      // main.cpp:547
      }
      traceMs += helloVk.lastTraceMs();
    }
    const vkrt_counters c = helloVk.counters();
    const double rays = (double)(c.rays_closest + c.rays_shadow);
    printf("%d frame(s) %dx%d, %d spp/frame, depth %d: %.3f ms GPU, %.1f Mrays/s (%llu closest + %llu shadow rays)\n", cfg.frames,
           cfg.width, cfg.height, cfg.samples, cfg.depth, traceMs, traceMs > 0 ? rays / traceMs / 1e3 : 0.0,
           (unsigned long long)c.rays_closest, (unsigned long long)c.rays_shadow);
    const std::string out = !outOverride.empty() ? outOverride : cfg.output;
    if(!out.empty())
    {
      std::vector<float> img, display;
      helloVk.drawPost(display);                                // main.cpp:605-612: what the window would show
      writePNG(out + ".png", display, cfg.width, cfg.height);
      if(hybrid)
      {  // composite of the raster plane and the ray-traced plane (post.frag:41-47), before gamma for the PFM
        img.resize(display.size());
        for(size_t k = 0; k < display.size(); k++) img[k] = std::pow(display[k], 2.2f);
      }
      else
        helloVk.downloadImage(img);
      writePFM(out + ".pfm", img, cfg.width, cfg.height);
      writePPM(out + ".ppm", img, cfg.width, cfg.height);
      printf("wrote %s.pfm / %s.ppm / %s.png\n", out.c_str(), out.c_str(), out.c_str());
    }
  }
  catch(const std::exception& e)
  {
    fprintf(stderr, "vkrt_render: %s\n", e.what());
    return 1;
  }
  return 0;
}
