// main.cpp -- `vkrt_render`: head-less stand-in for the reference's application shell
// (reference main.cpp:117-245 init order, :441-630 frame loop) for the path-tracer mode.
// No window, no ImGui, no swapchain: reads config.json, renders `frames` progressive frames and
// writes the rgba32f image (PFM) and a gamma-2.2 preview (PPM, post.frag:39).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include <cerrno>
#include <csignal>
#include <sys/wait.h>
#include <unistd.h>

#include <hip/hip_runtime.h>

#include "hello_vkrt.h"
#include "strip_gather.h"

using namespace vkrt_host;

// `--ranks N`: one child process per GPU (rank r renders on device r), started before this process touches HIP; the children
// find each other through the RCCL id file, which lives in a private directory (mkdtemp: mode 0700, unpredictable name, so no
// other user can plant a file or a symbolic link where rank 0 will write).  The ranks run a collective: if one of them dies --
// a bad device index, an allocation failure, an exception after ncclCommInitRank -- the others would block for ever in
// ncclCommInitRank / ncclAllGather.  So the children are reaped in the order they exit, and on the first abnormal or non-zero
// exit the remaining ones are terminated (SIGTERM, then SIGKILL) and that rank's code is returned.
static int spawnRanks(int ranks, int argc, char** argv)
{
  // RCCL between processes needs dmabuf IPC on hosts whose driver has no legacy IPC (hipIpcGetMemHandle: invalid argument otherwise)
  setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
  char dir[] = "/tmp/vkrt_rccl_XXXXXX";
  if(!mkdtemp(dir))
  {
    perror("mkdtemp");
    return 1;
  }
  const std::string idFile = std::string(dir) + "/id";
  std::vector<pid_t> kids;
  auto cleanup = [&]() {
    unlink(idFile.c_str());
    unlink((idFile + ".tmp").c_str());
    rmdir(dir);
  };
  auto killRest = [&](int sig) {
    for(pid_t k : kids)
      if(k > 0) kill(k, sig);
  };
  for(int r = 0; r < ranks; r++)
  {
    const pid_t pid = fork();
    if(pid == 0)
    {
      std::vector<std::string> keep;
      for(int i = 0; i < argc; i++)
      {
        if(!strcmp(argv[i], "--ranks")) { i++; continue; }
        keep.push_back(argv[i]);
      }
      keep.insert(keep.end(), {"--rank", std::to_string(r), "--world", std::to_string(ranks), "--rccl-id-file", idFile, "--device", std::to_string(r)});
      std::vector<char*> av;
      for(auto& k : keep) av.push_back(const_cast<char*>(k.c_str()));
      av.push_back(nullptr);
      execv("/proc/self/exe", av.data());
      perror("execv");
      _exit(127);
    }
    if(pid < 0)
    {
      perror("fork");
      killRest(SIGTERM);
      for(pid_t k : kids) waitpid(k, nullptr, 0);
      cleanup();
      return 1;
    }
    kids.push_back(pid);
  }
  int result = 0;
  size_t alive = kids.size();
  bool terminating = false;
  while(alive > 0)
  {
    int st = 0;
    const pid_t done = waitpid(-1, &st, 0);  // whichever rank exits first
    if(done < 0)
    {
      if(errno == EINTR) continue;
      break;
    }
    size_t idx = kids.size();
    for(size_t k = 0; k < kids.size(); k++)
      if(kids[k] == done) idx = k;
    if(idx == kids.size())
      continue;  // not one of ours
    kids[idx] = -1;
    alive--;
    const int rc = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
    if(rc != 0 && !terminating)
    {
      fprintf(stderr, "vkrt_render: rank %zu ended with code %d; stopping the other %zu rank(s)\n", idx, rc, alive);
      result = rc;
      terminating = true;
      killRest(SIGTERM);
      // a rank stuck inside a collective may ignore SIGTERM for a while: give it two seconds, then insist
      for(int spin = 0; spin < 20 && alive > 0; spin++)
      {
        int s2 = 0;
        pid_t d2;
        while((d2 = waitpid(-1, &s2, WNOHANG)) > 0)
          for(size_t k = 0; k < kids.size(); k++)
            if(kids[k] == d2) { kids[k] = -1; alive--; }
        if(alive > 0) usleep(100000);
      }
      killRest(SIGKILL);
    }
  }
  cleanup();
  return result;
}

int main(int argc, char** argv)
{
  std::string cfgPath = "config.json", outOverride, idFile;
  int device = 0, ranks = 0, rank = 0, world = 1;
  for(int i = 1; i < argc; i++)
  {
    if(!strcmp(argv[i], "--ranks") && i + 1 < argc) { ranks = atoi(argv[++i]); continue; }
    if(!strcmp(argv[i], "--rank") && i + 1 < argc) { rank = atoi(argv[++i]); continue; }
    if(!strcmp(argv[i], "--world") && i + 1 < argc) { world = atoi(argv[++i]); continue; }
    if(!strcmp(argv[i], "--rccl-id-file") && i + 1 < argc) { idFile = argv[++i]; continue; }
    if(!strcmp(argv[i], "--config") && i + 1 < argc) cfgPath = argv[++i];
    else if(!strcmp(argv[i], "--output") && i + 1 < argc) outOverride = argv[++i];
    else if(!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
    else if(!strcmp(argv[i], "-h") || !strcmp(argv[i], "--help"))
    {
      printf("usage: vkrt_render [--config config.json] [--output prefix] [--device N]\n"
             "                   [--ranks N]                                  one process per GPU on this node, strips gathered over RCCL\n"
             "                   [--rank R --world N --rccl-id-file PATH]     one rank of a job started by another launcher\n");
      return 0;
    }
    else cfgPath = argv[i];
  }
  if(ranks > 0)
    return spawnRanks(ranks, argc, argv);  // (nothing above initialises the GPU)
  if(world < 1 || rank < 0 || rank >= world || (world > 1 && idFile.empty()))
  {
    fprintf(stderr, "vkrt_render: --rank/--world/--rccl-id-file inconsistent\n");
    return 2;
  }
  const bool gatherMode = !idFile.empty();
  try
  {
    const AppConfig cfg = loadConfig(cfgPath);
    const size_t slash = cfgPath.find_last_of("/\\");
    const std::string cfgDir = slash == std::string::npos ? "." : cfgPath.substr(0, slash);
    std::string scenePath = cfg.scenePath();
    if(!scenePath.empty() && scenePath[0] != '/') scenePath = cfgDir + "/" + scenePath;

    HelloVkrt helloVk(device);
    helloVk.setup(cfg.width, cfg.height);                       // main.cpp:209-216
    helloVk.setShard((uint32_t)rank, (uint32_t)world);
    std::unique_ptr<StripGather> gather;
    if(gatherMode)
    {
      StripLayout L;
      L.width = (uint32_t)cfg.width; L.height = (uint32_t)cfg.height; L.stripRows = 16; L.world = (uint32_t)world;
      gather.reset(new StripGather(L, (uint32_t)rank, device, idFile));
    }
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
    helloVk.m_buildFlags = cfg.build == "lbvh" ? VKRT_BUILD_LBVH_GPU : cfg.build == "sah" ? VKRT_BUILD_SAH_HOST : VKRT_BUILD_PLOC_GPU;
    helloVk.m_watertight = cfg.watertight; helloVk.m_anyHitDissolve = cfg.anyHitDissolve; helloVk.m_skipDeadShadowRays = cfg.skipDeadShadowRays;
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
    for(int f = 0; f < cfg.frames;)                             // main.cpp:441 loop body
    {
      helloVk.updateUniformBuffer();                            // main.cpp:503
      helloVk.updateFrame();                                    // main.cpp:504
      helloVk.m_seed = (uint32_t)cfg.seed + (cfg.seedPerFrame ? (uint32_t)f : 0u);
      int done = 1;
      if(!hybrid)
      {
        // the camera is at rest for the whole run: "framesPerCall" iterations of the loop go to the library as one call
        done = std::min(cfg.framesPerCall, cfg.frames - f);
        if(done > 1)
          helloVk.pathtraceFrames(cfg.clearColor, done, cfg.seedPerFrame);
        else
          helloVk.pathtrace(cfg.clearColor);                    // main.cpp:507
        if(gather && f + done == cfg.frames)  // every rank keeps accumulating its own strips; the image is needed once, at the end
          gather->gather(helloVk.offscreenDevice(), nullptr);
      }
      else
      {
        helloVk.rasterizeGltf(cfg.clearColor);                  // main.cpp:513
        helloVk.raytraceRasterizedScene();                      // main.cpp:547
        if(gather && f + 1 == cfg.frames)  // the ranks composite their own strips (post.frag is per pixel); the display strips travel
          gather->gather(helloVk.drawPostDevice(), nullptr);
      }
      traceMs += helloVk.lastTraceMs();
      f += done;
    }
    const vkrt_counters c = helloVk.counters();
    const double rays = (double)(c.rays_closest + c.rays_shadow);
    printf("%d frame(s) %dx%d, %d spp/frame, depth %d: %.3f ms GPU, %.1f Mrays/s (%llu closest + %llu shadow rays)\n", cfg.frames,
           cfg.width, cfg.height, cfg.samples, cfg.depth, traceMs, traceMs > 0 ? rays / traceMs / 1e3 : 0.0,
           (unsigned long long)c.rays_closest, (unsigned long long)c.rays_shadow);
    const std::string out = !outOverride.empty() ? outOverride : cfg.output;
    if(gather)
    {
      // rank 0 writes the gathered linear image (PFM) and its gamma preview; all ranks hold the same full image
      if(hipDeviceSynchronize() != hipSuccess) throw std::runtime_error("hipDeviceSynchronize failed");
      if(rank == 0 && !out.empty())
      {
        std::vector<float> img((size_t)cfg.width * cfg.height * 4);
        if(hipMemcpy(img.data(), gather->fullImage(), img.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) throw std::runtime_error("hipMemcpy failed");
        if(hybrid)
        {  // what was gathered is the display image (after post.frag's gamma): PNG as is, linear values for the PFM / PPM writers
          writePNG(out + ".png", img, cfg.width, cfg.height);
          for(float& v : img) v = std::pow(v, 2.2f);
        }
        writePFM(out + ".pfm", img, cfg.width, cfg.height);
        writePPM(out + ".ppm", img, cfg.width, cfg.height);
        printf("rank 0 of %d wrote %s.pfm / %s.ppm (strips gathered with one RCCL all-gather)\n", world, out.c_str(), out.c_str());
      }
    }
    else if(!out.empty())
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
