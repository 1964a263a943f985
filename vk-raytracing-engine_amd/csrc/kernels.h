// kernels.h -- host-callable launch wrappers implemented in the .hip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include "device_scene.h"

hipError_t vkrt_launch_pathtrace(const TraceParams& P, unsigned gridBlocks, bool count, hipStream_t stream);
int        vkrt_pathtrace_block_size();
hipError_t vkrt_pathtrace_occupancy(size_t ldsBytes, int* blocksPerCU);
hipError_t vkrt_launch_trace_rays(const DevScene& sc, unsigned n, const float* o, const float* d, float tmin, float tmax, int anyHit,
                                  float* t, float* u, float* v, int* gid, hipStream_t stream);
hipError_t vkrt_launch_eval_math(int op, unsigned n, const float* a, const float* b, float* out, hipStream_t stream);

// wavefront mode (wavefront.hip)
struct WfTiming
{
  hipEvent_t* events;   // optional pool of 2*rounds+2 events (NULL = no per-kernel timing)
  int capacity;
  int used;             // pairs recorded around k_wf_traverse launches
};
// Lanes.  A call renders `frames` progressive frames of one shard.  Its work is dealt to up to VKRT_WF_MAX_LANES lanes, each with its
// own record streams and its own HIP stream, so that the kernels of different lanes overlap (a draining traversal launch of one lane
// is filled up by the launches of the others):
//   * a call of several frames keeps `inFlight` consecutive frames in flight, one lane each (frame k belongs to group k % inFlight).
//     A frame in flight writes its pixel values into its group's staging plane, and a blend kernel at its end applies
//     raytrace.rgen:136-141 in frame order (it waits for the blend of frame k - 1): the image is what single-frame calls would
//     have left, bit for bit.  Every launch keeps its full size;
//   * a single frame is split spatially into `subframes` tile ranges, one lane each (a pixel is always blended by its own lane).
#define VKRT_WF_MAX_LANES 8
struct WfAsync
{
  hipStream_t streams[VKRT_WF_MAX_LANES];  // internal streams (created by the caller of vkrt_launch_wavefront)
  hipEvent_t fork, join[VKRT_WF_MAX_LANES];
  int count;                                // usable entries (0/1 = everything on the caller's stream)
  hipEvent_t* pool;                         // events for the blend order of the frames of one call, no timing
  int poolSize;
};
// events a call with these parameters takes from WfAsync::pool
inline int vkrt_wf_pool_events(int frames, int lanes) { return frames * lanes; }
size_t     vkrt_wf_state_bytes(uint32_t pathCapacity, int groups);
void       vkrt_wf_carve(void* base, uint32_t pathCapacity, int groups, WfBuffers* B);
struct WfOptions
{
  int subframes;   // VKRT_OPT_WF_SUBFRAMES
  int travBlock;   // VKRT_OPT_WF_TRAV_BLOCK (64 / 128 / 256)
  int inFlight;    // VKRT_OPT_WF_FRAMES_IN_FLIGHT (clamped to the frames of the call and to WfBuffers::groups)
};
// frames >= 1: frame k uses pc.frame + k and seed + k * seedStep
hipError_t vkrt_launch_wavefront(const TraceParams& P, const WfBuffers& B, const WfOptions& opt, int frames, uint32_t seedStep, bool count,
                                 hipStream_t stream, WfTiming* timing, const WfAsync* async);
// one launch of k_wf_traverse on the records of round r (wf_traverse.hip): travBlock 64 / 128 / 256 threads per workgroup, tg = grid,
// tlds = LDS bytes of the per-lane stacks; count = the instrumented instantiation.  Launch errors surface through hipGetLastError.
void       vkrt_wf_launch_traverse(const TraceParams& P, const WfBuffers& B, int r, unsigned travBlock, bool count, dim3 tg, size_t tlds,
                                   hipStream_t stream);

// hybrid mode (hybrid.hip)
struct NrdPlanes  // optional NRD front-end attachments (include/vkrt.h vkrt_nrd_planes) + the raster pass's view matrix
{
  float* normRough;
  float* viewZ;
  float* radHitD;
  float viewMatrix[16];
};
hipError_t vkrt_launch_gbuffer(const TraceParams& P, const float clearColor[4], int lightsCount, float* color, float* position, float* normal,
                               float* rough, const NrdPlanes* nrd, hipStream_t stream);
// giLater: non-NULL = the GI part follows on the wavefront streams (vkrt_launch_hybrid_gi): the kernel does shadows + AO only and
// leaves (seed, visibility) per pixel there instead of writing the accumulation image
hipError_t vkrt_launch_hybrid(const TraceParams& P, const float* color, const float* position, const float* normal, const float* rough, float* accum,
                              const NrdPlanes* nrd, uint2* giLater, hipStream_t stream);
// GI of the hybrid mode on the path tracer's wavefront streams (wavefront.hip): k_hybrid (direct part, tmp = per-pixel seed and
// visibility) -> k_hy_gi_init (first GI ray of every shaded pixel) -> pc.depth rounds of traverse / shade -> accumulation image.
struct HybridGi
{
  const float4* color;     // G-buffer planes (read)
  const float4* position;
  const float4* normal;
  const float2* rough;
  float4* accum;           // rgba32f accumulation image (read-modify-write)
  float4* nrdRadHitD;      // optional NRD plane (NULL = not requested)
  const float* nrdViewZ;
};
hipError_t vkrt_launch_hybrid_gi(const TraceParams& P, const WfBuffers& B, const HybridGi& G, unsigned travBlock, hipStream_t stream);
// the scratch plane k_hybrid leaves the per-pixel (seed, visibility) in for vkrt_launch_hybrid_gi
uint2* vkrt_wf_hybrid_tmp(const WfBuffers& B);
hipError_t vkrt_launch_post(int rtMode, int viewAccumulated, int useGI, unsigned n, const float* mainImg, const float* rtImg, float* out,
                            hipStream_t stream);
