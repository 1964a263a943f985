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
